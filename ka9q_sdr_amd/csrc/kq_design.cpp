// kq_design.cpp -- see kq_design.hpp.  Host only, control plane.
#include "kq_design.hpp"

#include <cmath>
#include <cstring>

namespace kq {

void host_fft(std::vector<cfloat> &v, int sign) {
  size_t const n = v.size();
  if (n < 2) return;
  unsigned bits = 0;
  while ((size_t(1) << bits) < n) bits++;
  for (size_t i = 0; i < n; i++) {
    size_t r = 0;
    for (unsigned b = 0; b < bits; b++)
      if (i >> b & 1) r |= size_t(1) << (bits - 1 - b);
    if (r > i) std::swap(v[i], v[r]);
  }
  std::vector<cfloat> tw(n / 2);
  for (size_t k = 0; k < n / 2; k++) {
    double const a = sign * 2.0 * M_PI * double(k) / double(n);
    tw[k] = cfloat(float(std::cos(a)), float(std::sin(a)));
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    size_t const half = len / 2, stride = n / len;
    for (size_t base = 0; base < n; base += len)
      for (size_t j = 0; j < half; j++) {
        cfloat const w = tw[j * stride];
        cfloat const a = v[base + j], b = v[base + j + half];
        cfloat const t(b.real() * w.real() - b.imag() * w.imag(), b.real() * w.imag() + b.imag() * w.real());
        v[base + j] = a + t;
        v[base + j + half] = a - t;
      }
  }
}

// I0 by power series in float (filter.c:282-293)
static float i0f(float x) {
  float const t = 0.25 * x * x;
  float sum = 1 + t, term = t;
  for (int k = 2; k < 40; k++) {
    term *= t / (k * k);
    sum += term;
    if (term < 1e-12 * sum) break;
  }
  return sum;
}

void make_kaiser(float *window, unsigned M, float beta) {
  float const numc = M_PI * beta;
  float const inv_denom = 1. / i0f(numc);
  float const pc = 2.0 / (M - 1);
  for (unsigned n = 0; n < M / 2; n++) {
    float const p = pc * n - 1;
    window[M - 1 - n] = window[n] = i0f(numc * std::sqrt(1 - p * p)) * inv_denom;
  }
  if (M & 1) window[(M - 1) / 2] = 1;
}

int window_filter(int L, int M, std::vector<cfloat> &response, float beta) {
  int const N = L + M - 1;
  if ((int)response.size() != N) return -1;
  std::vector<cfloat> buf(response);
  host_fft(buf, +1);
  std::vector<float> win(M);
  make_kaiser(win.data(), (unsigned)M, beta);
  float const gain = 1. / N;
  for (int n = M - 1; n >= 0; n--) buf[n] = buf[(n - M / 2 + N) % N] * win[n] * gain;
  for (int n = M; n < N; n++) buf[n] = 0;
  host_fft(buf, -1);
  response = buf;
  return 0;
}

int window_rfilter(int L, int M, std::vector<cfloat> &response, float beta) {
  int const N = L + M - 1;
  if ((int)response.size() != N / 2 + 1) return -1;
  // c2r: Hermitian extension, imaginary parts of DC and Nyquist ignored
  std::vector<cfloat> full(N);
  full[0] = response[0].real();
  full[N / 2] = response[N / 2].real();
  for (int k = 1; k < N / 2; k++) {
    full[k] = response[k];
    full[N - k] = std::conj(response[k]);
  }
  host_fft(full, +1);
  std::vector<float> tb(N);
  for (int n = 0; n < N; n++) tb[n] = full[n].real();
  std::vector<float> win(M);
  make_kaiser(win.data(), (unsigned)M, beta);
  float const gain = 1. / N;
  for (int n = M - 1; n >= 0; n--) tb[n] = tb[(n - M / 2 + N) % N] * win[n] * gain;
  for (int n = M; n < N; n++) tb[n] = 0;
  for (int n = 0; n < N; n++) full[n] = tb[n];
  host_fft(full, -1);
  for (int k = 0; k <= N / 2; k++) response[k] = full[k];
  return 0;
}

std::vector<cfloat> design_response(int N, int L_dec, int M_dec, int out_type, float low, float high, float beta) {
  int const N_dec = L_dec + M_dec - 1;
  float gain = 1. / ((float)N);
  if (out_type == FT_REAL || out_type == FT_CROSS_CONJ) gain *= M_SQRT1_2;
  std::vector<cfloat> r(N_dec);
  for (int n = 0; n < N_dec; n++) {
    float const f = (n <= N_dec / 2) ? (float)n / N_dec : (float)(n - N_dec) / N_dec;
    r[n] = (f >= low && f <= high) ? gain : 0;
  }
  window_filter(L_dec, M_dec, r, beta);
  return r;
}

float noise_gain(const std::vector<cfloat> &response, int N, int n_dec, bool real_in, int out_type) {
  int const count = (real_in && out_type == FT_REAL) ? n_dec / 2 + 1 : n_dec;
  float sum = 0;
  for (int i = 0; i < count; i++) sum += response[i].real() * response[i].real() + response[i].imag() * response[i].imag();
  if (out_type == FT_REAL || out_type == FT_CROSS_CONJ) return 2 * N * sum;
  return N * sum;
}

std::vector<cfloat> design_fm_audio_response(int AL, int AM, float dsamprate, float beta) {
  int const AN = AL + AM - 1;
  float const filter_gain = 10. / AN;
  std::vector<cfloat> r(AN / 2 + 1, cfloat(0, 0));
  for (int j = 0; j <= AN / 2; j++) {
    float const f = (float)j * dsamprate / AN;
    if (f >= 300 && f <= 6000) r[j] = filter_gain * 300. / f;
  }
  window_rfilter(AL, AM, r, beta);
  return r;
}

}  // namespace kq
