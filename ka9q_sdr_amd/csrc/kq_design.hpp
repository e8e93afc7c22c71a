// kq_design.hpp -- host-side (control-plane) filter response design for libka9q_hip.
// Runs once per retune, never per block: Kaiser-windowed frequency responses as the reference
// designs them in set_filter / window_filter / window_rfilter (filter.c:337-546) and the FM
// de-emphasis response of fm.c:54-66.
#pragma once
#include <complex>
#include <vector>

namespace kq {

using cfloat = std::complex<float>;

enum FilterType { FT_NONE = 0, FT_COMPLEX = 1, FT_CROSS_CONJ = 2, FT_REAL = 3 };  // filter.h:17-22

// Small host FFT (power of two, float, unnormalised; sign -1 forward / +1 backward)
void host_fft(std::vector<cfloat> &v, int sign);

void make_kaiser(float *window, unsigned M, float beta);                  // filter.c:337-357
int window_filter(int L, int M, std::vector<cfloat> &response, float beta);   // filter.c:365-415
int window_rfilter(int L, int M, std::vector<cfloat> &response, float beta);  // filter.c:420-469

// set_filter (filter.c:500-546): brick wall between low..high (cycles/sample at the OUTPUT rate),
// Kaiser windowed.  N = master FFT size, L_dec = olen, M_dec = (M-1)/D+1.
std::vector<cfloat> design_response(int N, int L_dec, int M_dec, int out_type, float low, float high, float beta);

// noise_gain (filter.c:472-497)
float noise_gain(const std::vector<cfloat> &response, int N, int n_dec, bool real_in, int out_type);

// FM post-detection response (fm.c:42, 56-65): 300 Hz high-pass, -6 dB/octave to 6 kHz, Kaiser
// windowed for a REAL->REAL filter of AL new samples and AM taps.  Returns AN/2+1 bins.
std::vector<cfloat> design_fm_audio_response(int AL, int AM, float dsamprate, float beta);

}  // namespace kq
