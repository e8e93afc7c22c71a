// kq_design.hpp -- filter response design for libka9q_hip (control plane: per retune, never per block).  The work is
// done by device kernels (kq_design.hip): Kaiser-windowed frequency responses as the reference designs them in
// set_filter / window_filter / window_rfilter (filter.c:337-546) and the FM de-emphasis response of fm.c:54-66.
// Every function runs on the calling thread's current device and fails (-1 / empty vector) without one.
#pragma once
#include <complex>
#include <vector>

namespace kq {

using cfloat = std::complex<float>;

enum FilterType { FT_NONE = 0, FT_COMPLEX = 1, FT_CROSS_CONJ = 2, FT_REAL = 3 };  // filter.h:17-22

struct DesignJob {  // one response of a batch (device layout)
  float low, high;  // band edges in cycles per sample (brick wall), or low = bin spacing in Hz (de-emphasis curve)
  float beta;       // Kaiser window shape
  float gain;       // passband value of the target spectrum
};
struct DesignTarget {  // where a response designed on a stream goes (design_launch); resp == null: the job is void
  void *resp;          // (L_dec + M_dec - 1) complex bins, device memory
  float *noise_gain;   // device memory, or null
  float ng_scale;
  float pad;
};
struct BandEdges {
  float low, high, beta;
};

int make_kaiser(float *window, unsigned M, float beta);                       // filter.c:337-357
int window_filter(int L, int M, std::vector<cfloat> &response, float beta);   // filter.c:365-415
int window_rfilter(int L, int M, std::vector<cfloat> &response, float beta);  // filter.c:420-469

// set_filter (filter.c:500-546) for a batch of slaves of one geometry: brick walls between low..high (cycles/sample
// at the OUTPUT rate), Kaiser windowed; N = master FFT size, L_dec = olen, M_dec = (M-1)/D+1.  One kernel launch.
// responses: edges.size() * (L_dec + M_dec - 1) bins; noise_gains: filter.c:472-497 for a complex master.
int design_responses(int N, int L_dec, int M_dec, int out_type, const std::vector<BandEdges> &edges, std::vector<cfloat> &responses,
                     std::vector<float> &noise_gains);
// one of them; empty on failure
std::vector<cfloat> design_response(int N, int L_dec, int M_dec, int out_type, float low, float high, float beta,
                                    float *noise_gain_out = nullptr);

// design_responses queued on a stream, results left on the device (kq_design.hip); hipStream_t passed as void *
int design_prepare(int L_dec, int M_dec);
// ctl_queue != null: `ctl_records` write records of a bank's control queue (kq_ctl.hpp) are applied by the same launch
int design_launch(void *stream, int L_dec, int M_dec, const DesignJob *jobs, const DesignTarget *targets, unsigned count, void *scratch,
                  const void *ctl_queue = nullptr, unsigned ctl_records = 0);
void design_scales(int N, int out_type, float *gain, float *ng_scale);

// FM post-detection response (fm.c:42, 56-65): 300 Hz high-pass, -6 dB/octave to 6 kHz, Kaiser
// windowed for a REAL->REAL filter of AL new samples and AM taps.  Returns AN/2+1 bins.
std::vector<cfloat> design_fm_audio_response(int AL, int AM, float dsamprate, float beta);

}  // namespace kq
