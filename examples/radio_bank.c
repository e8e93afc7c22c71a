/* radio_bank.c -- the channel-bank surface of libka9q_hip.so from plain C, the way a multi-channel `radio`
 * would drive it.  One FM channel on a synthetic 192 kHz stream carrying a 1 kHz tone at
 * 3 kHz peak deviation; prints the measured deviation and offset (fm.c:146-158) per block.
 *
 *   gcc -std=gnu11 -O2 -Iinclude examples/radio_bank.c -Lka9q_sdr_amd/lib -lka9q_hip -Wl,-rpath,$PWD/ka9q_sdr_amd/lib -lm
 */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ka9q_hip.h"

int main(void){
  if(kq_device_count() <= 0){
    fprintf(stderr, "no HIP device: %s\n", kq_last_error());
    return 2;
  }
  unsigned const L = 3840 * 2, M = 513;               /* N = 8192 */
  kq_bank_config bc = { .device = 0, .samprate = 192000, .L = L, .M = M, .decimate = 4, .max_channels = 1,
                        .max_blocks = 4, .gain_factor = 1.0f, .compute_n0 = 1, .fwd_mode = KQ_FWD_AUTO };
  kq_bank *bank = kq_bank_create(&bc);
  if(!bank){
    fprintf(stderr, "kq_bank_create: %s\n", kq_last_error());
    return 1;
  }
  kq_channel_config cc = { .demod_type = KQ_FM_DEMOD, .channels = 1, .low = -8000, .high = 8000, .kaiser_beta = 3.0f,
                           .headroom = 0.1778f, .second_lo = -20000.0 };     /* carrier at +20 kHz */
  int const ch = kq_bank_add_channel(bank, &cc);
  if(ch < 0){
    fprintf(stderr, "kq_bank_add_channel: %s\n", kq_last_error());
    return 1;
  }
  unsigned const nblocks = 4;
  size_t const n = (size_t)L * nblocks;
  float complex *iq = malloc(n * sizeof *iq);
  double phase = 0;
  unsigned lcg = 12345u;
  for(size_t i = 0; i < n; i++){
    double const t = (double)i / 192000.;
    double const f = 20000. + 3000. * cos(2 * M_PI * 1000. * t);            /* instantaneous frequency */
    phase += 2 * M_PI * f / 192000.;
    /* a little noise, as every real signal has: the FM SNR estimate (fm.c:100-102) divides by the envelope's variance,
     * which for a synthetic constant envelope is float rounding of either sign -- and a negative one reads as SNR 0, which
     * closes the squelch (at 2e-3 of noise the in-channel SNR was 63 dB, the variance three units of float rounding: any
     * change in the order of the kernel's sums could flip it; 1e-2 leaves a factor 30) */
    lcg = lcg * 1664525u + 1013904223u;
    float const nr = ((lcg >> 8) & 0xffff) / 65536.f - 0.5f;
    lcg = lcg * 1664525u + 1013904223u;
    float const ni = ((lcg >> 8) & 0xffff) / 65536.f - 0.5f;
    iq[i] = (0.5f * (float)cos(phase) + 1e-2f * nr) + (0.5f * (float)sin(phase) + 1e-2f * ni) * I;
  }
  if(kq_bank_push_iq(bank, iq, n, KQ_IQ_CF32, 0) != 0 || kq_bank_process(bank) != (int)nblocks || kq_bank_sync(bank) != 0){
    fprintf(stderr, "processing failed: %s\n", kq_last_error());
    return 1;
  }
  int rc = 0;
  for(unsigned b = 0; b < nblocks; b++){
    kq_chan_status st;
    float audio[4096];
    size_t got = 0;
    kq_bank_pull_status(bank, ch, b, &st);
    kq_bank_pull_audio(bank, ch, b, audio, sizeof audio / sizeof *audio, &got);
    printf("block %u: nout %d pdeviation %.1f Hz foffset %.1f Hz snr %.1f n0 %.3g\n", b, st.nout, st.pdeviation, st.foffset,
           st.snr, st.n0);
    if(b > 0 && fabsf(st.pdeviation - 3000.f) > 150.f)
      rc = 3;
    if(got != (size_t)st.nout)
      rc = 4;
  }
  kq_bank_destroy(bank);
  free(iq);
  puts(rc == 0 ? "ok" : "unexpected result");
  return rc;
}
