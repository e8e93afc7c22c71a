/* kq_decimate.c -- oracle restatement of the half-band decimators of the front-end daemons
 * (test infrastructure only).  Follows decimate.c:108-160 (portable branch) and the cascade of
 * hackrf.c:227-241, 295-300.  Pinned: tests compare it with oracle/_ref/libref_decimate.so, the reference's own
 * decimate.c compiled unmodified.
 */
#include <string.h>
#include "kq_oracle.h"

/* 15-tap half-band, unity centre tap, four coefficient pairs on the odd taps (decimate.c:108-144):
 * y[k] = e[k-3] + sum_i c[i] * (o[k-i] + o[k-7+i]),  e[k] = in[2k], o[k] = in[2k+1] */
void kqo_hb15_block(kqo_hb15_state *st, float *output, const float *input, int cnt){
  float even[4], odd[4], old_odd[4], c[4];
  memcpy(c, st->coeffs, sizeof c);
  memcpy(even, st->even_samples, sizeof even);
  memcpy(odd, st->odd_samples, sizeof odd);
  memcpy(old_odd, st->old_odd_samples, sizeof old_odd);
  while(cnt--){
    even[0] = *input++;
    odd[0] = *input++;
    float result = even[3];
    for(int i = 2; i >= 0; i--)
      even[i + 1] = even[i];
    for(int i = 0; i < 4; i++)
      result += (odd[i] + old_odd[i]) * c[i];
    *output++ = result;
    for(int i = 0; i < 3; i++)
      old_odd[i] = old_odd[i + 1];
    old_odd[3] = odd[3];
    for(int i = 2; i >= 0; i--)
      odd[i + 1] = odd[i];
  }
  memcpy(st->even_samples, even, sizeof even);
  memcpy(st->odd_samples, odd, sizeof odd);
  memcpy(st->old_odd_samples, old_odd, sizeof old_odd);
}

/* 3-tap half-band 1,2,1 (decimate.c:146-160) */
void kqo_hb3_block(float *state, float *output, const float *input, int cnt){
  float old = *state;
  while(cnt--){
    float const a = *input++, b = *input++;
    *output++ = 2 * a + b + old;
    old = b;
  }
  *state = old;
}

/* Goodman/Carey F8 coefficients as hackrf.c:229-238 sets them ([3] next to the centre, [0] at the tails) */
void kqo_hb15_init(kqo_hb15_state *st){
  memset(st, 0, sizeof *st);
  st->coeffs[3] = 490. / 802;
  st->coeffs[2] = -116. / 802;
  st->coeffs[1] = 33. / 802;
  st->coeffs[0] = -6. / 802;
}
