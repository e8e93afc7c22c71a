/* kq_packet.c -- oracle restatement of the AFSK-1200 / HDLC decoder (test infrastructure only).
 * Follows packet.c:36-48 (constants), packet.c:201-212 (PCM ingest into the REAL master), packet.c:267-414
 * (decode_task) and ax25.c:138-156 (crc_good).  crc_good is pinned against oracle/_ref/libref_ax25.so, the
 * reference's ax25.c compiled unmodified; the decoder itself sits on filter.c (FFTW), so it is PARITY UNPINNED like
 * the rest of the filter clients and is checked on synthesized AX.25 frames with known content.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "kq_oracle.h"

/* ax25.c:138-156: bit-serial CRC-CCITT (reflected 0x8408), good residue 0xf0b8 */
int kqo_crc_good(const unsigned char *frame, int length){
  unsigned short crc = 0xffff;
  while(length-- > 0){
    unsigned char byte = *frame++;
    for(int i = 0; i < 8; i++){
      unsigned short feedback = 0;
      if((crc ^ byte) & 1)
        feedback = 0x8408;
      crc = (crc >> 1) ^ feedback;
      byte >>= 1;
    }
  }
  return crc == 0xf0b8;
}

struct kqo_afsk {
  kqo_filter_in *in;           /* packet.c:190 create_filter_input(AL, AM, REAL) */
  kqo_filter_out *out;         /* packet.c:272-273 */
  int input_pointer;
  kqo_osc mark, space;         /* packet.c:276-284 */
  int symphase;
  float complex mark_accum, space_accum, mark_offset_accum, space_offset_accum;
  float last_val, mid_val;
  unsigned char hdlc_frame[KQO_AFSK_FRAME_MAX];
  int frame_bit, flagsync, ones;
  int decoded_packets;
  /* decoded frames, back to back */
  unsigned char *frames;
  int *lens;
  int nframes, cap_frames;
  size_t frame_bytes, cap_bytes;
  long long blocks;
};

static float cnrmf_(float complex x){ return crealf(x) * crealf(x) + cimagf(x) * cimagf(x); } /* dsp.h:26-28 */

kqo_afsk *kqo_afsk_create(void){
  kqo_afsk *a = calloc(1, sizeof *a);
  if(!a)
    return NULL;
  a->in = kqo_create_filter_input(KQO_AFSK_AL, KQO_AFSK_AM, KQO_REAL);
  a->out = kqo_create_filter_output(a->in, NULL, 1, KQO_COMPLEX);
  /* analytic, band-limited signal: +100 .. +4000 Hz, beta 3.0 */
  kqo_set_filter(a->out, +100.f / KQO_AFSK_SAMPRATE, +4000.f / KQO_AFSK_SAMPRATE, 3.0f);
  kqo_set_osc(&a->mark, -1200. / KQO_AFSK_SAMPRATE, 0.0);
  kqo_set_osc(&a->space, -2200. / KQO_AFSK_SAMPRATE, 0.0);
  return a;
}

void kqo_afsk_destroy(kqo_afsk *a){
  if(!a)
    return;
  kqo_delete_filter_output(a->out);
  kqo_delete_filter_input(a->in);
  free(a->frames);
  free(a->lens);
  free(a);
}

static void emit_frame(kqo_afsk *a, int bytes){
  if(a->nframes == a->cap_frames){
    a->cap_frames = a->cap_frames ? 2 * a->cap_frames : 16;
    a->lens = realloc(a->lens, sizeof(int) * a->cap_frames);
  }
  if(a->frame_bytes + bytes > a->cap_bytes){
    a->cap_bytes = 2 * (a->cap_bytes + bytes);
    a->frames = realloc(a->frames, a->cap_bytes);
  }
  memcpy(a->frames + a->frame_bytes, a->hdlc_frame, bytes);
  a->frame_bytes += bytes;
  a->lens[a->nframes++] = bytes;
  a->decoded_packets++;
}

/* packet.c:302-410: one filter block through the correlators, the bit clock and the HDLC deframer */
static void decode_block(kqo_afsk *a){
  int const samppbit = KQO_AFSK_SAMPPBIT;
  kqo_execute_filter_output(a->out);
  a->blocks++;
  for(unsigned n = 0; n < a->out->olen; n++){
    float complex s;
    s = a->out->output_c[n] * kqo_step_osc(&a->mark);    /* float complex x double complex, rounded to float */
    a->mark_accum += s;
    a->mark_offset_accum += s;
    s = a->out->output_c[n] * kqo_step_osc(&a->space);
    a->space_accum += s;
    a->space_offset_accum += s;

    if(++a->symphase == samppbit / 2){
      a->mid_val = cnrmf_(a->mark_offset_accum) - cnrmf_(a->space_offset_accum);
      a->mark_offset_accum = a->space_offset_accum = 0;
    }
    if(a->symphase < samppbit)
      continue;

    a->symphase = 0;
    float const cur_val = cnrmf_(a->mark_accum) - cnrmf_(a->space_accum);
    a->mark_accum = a->space_accum = 0;

    if(cur_val * a->last_val < 0){
      /* transition: Gardner-style clock adjustment, NRZI zero */
      a->symphase += ((cur_val - a->last_val) * a->mid_val) > 0 ? +1 : -1;
      if(a->ones == 6){
        if(a->flagsync){
          a->frame_bit -= 7;
          int const bytes = a->frame_bit / 8;
          if(bytes > 0 && bytes <= KQO_AFSK_FRAME_MAX && kqo_crc_good(a->hdlc_frame, bytes))
            emit_frame(a, bytes);
        }
        memset(a->hdlc_frame, 0, sizeof a->hdlc_frame);
        a->frame_bit = 0;
        a->flagsync = 1;
      } else if(a->ones == 5){
        /* stuffed zero dropped */
      } else if(a->ones < 5){
        if(a->flagsync)
          a->frame_bit++;
      }
      a->ones = 0;
    } else {
      /* NRZI one */
      if(++a->ones == 7){
        memset(a->hdlc_frame, 0, sizeof a->hdlc_frame);
        a->frame_bit = 0;
        a->flagsync = 0;
      } else if(a->flagsync){
        /* the reference indexes hdlc_frame[1024] without a bound (packet.c:401); beyond it the write is undefined
         * there, dropped here -- the bit count keeps running, so such a frame can never pass the length test */
        if(a->frame_bit >= 0 && a->frame_bit < 8 * KQO_AFSK_FRAME_MAX)
          a->hdlc_frame[a->frame_bit / 8] |= 1 << (a->frame_bit % 8);
        a->frame_bit++;
      }
    }
    a->last_val = cur_val;
  }
}

/* float samples straight into input.r[] (what packet.c:207 stores) */
void kqo_afsk_push(kqo_afsk *a, const float *samples, int n){
  for(int i = 0; i < n; i++){
    a->in->input_r[a->input_pointer++] = samples[i];
    if(a->input_pointer == (int)a->in->ilen){
      kqo_execute_filter_input(a->in);
      a->input_pointer = 0;
      decode_block(a);
    }
  }
}

/* packet.c:204-207: big-endian PCM words; ntohs() yields an unsigned value, so negative samples arrive as
 * 32768..65535 times SCALE (a latent quirk of the reference, kept) */
void kqo_afsk_push_pcm_be(kqo_afsk *a, const unsigned char *be, int nwords){
  for(int i = 0; i < nwords; i++){
    unsigned const v = ((unsigned)be[2 * i] << 8) | be[2 * i + 1];
    float const f = v * (float)(1. / 32768);
    kqo_afsk_push(a, &f, 1);
  }
}

int kqo_afsk_nframes(const kqo_afsk *a){ return a->nframes; }
int kqo_afsk_frame(const kqo_afsk *a, int i, unsigned char *dst, int cap){
  if(i < 0 || i >= a->nframes)
    return -1;
  size_t off = 0;
  for(int k = 0; k < i; k++)
    off += a->lens[k];
  int const n = a->lens[i] < cap ? a->lens[i] : cap;
  memcpy(dst, a->frames + off, n);
  return a->lens[i];
}
/* filter output of the block decoded last (olen complex) and the soft decision state, for parity checks */
const float complex *kqo_afsk_filter_output(const kqo_afsk *a){ return a->out->output_c; }
void kqo_afsk_state(const kqo_afsk *a, int *symphase, int *frame_bit, int *flagsync, int *ones, float *last_val,
                    float *mid_val){
  *symphase = a->symphase; *frame_bit = a->frame_bit; *flagsync = a->flagsync; *ones = a->ones;
  *last_val = a->last_val; *mid_val = a->mid_val;
}
