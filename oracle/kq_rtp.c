/* kq_rtp.c -- oracle restatement of the I/Q packet ingest rules (test infrastructure only): RTP header parsing
 * (multicast.c:242-277), the payload types and the obsolete 24-byte status block that radio skips (multicast.h:15-20,
 * main.c:315-341, sdr.h:18-48), the sequence / timestamp bookkeeping that decides how many samples enter the filter
 * (multicast.c:305-340) and proc_samples' use of it (radio.c:62-104).  PARITY UNPINNED: multicast.c includes
 * <bsd/string.h>, absent here; checked on hand-built packet sequences with known outcomes.
 */
#include <string.h>
#include "kq_oracle.h"

static unsigned get16(const unsigned char *p){ return ((unsigned)p[0] << 8) | p[1]; }
static uint32_t get32(const unsigned char *p){ return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* multicast.c:242-277; returns the offset of the first byte after the header (and extension) */
int kqo_ntoh_rtp(kqo_rtp_header *rtp, const unsigned char *data){
  const unsigned char *dp = data;
  rtp->version = *dp >> 6;
  rtp->pad = (*dp >> 5) & 1;
  rtp->extension = (*dp >> 4) & 1;
  rtp->cc = *dp & 0xf;
  dp++;
  rtp->marker = (*dp >> 7) & 1;
  rtp->type = *dp & 0x7f;
  dp++;
  rtp->seq = (uint16_t)get16(dp);
  dp += 2;
  rtp->timestamp = get32(dp);
  dp += 4;
  rtp->ssrc = get32(dp);
  dp += 4;
  dp += 4 * rtp->cc;
  if(rtp->extension){
    dp += 2;
    unsigned const ext_len = 4 + get16(dp);     /* as the reference computes it */
    dp += 2;
    dp += ext_len;
  }
  return (int)(dp - data);
}

/* multicast.c:305-340 */
int kqo_rtp_process(kqo_rtp_state *state, const kqo_rtp_header *rtp, int sampcnt){
  if(rtp->ssrc != state->ssrc){
    state->init = 0;
    state->ssrc = rtp->ssrc;
  }
  if(!state->init){
    state->packets = 0;
    state->seq = rtp->seq;
    state->timestamp = rtp->timestamp;
    state->dupes = 0;
    state->drops = 0;
    state->init = 1;
  }
  state->packets++;
  short const seq_step = (short)(rtp->seq - state->seq);
  if(seq_step != 0){
    if(seq_step < 0){
      state->dupes++;
      return -1;
    }
    state->drops += seq_step;
  }
  state->seq = rtp->seq + 1;
  int const time_step = (int)(rtp->timestamp - state->timestamp);
  if(time_step < 0)
    return time_step;
  state->timestamp = rtp->timestamp + sampcnt;
  return time_step;
}

/* One datagram as rtp_recv (main.c:315-341) and proc_samples (radio.c:62-104) treat it.
 * Returns 1 when samples enter the filter: *zeros lost samples to inject first, then *count samples of *format
 * (KQO_IQ_S16 / KQO_IQ_S8) starting at packet + *offset.  Returns 0 when the datagram is ignored or dropped. */
int kqo_iq_packet(kqo_iq_ingest *in, const unsigned char *packet, int size, int *zeros, int *offset, int *count, int *format){
  *zeros = *offset = *count = 0;
  *format = KQO_IQ_S16;
  if(size < 12)                                                 /* RTP_MIN_SIZE */
    return 0;
  kqo_rtp_header rtp;
  int const hdr = kqo_ntoh_rtp(&rtp, packet);
  size -= hdr;
  if(rtp.pad){
    size -= packet[hdr + size - 1];                             /* main.c:324-328 */
    rtp.pad = 0;
  }
  if(rtp.type != 97 && rtp.type != 98)                          /* IQ_PT, IQ_PT8 */
    return 0;
  int const off = hdr + 24;                                     /* obsolete status block, main.c:338-341 */
  size -= 24;
  int const sampcount = rtp.type == 97 ? size / 4 : size / 2;   /* radio.c:64-72 */
  if(rtp.ssrc != in->rtp.ssrc)
    in->samples = 0;                                            /* radio.c:73-77 */
  int const time_step = kqo_rtp_process(&in->rtp, &rtp, sampcount);
  if(time_step < 0 || time_step > 192000)                       /* radio.c:79-82 */
    return 0;
  if(time_step > 0)
    in->samples += time_step;                                   /* radio.c:88 */
  in->samples += sampcount;                                     /* radio.c:104 */
  *zeros = time_step;
  *offset = off;
  *count = sampcount;
  *format = rtp.type == 97 ? KQO_IQ_S16 : KQO_IQ_S8;
  return 1;
}
