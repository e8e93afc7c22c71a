/* ref_audio_capture.c -- TEST INFRASTRUCTURE: drives the reference's send_mono_output / send_stereo_output
 * (audio.c:32-132, compiled in place into oracle/_ref/libref_audio.so by oracle/Makefile) without a network, so that
 * the oracle's PCM packetiser (kq_chan.c kqo_pcm_rtp) and the product's kq_bank_pull_rtp_audio can be pinned on what the
 * reference really sends.  Compiled only where the reference tree is present (it uses the reference's own radio.h for
 * struct demod), together with audio.c.
 *
 * audio.c has two external references that live in multicast.c, which this image cannot build (<bsd/string.h>):
 *   setup_mcast  -- only called from setup_output(), never reached here: returns -1;
 *   hton_rtp     -- serialises struct rtp_header.  It is bound to a CAPTURE function here: the twelve bytes it writes
 *                   are this harness's own record of the fields audio.c filled in (not the RTP wire format, which
 *                   therefore stays unpinned):  0xA5, marker<<7 | type, seq (LE16), timestamp (LE32), ssrc (LE32).
 * Everything else in the datagram -- the payload words, how the block is cut into 480-word packets, which packets are
 * suppressed as silent, how timestamp / seq / packets / bytes / silent move -- is audio.c's own.
 */
#define _GNU_SOURCE 1
#include <complex.h>
#include <stdint.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include "radio.h"

unsigned char *hton_rtp(unsigned char *data, struct rtp_header *rtp){
  data[0] = 0xA5;
  data[1] = (unsigned char)(((rtp->marker & 1) << 7) | (rtp->type & 0x7f));
  data[2] = rtp->seq & 0xff;
  data[3] = rtp->seq >> 8;
  for(int i = 0; i < 4; i++){
    data[4 + i] = (rtp->timestamp >> (8 * i)) & 0xff;
    data[8 + i] = (rtp->ssrc >> (8 * i)) & 0xff;
  }
  return data + 12;
}
int setup_mcast(char const *target, struct sockaddr *sock, int output, int ttl, int offset){
  (void)target; (void)sock; (void)output; (void)ttl; (void)offset;
  return -1;
}

/* demod->output.rtp + output.silent, the state audio.c carries from call to call */
struct ref_audio_state {
  uint32_t ssrc;
  uint16_t seq;
  uint32_t timestamp;
  int silent;
  long long packets, bytes;
};

/* One call of send_mono_output (stereo = 0, nfloats samples) or send_stereo_output (stereo = 1, nfloats / 2 frames).
 * The datagrams come back as [len LE16][bytes] records in dst.  Returns the number of datagrams, -1 on error. */
int ref_audio_send(struct ref_audio_state *st, const float *audio, int nfloats, int stereo, unsigned char *dst, int cap, int *used){
  int sv[2];
  if(socketpair(AF_UNIX, SOCK_DGRAM, 0, sv) != 0)
    return -1;
  int const big = 1 << 22;
  setsockopt(sv[0], SOL_SOCKET, SO_SNDBUF, &big, sizeof big);
  struct demod demod;
  memset(&demod, 0, sizeof demod);
  demod.output.fd = sv[0];
  demod.output.rtp.ssrc = st->ssrc;
  demod.output.rtp.seq = st->seq;
  demod.output.rtp.timestamp = st->timestamp;
  demod.output.rtp.packets = st->packets;
  demod.output.rtp.bytes = st->bytes;
  demod.output.silent = st->silent;
  int n = 0, pos = 0, left = stereo ? nfloats / 2 : nfloats;
  /* audio.c loops over 480-word packets inside one call; a datagram socket holds only so many: feed the call in
   * pieces of at most 64 packets and drain in between (the packetiser's state carries over exactly as across calls) */
  while(left > 0){
    int const piece = left > 64 * 240 ? 64 * 240 : left;   /* multiple of both packet sizes (480 mono, 240 stereo frames) */
    if(stereo)
      send_stereo_output(&demod, audio, piece);
    else
      send_mono_output(&demod, audio, piece);
    audio += stereo ? 2 * piece : piece;
    left -= piece;
    for(;;){
      unsigned char pkt[2048];
      ssize_t const r = recv(sv[1], pkt, sizeof pkt, MSG_DONTWAIT);
      if(r < 0)
        break;
      if(pos + 2 + r > cap){
        close(sv[0]); close(sv[1]);
        return -1;
      }
      dst[pos] = r & 0xff;
      dst[pos + 1] = (r >> 8) & 0xff;
      memcpy(dst + pos + 2, pkt, (size_t)r);
      pos += 2 + (int)r;
      n++;
    }
  }
  close(sv[0]);
  close(sv[1]);
  st->seq = demod.output.rtp.seq;
  st->timestamp = demod.output.rtp.timestamp;
  st->packets = demod.output.rtp.packets;
  st->bytes = demod.output.rtp.bytes;
  st->silent = demod.output.silent;
  *used = pos;
  return n;
}
