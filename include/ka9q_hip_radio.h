/* ka9q_hip_radio.h -- the demodulator thread entry points of the reference's radio.h, served by libka9q_hip.so.
 *
 *   void *demod_fm(void *);  void *demod_am(void *);  void *demod_linear(void *);      radio.h:235-237
 *
 * are what radio.c's set_mode() starts through Demodtab[] (modes.c:25-29, radio.c:372) with a `struct demod *`.
 * The library's versions take the same argument, honour the same fields and call the same two hand-off functions
 * (send_mono_output / send_stereo_output, audio.c:32,82), which the host program keeps providing.  Everything
 * between -- the slave half of the pre-detection filter, compute_n0, the demodulator, the FM audio filter and PL
 * tone measurement, the carrier PLL -- runs on the GPU (a channel bank of one, fed from the master's device-resident
 * input window).
 *
 * A host that builds against the reference tree keeps using the reference's own radio.h: this header then only
 * adds nothing (its struct definitions are skipped when _RADIO_H is already defined).  Without the reference tree it
 * supplies `struct demod` field for field (radio.h:64-193, with struct status of sdr.h:18-27 and struct rtp_state of
 * multicast.h:41-50); tests/test_radio_layout.py checks every offset against the reference's header when that is
 * available.
 *
 * Behaviour a maintainer must know:
 *   - read per block: terminate, filter.{low,high,kaiser_beta,isb}, output.channels, shift.freq, opt.*, agc.*
 *     (opt.* and agc.{headroom,hangtime,recovery_rate} at thread start only, like the reference's prologues);
 *   - written: filter.out (a slave created with create_filter_output, its output.c refreshed every block),
 *     output.channels (FM, AM: 1), agc.gain, sig.{bb_power,n0,snr,foffset,pdeviation,plfreq,cphase,pll_lock,lock_timer};
 *     sig.if_power stays with proc_samples (radio.c:143-145);
 *   - terminate is read with acquire semantics and the thread deletes its slave (filter.out) once it has seen it set: set
 *     it after the last set_filter / noise_gain on that slave (radio.c:336-338 does, followed by pthread_join);
 *   - audio_master (FM) stays NULL: the post-detection filter lives on the device;
 *   - linear: filter.out->output.c holds the filter output before AGC (the reference scales it in place,
 *     linear.c:280); the audio handed to send_*_output is the scaled, shifted signal as in the reference.
 */
#ifndef KA9Q_HIP_RADIO_H
#define KA9Q_HIP_RADIO_H 1

#include <pthread.h>
#include <stdint.h>
#include <sys/socket.h>

#include "ka9q_hip_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifndef _RADIO_H /* the reference's own radio.h is not part of this translation unit */

/* sdr.h:18-27 */
struct status {
  long long timestamp;
  double frequency;
  uint32_t samprate;
  uint8_t lna_gain;
  uint8_t mixer_gain;
  uint8_t if_gain;
  uint8_t unused;
};

/* multicast.h:41-50 */
struct rtp_state {
  uint32_t ssrc;
  int init;
  uint16_t seq;
  uint32_t timestamp;
  long long packets;
  long long bytes;
  long long drops;
  long long dupes;
};

/* radio.h:18-22 */
enum demod_type {
  LINEAR_DEMOD = 0,
  AM_DEMOD,
  FM_DEMOD,
};

struct packet;      /* radio.h:54-60, only pointed to */

/* radio.h:64-193 */
struct demod {
  struct {
    int fd;
    int ctl_fd;
    char dest_address_text[256];
    struct sockaddr_storage source_address;
    struct sockaddr_storage dest_address;
    struct rtp_state rtp;
    long long samples;
    int samprate;
    pthread_cond_t qcond;
    pthread_mutex_t qmutex;
    struct packet *queue;
  } input;

  struct {
    struct status status;
    double calibration;
    float DC_i, DC_q;
    float sinphi;
    float imbalance;
    float min_IF;
    float max_IF;
    float gain_factor;
    pthread_mutex_t status_mutex;
    pthread_cond_t status_cond;
  } sdr;

  struct {
    int lock;
    double freq;
    double shift;
    int step;
    int item;
  } tune;

  pthread_t doppler_thread;
  char *doppler_command;
  struct osc doppler;
  struct osc second_LO;
  struct osc shift;

  struct notchfilter *nf;

  struct {
    struct filter_in *in;
    struct filter_out *out;
    int L;
    int M;
    int interpolate;
    int decimate;
    float low;
    float high;
    float kaiser_beta;
    float noise_bandwidth;
    int isb;
  } filter;

  pthread_t demod_thread;
  int terminate;

  enum demod_type demod_type;
  char mode[16];

  struct {
    int flat;
    int pll;
    int square;
    float loop_bw;
  } opt;

  struct {
    float headroom;
    float hangtime;
    float recovery_rate;
    float attack_rate;
    float gain;
  } agc;

  struct {
    float if_power;
    float bb_power;
    float n0;
    float snr;
    float foffset;
    float pdeviation;
    float cphase;
    float plfreq;
    float lock_timer;
    int pll_lock;
  } sig;

  struct filter_in *audio_master;

  struct {
    int samprate;
    int silent;
    struct rtp_state rtp;
    char dest_address_text[256];
    struct sockaddr_storage source_address;
    struct sockaddr_storage dest_address;
    int fd;
    int rtcp_fd;
    int status_fd;
    int channels;
  } output;
};

/* radio.h:235-241 */
void *demod_fm(void *);
void *demod_am(void *);
void *demod_linear(void *);
int send_mono_output(struct demod *, const float *, int);   /* provided by the host program (audio.c:82) */
int send_stereo_output(struct demod *, const float *, int); /* provided by the host program (audio.c:32) */

#endif /* _RADIO_H */

/* radio.c:383-425 on the device-resident spectrum of a master created by this library: what the demodulator threads
 * use, exported for hosts that want the estimate without pulling filter_in.fdomain through radio.c's own loop.
 * `low` / `high` in Hz as in demod->filter.  NaN on error. */
float kq_compat_compute_n0(struct filter_in *master, int samprate, float low, float high);

#ifdef __cplusplus
}
#endif
#endif
