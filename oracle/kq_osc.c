/* kq_osc.c -- oracle restatement of the complex NCO (test infrastructure only).
 * Follows osc.c:14-59 and dsp.c:38-50 (csincospi = sincos(x*M_PI), dsp.h:49).
 * Pinned: tests compare it step for step with oracle/_ref/libref_osc.so, which is the
 * reference's own osc.c + dsp.c compiled unmodified.
 */
#define _GNU_SOURCE 1
#include <math.h>
#include <complex.h>
#include "kq_oracle.h"

/* unit phasor at angle pi*x -- dsp.c:38-42 with the non-Apple macro of dsp.h:49 */
static double complex unit_pi(double x){
  double s, c;
  sincos(x * M_PI, &s, &c);
  return c + s * _Complex_I;
}

/* osc.c:14-18: NaN components or squared magnitude below 0.9 => not initialised */
int kqo_is_phasor_init(double complex x){
  double const re = creal(x), im = cimag(x);
  if(isnan(re) || isnan(im))
    return 0;
  return (re * re + im * im) < 0.9 ? 0 : 1;
}

/* osc.c:22-36: phase is preserved once initialised; step phasors from frequency and sweep */
void kqo_set_osc(kqo_osc *o, double f, double r){
  if(!kqo_is_phasor_init(o->phasor)){
    o->phasor = 1;
    o->steps = 0;
  }
  o->freq = f;
  o->rate = r;
  o->phasor_step = unit_pi(2 * f);
  o->phasor_step_step = (r != 0) ? unit_pi(2 * r) : 1;
}

/* osc.c:53-59 */
void kqo_renorm_osc(kqo_osc *o){
  o->steps = 0;
  o->phasor /= cabs(o->phasor);
  if(o->rate != 0)
    o->phasor_step /= cabs(o->phasor_step);
}

/* osc.c:39-51: return the current phasor, then advance (frozen when freq == 0) */
double complex kqo_step_osc(kqo_osc *o){
  double complex const now = o->phasor;
  if(o->freq != 0){
    o->phasor *= o->phasor_step;
    if(o->rate != 0)
      o->phasor_step *= o->phasor_step_step;
  }
  o->steps++;
  if(o->steps == KQO_RENORM_RATE)
    kqo_renorm_osc(o);
  return now;
}
