/* oracle/_ref/libdftpav_ref_cr.so: the reference's own objects linked against THIS instead of libm for exp / log / pow / sin /
 * cos / sincos -- TEST INFRASTRUCTURE (oracle/Makefile.ref).
 *
 * The bits of those functions belong to the host, not to the reference: glibc's are not correctly rounded (its sincos differs
 * from its own sin for 1 argument in 1 600) and are IFUNC-dispatched by CPU.  A correctly rounded libm is the one libm every
 * host can agree on, and the one the reference-order device kernel implements (dftpav_amd/csrc/cr_trig.h, double-double).  Here
 * the correctly rounded values come from binary128 (libquadmath), exactly as oracle order 2 forms them (dftpav_oracle.c:
 * O_COS_SIN, o_exp, o_log, o_pow3) -- so this library is "the reference's compiled program on a correctly rounded libm", and
 * the device's reference order can be held bit-equal to the reference's OWN CODE on the configurations with libm calls in the
 * loop (gear shifts: cos / sin of the junction angles; moving obstacles: exp / log / x^3), not only to the restatement.
 * Hidden visibility: the references from traj_optimizer.o bind to these definitions at link time, whatever the process has
 * loaded.  atan2 (set-up only) and sqrt (correctly rounded by IEEE 754) stay libm's. */
#include <quadmath.h>

#define HIDDEN __attribute__((visibility("hidden")))

HIDDEN double exp(double x) { return (double)expq((__float128)x); }
HIDDEN double log(double x) { return (double)logq((__float128)x); }
HIDDEN double sin(double x) { return (double)sinq((__float128)x); }
HIDDEN double cos(double x) { return (double)cosq((__float128)x); }
HIDDEN void sincos(double x, double *s, double *c) {
  *s = (double)sinq((__float128)x);
  *c = (double)cosq((__float128)x);
}
/* the reference's only pow is the cube of a norm (poly_traj_utils.hpp:109); the product of three doubles in binary128 */
HIDDEN double pow(double x, double y) {
  if (y == 3.0) return (double)((__float128)x * (__float128)x * (__float128)x);
  return (double)powq((__float128)x, (__float128)y);
}
