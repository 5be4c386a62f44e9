/* CPU check of the device's division by 1e-4 (historian_amd/csrc/hx_lse.h, div_by_1em4): the three-operation sequence
 * q0 = RN(a * 1e4), r = fma(-1e-4, q0, a), q = fma(r, 1e4, q0) against IEEE division a / 1e-4, which is what the reference
 * computes (src/logsumexp.h:53-57) - over random arguments of the table look-up's range, arguments at and next to every bin
 * boundary n * 1e-4, and the tiny remainders x - n * 1e-4 that go through the same division.  Prints the number of
 * arguments checked and of mismatches. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double seq(double a) {
  const double q0 = a * 1e4;
  const double r = fma(-1e-4, q0, a);
  return fma(r, 1e4, q0);
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next64(void) {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return rng_state;
}
static long long checked = 0, bad = 0;
static void check(double a) {
  const double want = a / 1e-4, got = seq(a);
  ++checked;
  if (memcmp(&want, &got, sizeof(double)) != 0) {
    if (bad < 5) fprintf(stderr, "mismatch at %.17g: %.17g vs %.17g\n", a, want, got);
    ++bad;
  }
}
int main(int argc, char** argv) {
  const long long n_random = argc > 1 ? atoll(argv[1]) : 20000000;
  /* every bin boundary of the table and its neighbours */
  for (int n = 0; n <= 100001; ++n) {
    double b = (double)n * 1e-4;
    check(b);
    double up = b, dn = b;
    for (int k = 0; k < 4; ++k) { up = nextafter(up, INFINITY); dn = nextafter(dn, 0.0); check(up); check(dn); }
  }
  /* random arguments of [0, 10) and random remainders of [0, 1e-4), uniform in value and uniform in bit pattern */
  for (long long k = 0; k < n_random; ++k) {
    const double u = (double)(next64() >> 11) * (1.0 / 9007199254740992.0);
    check(u * 10.0);
    check(u * 1e-4);
    uint64_t bits = next64() & 0x3FFFFFFFFFFFFFFFull;       /* positive doubles below 2.0 ... */
    double v;
    memcpy(&v, &bits, sizeof v);
    /* ... down to 1e-290: below that the exact residual a - 1e-4 * q0 underflows and the sequence is only faithful.  The
     * look-up's arguments are differences of log-probabilities and remainders of them: zero, or far above that. */
    if (v == v && v < 10.0 && v >= 1e-290) check(v);
  }
  printf("%lld %lld\n", checked, bad);
  return 0;
}
