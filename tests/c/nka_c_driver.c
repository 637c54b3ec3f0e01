/* Drives the C compatibility header through a short scenario; prints num_vec
 * and checksums per call (compared by tests with the oracle's C flavour).
 * With the argument `dp` the accelerator is created with a user dot product
 * (the dp argument of the reference's nka_init, src-C/...h:4): a sum in REVERSE
 * index order, so that the test can tell it was really used. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "nka_c_compat.h"

static long dp_calls = 0;
static double reverse_dot(int n, double *x, double *y) {
  double d = 0.0;
  dp_calls++;
  for (int i = n - 1; i >= 0; i--) d += x[i] * y[i];
  return d;
}

int main(int argc, char **argv) {
  const int with_dp = argc > 1 && strcmp(argv[1], "dp") == 0;
  FILE *raw = argc > 2 ? fopen(argv[2], "wb") : 0;   /* raw outputs for a bit-exact check */
  enum { N = 501, MVEC = 4, NCALLS = 12 };
  static double f[N];
  long long x = 1;
  NKA a = nka_init(N, MVEC, 0.05, with_dp ? reverse_dot : 0);
  if (nka_vec_len(a) != N || nka_max_vec(a) != MVEC || nka_vec_tol(a) != 0.05) return 2;
  for (int t = 1; t <= NCALLS; t++) {
    for (int i = 0; i < N; i++) {
      x = (1103515245LL * x + 12345LL) % 2147483648LL;
      f[i] = (double)x / 1073741824.0 - 1.0;
    }
    nka_accel_update(a, f);
    if (raw) fwrite(f, sizeof(double), N, raw);
    if (t == 6) nka_relax(a);
    if (t == 9) nka_restart(a);
    double s = 0.0, q = 0.0;
    for (int i = 0; i < N; i++) { s += f[i]; q += f[i] * f[i]; }
    printf("%3d%3d%25.16e%25.16e\n", t, nka_num_vec(a), s, sqrt(q));
  }
  nka_delete(a);
  if (raw) fclose(raw);
  if (with_dp && dp_calls == 0) return 3;
  return 0;
}
