/* Drives the C compatibility header through a short scenario; prints num_vec
 * and checksums per call (compared by tests with the oracle's C flavour). */
#include <math.h>
#include <stdio.h>

#include "nka_c_compat.h"

int main(void) {
  enum { N = 501, MVEC = 4, NCALLS = 12 };
  static double f[N];
  long long x = 1;
  NKA a = nka_init(N, MVEC, 0.05, 0);
  if (nka_vec_len(a) != N || nka_max_vec(a) != MVEC || nka_vec_tol(a) != 0.05) return 2;
  for (int t = 1; t <= NCALLS; t++) {
    for (int i = 0; i < N; i++) {
      x = (1103515245LL * x + 12345LL) % 2147483648LL;
      f[i] = (double)x / 1073741824.0 - 1.0;
    }
    nka_accel_update(a, f);
    if (t == 6) nka_relax(a);
    if (t == 9) nka_restart(a);
    double s = 0.0, q = 0.0;
    for (int i = 0; i < N; i++) { s += f[i]; q += f[i] * f[i]; }
    printf("%3d%3d%25.16e%25.16e\n", t, nka_num_vec(a), s, sqrt(q));
  }
  nka_delete(a);
  return 0;
}
