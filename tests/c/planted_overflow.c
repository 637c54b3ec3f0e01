/* planted_overflow.c -- proves that the sanitizer build of the checker (oracle/Makefile `asan`) is really instrumented.
 *   planted_overflow ok      a few accel_update calls of the C restatement on buffers of the right size: exit 0, no report
 *   planted_overflow plant   the same with f one element SHORT: the restatement reads/writes f[0 .. vlen) and AddressSanitizer
 *                            must stop the run with a heap-buffer-overflow report (tests/test_sanitizers_cpu.py asserts it)
 *   planted_overflow shift   a PLANTED undefined shift next to the calls: UBSan must stop the run
 * Test infrastructure; nothing here is product code. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nka_oracle.h"

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "ok";
  const int64_t n = 257;
  const int mvec = 4;
  const int64_t have = strcmp(mode, "plant") == 0 ? n - 1 : n;
  nka_oracle *a = nka_oracle_init(n, mvec, 0);
  double *f = (double *)malloc(sizeof(double) * (size_t)have);
  unsigned long long x = 12345;
  for (int t = 0; t < 9; t++) {
    for (int64_t i = 0; i < have; i++) {
      x = (1103515245ull * x + 12345ull) % 2147483648ull;      /* the integer LCG of the golden scenarios */
      f[i] = (double)x / 1073741824.0 - 1.0;
    }
    nka_oracle_accel_update(a, f);
  }
  if (strcmp(mode, "shift") == 0) {
    volatile int by = 40;
    volatile int one = 1;
    printf("%d\n", one << by);                                   /* PLANTED: shift exponent 40 is too large for int */
  }
  printf("num_vec %d f[0] %.17g\n", nka_oracle_num_vec(a), f[0]);
  free(f);
  nka_oracle_delete(a);
  return 0;
}
