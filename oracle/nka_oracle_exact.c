/*
 * oracle/nka_oracle_exact.c -- the restatement in EXTENDED precision (x87 long double: 64-bit significand,
 * unit roundoff 5.4e-20, 2048 times finer than double).  TEST INFRASTRUCTURE ONLY (see nka_oracle.h).
 *
 * Not a second implementation: it compiles oracle/nka_oracle.c -- the file pinned bit for bit to the compiled
 * reference -- a second time with real_t = long double, so the statements, their order and the list logic
 * (src-F08/nka_type.F90:249-457) are the pinned ones by construction; only the arithmetic is finer: every
 * stored vector, inner product (sequential, in extended precision), factor entry, coefficient and the combine.
 * Inputs are the caller's doubles exactly, the result is rounded to double once.  It serves as the "exact"
 * trajectory: tests/parity_util.py measures ||f_reference - f_exact|| and ||f_device - f_exact|| on the same
 * calls and holds the device to the reference's own distance from the truth.
 */
#define NKA_ORACLE_EXTENDED 1
#include "nka_oracle.c"
