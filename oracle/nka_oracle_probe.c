/*
 * oracle/nka_oracle_probe.c -- ERROR ATTRIBUTION instruments for the double restatement.
 * TEST INFRASTRUCTURE ONLY (see nka_oracle.h).
 *
 * The device path deviates from a literal transcription of src-F08/nka_type.F90:263-290, 371 in three ways,
 * all below the stated tolerance: its inner products (a) use fused multiply-adds,
 * (b) are summed in a blocked, tree-shaped order instead of sequentially, and (c) the Gram row of the
 * normalised vector is formed as fl(<d,w_k>/s) from raw sums (nka_oracle_set_gram_from_raw_sums in
 * nka_oracle.c).  The dot products below, installed through nka_oracle_set_dot_prod, switch (a) and (b)
 * on ONE AT A TIME in the restatement, so that its distance from the extended-precision trajectory
 * (nka_oracle_exact.c) can be attributed to each of them (tools/error_attribution.py).
 */
#include "nka_oracle.h"

#include <math.h>
#include <stdlib.h>

/* (a) alone: the reference's sequential order, every step a fused multiply-add */
double nka_oracle_dot_fma(void *ctx, int64_t n, const double *x, const double *y) {
  (void)ctx;
  double s = 0.0;
  for (int64_t i = 0; i < n; i++) s = fma(x[i], y[i], s);
  return s;
}

/* The summation ORDER of the device's pass PA (nka_amd/csrc/nka_kernels.hpp: k_dots_win, block_reduce_store,
 * k_finalize_dots) on 256 compute units: tiles of 512 elements dealt round-robin to G = min(256, n / 512)
 * blocks (at least one) of 256 threads; a thread accumulates its two elements of each of its block's tiles in
 * order; the ragged tail (n mod 512) goes to the last block, one element per thread and round; then the
 * wavefront butterfly (x[i] += x[i + off], off = 32 ... 1), the four wavefronts of a block in order, and the
 * per-block partial sums by 64 lanes striding over the blocks followed by the same butterfly. */
static double wave_sum(double *x) {
  for (int off = 32; off >= 1; off >>= 1)
    for (int l = 0; l < off; l++) x[l] += x[l + off];
  return x[0];
}

static double blocked_dot(int64_t n, const double *x, const double *y, int use_fma) {
  const int64_t ntile = n / 512;
  int G = (int)(ntile < 256 ? ntile : 256);
  if (G < 1) G = 1;
  double *partial = (double *)calloc((size_t)G, sizeof(double));
  double acc[256];
  for (int b = 0; b < G; b++) {
    for (int t = 0; t < 256; t++) acc[t] = 0.0;
    for (int64_t tile = b; tile < ntile; tile += G)
      for (int t = 0; t < 256; t++)
        for (int q = 0; q < 2; q++) {
          const int64_t e = tile * 512 + 2 * t + q;
          acc[t] = use_fma ? fma(x[e], y[e], acc[t]) : acc[t] + x[e] * y[e];
        }
    if (b == G - 1)
      for (int64_t i0 = ntile * 512; i0 < n; i0 += 256)
        for (int t = 0; t < 256 && i0 + t < n; t++) {
          const int64_t e = i0 + t;
          acc[t] = use_fma ? fma(x[e], y[e], acc[t]) : acc[t] + x[e] * y[e];
        }
    double r = 0.0;
    for (int w = 0; w < 4; w++) {
      const double ws = wave_sum(acc + 64 * w);
      r = (w == 0) ? ws : r + ws;
    }
    partial[b] = r;
  }
  double lanes[64];
  for (int l = 0; l < 64; l++) {
    double r = 0.0;
    for (int b = l; b < G; b += 64) r += partial[b];
    lanes[l] = r;
  }
  free(partial);
  return wave_sum(lanes);
}

/* (b) alone: the device's order, separately rounded products */
double nka_oracle_dot_blocked(void *ctx, int64_t n, const double *x, const double *y) {
  (void)ctx;
  return blocked_dot(n, x, y, 0);
}

/* (a) + (b): the inner product as the device forms it */
double nka_oracle_dot_device(void *ctx, int64_t n, const double *x, const double *y) {
  (void)ctx;
  return blocked_dot(n, x, y, 1);
}
