/*
 * nka_hip_diag.h -- entry points that exist ONLY in the diagnostic build of the library
 * (nka_amd/libnka_hip_diag.so, compiled with -DNKA_DIAGNOSTIC by `make diag`; the phase stamps additionally
 * need -DNKA_SOLVE_STAMPS, `make stamps`).  The product, libnka_hip.so, does not export them: every choice they
 * override is made automatically there.  They serve in-process A/B measurements
 * (tools/ab_inproc.py: same allocations, same thermal state) and the tests that hold every kernel variant to
 * the same bits.  Same ABI otherwise: include nka_hip.h first.
 */
#ifndef NKA_HIP_DIAG_H
#define NKA_HIP_DIAG_H

#include "nka_hip_ext.h"
#include "nka_hip_vec.h"

#ifdef __cplusplus
extern "C" {
#endif

/* DIAGNOSTIC.  Every choice below is made automatically; the switches exist
 * for A/B measurements inside one process (same allocations, same thermal state) and for the tests
 * that hold every kernel variant to the same bits.  No environment variable selects a variant.
 * "pa_pipe" / "pb_pipe": -1 automatic (default); 0 = every load of a tile in flight
 * (k_dots / k_combine, any list length); 201..204 = rolling window (k_dots_win /
 * k_combine_win, instantiated for every width 1..32) with 1..4 blocks per CU.
 * "pb_tickets": how the blocks of the rolling-window PB
 * get their tiles: -1 automatic (default: tickets from 64 tiles per block), 0 =
 * static mapping (tile t -> block t mod G), 1, 2, 4, 8 = from that many global
 * ticket counters, so that the blocks advance as one compact front.
 * "pb_tile" (-1 automatic, 1, 2): 512- or 1024-element tiles for the shortest lists (<= 14 words
 * per element and tile), where double-width tiles let one ticket counter serve the pass.
 * "serial_solve" = 0/1: the scalar step on one wavefront (k_solve_rows) or as the reference's loops on one lane.
 * Results are bit-identical across variants.
 * "list_word" = 0/1: 0 ignores (and stops publishing) the list word -- the host's own count of the list length
 * only, the behaviour before round 4 (nka_hip_list_bound).
 * Round 5: "pb_reverse" = 0/1: the rolling-window PB walks its tiles from the END of the vectors, the reverse of PA's order
 * (the Infinity-Cache study, profiles/r05/ab_mall_reuse.txt).  "prime_pad": list lengths 23, 29, 31 are primes, the only
 * ring of their window kernels is the whole width (up to 311 VGPRs and scratch in PA): -1 automatic (= 1) both passes run them
 * at the next width with one dead ring slot; 0 = exact widths everywhere (profiles/r05/multipass.txt).  Same bits.
 * "fail_after_solve" = 1: the NEXT update returns NKA_HIP_EHIP right behind its enqueued scalar step, as a failing HIP call
 * there would: the handle must then be poisoned (every later call but destroy: NKA_HIP_ESTATE).
 * "chain_many" = -1/0/1: reference-order sums of the longest vectors by the whole device (k_chain_blocks / _predict / _apply): -1 automatic
 * (from 2^19 elements on), 0 never (one compute unit per sum: k_chain_sums), 1 wherever a vector has a full block.  Same bits.
 * "chain_walk" = 0/1: reference-order sums of long vectors (k_chain_sums) walk every block element after element instead of
 * taking whole blocks through the chain in integer arithmetic (chain_block_summary / chain_block_apply).  Same bits. */
int nka_hip_set_tuning(nka_hip_t a, const char *key, int32_t value);

/* Launch geometry knobs for tuning: blocks per CU of PA and PB (0 = automatic). */
int nka_hip_set_grid(nka_hip_t a, int32_t pa_blocks_per_cu, int32_t pb_blocks_per_cu);

/* Measurement aid: mean device time (ms) of the pure-read pass PA of the NEXT update,
 * launched `reps` times back to back (it only writes scratch: state unchanged). */
int nka_hip_debug_time_pa(nka_hip_t a, const double *f_dev, int32_t reps, float *ms_mean);

/* Test bench of the reference-order sums of long vectors: start + x[0]*y[0] + x[1]*y[1] + ... (one rounding per product and
 * per addition, in that order) over any two device arrays of n doubles, formed by the kernel the update uses (k_chain_sums, one
 * workgroup).  walk = 1: every block element after element; 0: whole blocks through the chain where that is provably the same
 * (chain_block_summary / _apply); walk | 2: the same sum through the many-compute-unit form (k_chain_blocks / k_chain_predict /
 * k_chain_apply: needs n >= 1024).  *ms (optional) = the kernels' device time.  The handle's state is not touched (scratch only). */
int nka_hip_debug_chain_sum(nka_hip_t a, const double *x_dev, const double *y_dev, int64_t n, double start, int32_t walk,
                            double *sum, float *ms);

/* Diagnostic builds of the library (-DNKA_SOLVE_STAMPS) stamp the phases of the
 * one-wavefront scalar step with s_memtime; this returns the 16 stamps of the most
 * recent update (zeros in a normal build).  tools/solve_phases.py prints them. */
int nka_hip_get_stamps(nka_hip_t a, double *out16);

/* Diagnostic A/B switch like nka_hip_set_tuning: "tickets" = -1 automatic, 0 static tile mapping,
 * 1, 2, 4, 8 ticket counters for the combine stage (update_many_keep / axpy_many_keep). */
int nka_hip_vec_set_tuning(nka_hip_vec_ws_t ws, const char *key, int32_t value);

#ifdef __cplusplus
}
#endif
#endif
