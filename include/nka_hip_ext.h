/*
 * nka_hip_ext.h -- the OPTIONAL entry points of libnka_hip.so, beside the core of include/nka_hip.h.
 *
 * Nothing here is needed to replace the reference (the core mirrors its surface one to one) and nothing here changes what
 * the core returns.  Four groups:
 *   validation    nka_hip_get_state / get_reductions / get_w / get_v (the private list state of F08:155-168 for parity tests);
 *                 the long note on the sum orders of nka_hip_set_sum_order (declared in the core)
 *   performance   nka_hip_accel_update_swap (buffers change hands: read its ownership rules), nka_hip_list_bound,
 *                 nka_hip_capture_safe (hipGraph), nka_hip_set_timing / get_timing / set_timing_stride
 *   distribution  nka_hip_p2p_* : the peer-to-peer exchange of the sums (opt-in prototype; RCCL -- core -- is the default)
 *   housekeeping  nka_hip_invalidate_pointer_cache
 * SUPPORTED combinations of (sum order x transport); DESIGN.md section 6 names the test that holds each cell, and a
 * combination outside the table returns NKA_HIP_EINVAL / NKA_HIP_ESTATE instead of running:
 *   sum order \ transport        none   set_allreduce hook   RCCL (comm_init_rank)   p2p mailboxes
 *   AUTO = BLOCKED_ROUNDED        yes    yes (2 exchanges)    yes (2 exchanges)       yes (opt-in; send-and-gather kernel, twice)
 *   BLOCKED (opt-in fast mode)    yes    yes (1 exchange)     yes (1 exchange)        yes (opt-in; fused into the final sums)
 *   REFERENCE_ORDER               yes    validation only: 2N exchanges per update through any hook that sums; mvec <= 250
 *   user dot product (core)       yes    the user's dp IS the global reduction: hooks are not applied on top
 *   out-of-place entry            every row above except the user dot product; not capturable into a graph
 * F08 = /root/reference/src-F08/nka_type.F90, C = /root/reference/src-C/nonlinear_krylov_accelerator.{h,c}.
 */
#ifndef NKA_HIP_EXT_H
#define NKA_HIP_EXT_H

#include "nka_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- validation ----------------------------------------------------------------------------------------------------- */

/* HOW THE INNER PRODUCTS ARE SUMMED -- the long form of the core header's note on nka_hip_set_sum_order.  The reference's sums are sequential (its default dot product: C .c:200-208;
 * `dot_product` in F08:216-219); the fast passes sum in blocks, fused, and take the Gram row of the normalised
 * difference from raw sums -- closer to the exact result than the reference from n ~ 1e3 up, but other bits.
 *   NKA_HIP_SUMS_REFERENCE_ORDER  every sum of an update exactly as the reference forms it -- the norm first, then
 *       <w1',w_k>, <f,w_k>, <f,w1'> on the ROUNDED w1' = d/s, element after element, one rounding per product and per
 *       addition -- on one workgroup.  Everything else of an update being bit-exact given its sums, accel_update then
 *       returns the bits of the reference flavour the handle runs (compiled without contraction, as oracle/Makefile
 *       does) at ANY n: a validation mode for callers moving over from the reference.  Cost: two chains of n dependent
 *       roundings per update -- on par with the fast passes up to n = 64, +12-18 us at n = 512; beyond one chunk (~ 700
 *       elements at mvec = 20) the sums go through their chains 1024 products at a time wherever that is provably the
 *       element-after-element result -- one compute unit per sum (k_chain_sums), from 2^19 elements on the block summaries of
 *       every sum by the whole device and one wavefront per sum to apply them (k_chain_blocks / k_chain_apply; 25 bytes of
 *       scratch per block and sum, allocated at the first such update): 0.16 ms at n = 1e4, 5 ms at n = 1e6, 0.13 s at
 *       n = 1e8, mvec = 20 on uniform random vectors (0.02 s on correlated ones) -- 22 x the compiled reference on its
 *       core, every output torch.equal to it in
 *       the same bench run (profiles/r05/reference_order_chain.txt).  SHARDED (an all-reduce installed): the reference's sum over
 *       the global vector is one chain of additions through the slices in rank order, so the ranks take turns -- rank r
 *       continues the running sums of ranks 0..r-1, the others contribute zeros, and the installed hook (any hook that
 *       sums) hands the prefix on: N rounds for the norm (w1' = d/s needs the GLOBAL s before it can be rounded), N for the
 *       rows, 2N small exchanges per update.  An N-rank run then returns the bits of the SINGLE-rank compiled reference.
 *       The handle must know where its slice lies: nka_hip_set_shard (nka_hip_comm_init_rank does it), else NKA_HIP_ESTATE.
 *   NKA_HIP_SUMS_AUTO (default)   reference order where it costs nothing -- a single rank and n <= 64 (every golden
 *       scenario of the reference among them) -- NKA_HIP_SUMS_BLOCKED_ROUNDED otherwise (round 6).
 *   NKA_HIP_SUMS_BLOCKED_ROUNDED  the fast passes with the norm first, in a short pass of its own (two streams: 51 instead
 *       of 49 words per element, one more exchange when sharded), and PA then on the ROUNDED w1' = fl(d/s) -- the vector that is
 *       stored: <w1',w_k> and <f,w1'> are then inner products of the stored vectors, as the reference defines them (F08:283-290,
 *       371), instead of fl(<d,w_k>/s).  That removes the one deviation of the fast passes that is not "a more accurate sum":
 *       what remains is the blocked order and the fma (profiles/r05/rounded_gram_row.txt).  Made the default in round 6: on the
 *       same 8 104 soak sequences 1 record beyond one tile ended beyond the typical bar against 7 (profiles/r06/soak_paired.txt).
 *   NKA_HIP_SUMS_BLOCKED          the single-pass FAST mode at every n (the default of rounds 1-5): ONE pure-read pass forms every
 *       sum of an update, the Gram row from raw sums; 5-9 % faster, one exchange per update; with the peer-to-peer mailboxes
 *       the final sums go straight into them and the scalar step gathers (no communication kernel at all).
 * A user dot product (nka_hip_set_host_dot) overrides them all.  Can be changed between updates.  REFERENCE_ORDER is
 * offered up to mvec = 250 (NKA_HIP_EINVAL beyond).
 * (The constants and nka_hip_set_sum_order / nka_hip_set_shard themselves are declared in the core header: the sum order is
 * part of the numerical contract.) */

/* List / factor state for parity tests (the reference keeps these private,
 * F08:155-168).  next, prev: mvec+1 ints (entry k-1 is slot k); h: (mvec+1)^2
 * doubles, column-major h(i,j) = h[(i-1)+(j-1)*(mvec+1)]; c: mvec+1 doubles,
 * the coefficients of the last update by slot.  Any pointer may be NULL. */
int nka_hip_get_state(nka_hip_t a, int32_t *subspace, int32_t *pending, int32_t *first,
                      int32_t *last, int32_t *free_, int32_t *next, int32_t *prev,
                      double *h, double *c);
/* The reduced inner products of the most recent update as the device solve saw
 * them, with d = w1 - f the new (not yet normalised) difference:
 * red[0] = <d,d>, red[1] = <f,d>, red[2+p] = <d,w_p>, red[2+mvec+p] = <f,w_p>
 * for the p-th older list entry (2+2*mvec doubles).  The solve divides the d
 * rows by s = sqrt(red[0]).  With these a CPU restatement of the scalar step
 * can be checked bit for bit.  With reference-order sums (nka_hip_set_sum_order) red[1] and red[2+p] are the sums on the
 * NORMALISED difference, <f,w1'> and <w1',w_p>, and the solve takes them as they are. */
int nka_hip_get_reductions(nka_hip_t a, double *red_out);
/* Copy stored vector w(:,slot) / v(:,slot) (1-based slot) to host memory. */
int nka_hip_get_w(nka_hip_t a, int32_t slot, double *host_out);
int nka_hip_get_v(nka_hip_t a, int32_t slot, double *host_out);

/* ---- performance ---------------------------------------------------------------------------------------------------- */

/* OUT-OF-PLACE form of accel_update, opt-in: two of PB's five store streams less (8n(7+L+k) bytes per update with
 * compact storage instead of 8n(9+L+k); 8n(6+L+2k) against 8n(8+L+2k) in the src-F08 rounding).  The reference keeps
 * COPIES of f_in and f_out (F08:361, 404) and returns f_out in the caller's array; here the buffers themselves change
 * hands instead of being copied:
 *   in   *f_io  : device buffer with f (vlen_local doubles, 16-byte aligned).  The library KEEPS it -- it becomes the
 *                 storage of w of the new pair, which is f_in itself -- until the handle is destroyed: do not write it,
 *                 free it only after nka_hip_destroy.
 *   out  *f_io  : a free device buffer of the library (>= vlen_local doubles, contents undefined) for the caller's NEXT
 *                 input; it dies with the handle.
 *   out  *f_acc : the accelerated f -- the v of the new pair, stored once -- to be READ only (solution update, next
 *                 residual), valid until the next accel_update* / restart / destroy on this handle.  Handing it to either
 *                 update entry as the next f is refused (NKA_HIP_EINVAL): it is the stored v of the pending pair.
 * Same arithmetic, same bits, same state as nka_hip_accel_update on the same inputs; the two entries can be mixed.  The
 * buffers an update displaces are known on the device only; they reach the host with the list word's record (no
 * synchronisation if the caller has synchronised since the previous out-of-place update, else this call waits for the
 * stream).  Not with nka_hip_set_host_dot; not capturable into a graph. */
int nka_hip_accel_update_swap(nka_hip_t a, double **f_io, const double **f_acc);

/* hipGraph capture.  nka_hip_accel_update only enqueues kernels (no allocation,
 * no synchronisation), so a caller may capture it into a graph and replay it.
 * A replay re-issues the kernel instances chosen at capture time; they stay
 * valid as long as the unroll widths chosen then cover the list, which is
 * guaranteed once nka_hip_capture_safe() returns 1: a pair is pending and the
 * host-side bound on the list length (nka_hip_list_bound) is mvec+1 -- from the
 * mvec+1-th update after init/restart on, unless the device has reported a list made
 * shorter by dependence drops.  restart()/relax() end that state.  A handle whose stream
 * was seen capturing stops using (and publishing) the list word for good: replays change
 * the list behind it.  Never 1 in the
 * debug mode (NKA_HIP_DEBUG=1 reads the state back after every update), with a user dot
 * product (nka_hip_set_host_dot: it runs on the host) or with a caller's all-reduce hook
 * (nka_hip_set_allreduce: a host callback a replay would not call again; the built-in
 * RCCL hook only enqueues on the stream and can be captured). */
int nka_hip_capture_safe(nka_hip_t a);

/* Upper bound on the list length (pending pair included) at the entry of the next update, as the HOST knows it
 * without synchronising: its own count (+1 per update up to mvec+1, -1 per relax, 0 after restart) tightened by the
 * LIST WORD -- a 64-bit word in pinned host memory that the combine pass of every update overwrites with (update
 * number, list length at its exit).  After a dependence drop (F08:326-345) the list is shorter than the host's
 * count; PA and PB are launched at the width of this bound, so a caller that synchronises once per iteration (every
 * solver reads its residual norm) runs both passes at exactly the list length, never at the padded full width.
 * A caller that never synchronises gets the plain count.  Results do not depend on the width (same bits). */
int nka_hip_list_bound(nka_hip_t a);

/* Per-phase device times from HIP events recorded on the handle's stream.
 * nka_hip_set_timing(a, capacity) keeps the events of the last `capacity`
 * updates in a ring (0 switches timing off); recording never synchronises.
 * nka_hip_get_timing(a, back, ms) synchronises and returns, for the update
 * `back` calls ago (0 = most recent):  ms[0] = PA k_dots (with its final sums
 * and the all-reduce), ms[1] = k_solve, ms[2] = PB k_combine, ms[3] = whole
 * update, first kernel start -> last kernel end. */
int nka_hip_set_timing(nka_hip_t a, int32_t capacity);
int nka_hip_get_timing(nka_hip_t a, int32_t back, float ms[4]);
/* With timing on, record the events of every stride-th update only (1..1024; the first update after the call is a
 * recorded one): four event records widen the kernel boundaries of an update by ~15 us, which matters below
 * n ~ 1e7.  get_timing then counts recorded updates. */
int nka_hip_set_timing_stride(nka_hip_t a, int32_t stride);

/* ---- distribution: the peer-to-peer exchange ------------------------------------------------------------------------- */

/* PEER-TO-PEER EXCHANGE (opt-in, one node): the sums of an update without a communication kernel -- one more way to supply
 * the global reduction the reference leaves to its caller (F08:58-64, set_dot_prod F08:209-214).  Every rank owns a
 * mailbox in fine-grained device memory which its peers map through hipIpc; the final-sums kernel of an update writes
 * each sum straight into every rank's mailbox (value, then the exchange number released at system scope) and the scalar
 * step starts by waiting for the N rows and adding them IN RANK ORDER -- the same additions in the same order on every
 * rank, hence the same bits; two kernel boundaries fewer than with an all-reduce kernel in between.  Set-up, collective:
 *   nka_hip_p2p_export(a, nranks, handle64)   allocate this rank's mailbox, fill 64 bytes (a hipIpcMemHandle_t);
 *   the caller gathers the nranks handles in rank order by any means (as it broadcasts the RCCL unique id);
 *   nka_hip_p2p_attach(a, handles, nranks, rank)   map the peers' mailboxes, install the exchange as the reduction
 *                                                   (it also serves nka_hip_allreduce_now and the reference-order chain,
 *                                                   as one small send-and-gather kernel).
 * One process per GPU (hipIpc does not map a handle into the process that exported it).  A wait for a peer is BOUNDED
 * (NKA_HIP_P2P_TIMEOUT_MS, default 10000): if a rank's sums do not arrive the gather stores NaNs, raises a status word and
 * lets the grid drain; the next synchronising query (num_vec, get_state, state_digest ...) returns NKA_HIP_ECOMM.
 * Capturable into a graph (the exchange number lives on the device).  nka_hip_p2p_detach, collective too (a peer must not
 * write into a mailbox that has been freed: synchronise all ranks first), drops it; nka_hip_destroy calls it.
 * Where hipIpc is refused (export or attach returns NKA_HIP_ECOMM) the caller falls back to the RCCL hook:
 * nka_amd/dist.py attach_allreduce(ladder=("p2p", "rccl", ...)) decides that collectively.
 * STATUS: a prototype, proven with ranks sharing one GPU (tests/test_p2p_exchange.py); over xGMI it needs a measured win
 * over RCCL's 336-byte all-reduce before it is preferred. */
int nka_hip_p2p_export(nka_hip_t a, int32_t nranks, void *handle64);
int nka_hip_p2p_attach(nka_hip_t a, const void *handles, int32_t nranks, int32_t rank);
int nka_hip_p2p_detach(nka_hip_t a);
/* The same exchange between several handles of ONE process (one per slice, each on its own stream, driven by host threads):
 * after nka_hip_p2p_export on every handle, nka_hip_p2p_mailbox returns the device address of a handle's mailbox and
 * nka_hip_p2p_attach_local takes the nranks addresses in rank order (entry `rank` must be the handle's own).  Nothing is
 * mapped, so detach frees only the handle's own mailbox: synchronise every handle before the first detach.  The scalar
 * step of a slice WAITS on the device for the sums of the others: every handle's stream must own a hardware queue
 * (GPU_MAX_HW_QUEUES >= the number of slices on one device, in the environment before HIP starts; the default of 4 makes
 * streams share queues and a wait then sits in front of the kernel it waits for until the timeout): attach_local returns
 * NKA_HIP_ESTATE when more slices share the handle's device than GPU_MAX_HW_QUEUES (default 4) allows. */
int nka_hip_p2p_mailbox(nka_hip_t a, void **mailbox);
int nka_hip_p2p_attach_local(nka_hip_t a, void *const *mailboxes, int32_t nranks, int32_t rank);

/* ---- housekeeping --------------------------------------------------------------------------------------------------- */

/* Device pointers that cross this ABI are checked against their allocation before any launch
 * (a kernel reading past a buffer faults the GPU): memory handed out by nka_hip_vec_alloc from a
 * registry of live allocations, any other pointer with hipMemGetAddressRange on every call.
 * NKA_HIP_CHECK_POINTERS=cached (opt-in) remembers foreign spans that passed for 100 ms per
 * thread; a caller that frees such a buffer itself calls this to drop what was remembered.
 * NKA_HIP_CHECK_POINTERS=0 switches the checks off. */
void nka_hip_invalidate_pointer_cache(void);

#ifdef __cplusplus
}
#endif
#endif /* NKA_HIP_EXT_H */
