/*
 * nka_hip.h -- CORE C ABI of the MI355X-native NKA accelerator (libnka_hip.so): what a caller needs to replace the
 * accelerator object of nncarlson/nka and its accel_update hot path, one entry per reference interface:
 *   F08  = /root/reference/src-F08/nka_type.F90          (type nka, F08:154-181)
 *   F08V = /root/reference/src-F08-vector/nka_type.F90
 *   C    = /root/reference/src-C/nonlinear_krylov_accelerator.{h,c}   (C .h:3-12)
 * Optional entry points live in two other headers and never change what the core returns:
 *   nka_hip_ext.h   state read-back for parity tests, the out-of-place update, hipGraph capture, the list word, timing,
 *                   the peer-to-peer exchange, the long note on the sum orders and the table of supported combinations
 *   nka_hip_vec.h   device hooks of the reference's abstract vector class (src-F08-vector/vector_class.F90)
 *
 * Plain C types only: an opaque handle, raw DEVICE pointers (memory of the HIP device the handle was created for), int64
 * lengths, a hipStream_t passed as void*.  The reference has no status codes (its precondition checks are ASSERTs,
 * F08:190-191,205,212,257-258); here every call returns 0 or a negative NKA_HIP_E* code and nka_hip_last_error() has the
 * text.  State-changing calls are ASYNCHRONOUS on the handle's stream (no host synchronisation inside accel_update); the
 * queries synchronise it.  Not thread-safe per handle; distinct handles are independent (also from different threads).  In
 * a multi-rank run every call is collective, like the reference (F08:58-64).  Slots are numbered as in Fortran: 1..mvec+1,
 * 0 = end of list.  There is no CPU path: without a HIP device nka_hip_create fails.
 */
#ifndef NKA_HIP_H
#define NKA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nka_hip_state *nka_hip_t;

enum {
  NKA_HIP_OK = 0,
  NKA_HIP_EINVAL = -1,  /* bad argument, or a combination that is not supported (the reference would ASSERT) */
  NKA_HIP_EHIP = -2,    /* a HIP runtime call failed                        */
  NKA_HIP_ENOMEM = -3,  /* device allocation failed                         */
  NKA_HIP_ECOMM = -4,   /* RCCL / user all-reduce failed                    */
  NKA_HIP_ESTATE = -5   /* device state failed the defined() invariants, or the call order is wrong */
};

/* Which of the reference's three roundings of the elementwise statements is mirrored:
 *   F08        w1-f ; x/s          ; (f - c*w) + c*v        F08:266,282-283,397
 *   F08_VECTOR (-1)*f+w ; (1/s)*x  ; ((-c)*w + c*v) + f     F08V:237,255-256,374
 *   C          w1-f ; x/s          ; f + c*(v - w)          C .c:299-300,317-320,423
 * The C flavour uses COMPACT storage: its combine only needs the difference v_k - w_k of a normalised pair, so that
 * difference is formed once and kept in the v array in place of v_k; f + c*(v-w) is then evaluated with bit-identical
 * operands while the combine pass reads one vector per pair instead of two (8n(9+L+k) bytes per update instead of
 * 8n(8+L+2k)).  NKA_HIP_FLAVOR_DEFAULT -- what every front end passes unless the caller names a flavour -- resolves to the
 * environment variable NKA_HIP_FLAVOR ("f08" | "f08vec" | "c") if set, else to NKA_HIP_FLAVOR_C; it differs from
 * NKA_HIP_FLAVOR_F08 only in the association of the combine (last bits).  nka_hip_flavor() reports what a handle runs. */
enum { NKA_HIP_FLAVOR_F08 = 0, NKA_HIP_FLAVOR_F08_VECTOR = 1, NKA_HIP_FLAVOR_C = 2 };
enum { NKA_HIP_FLAVOR_DEFAULT = -1 };

/* ---- NUMERICAL CONTRACT ----------------------------------------------------------------------------------------------
 * What nka_hip_accel_update returns, held against the COMPILED reference (oracle/_ref: /root/reference built here without
 * floating-point contraction) on the same call sequence.  err(x) = ||x - f_exact||_2 / ||f_in||_2, f_exact = the same calls
 * in extended precision (oracle/nka_oracle_exact.c, itself held to a 60-digit restatement, oracle/oracle_mp.py);
 * tests/parity_util.py is the executable form of every line below; the record behind the counts is
 * profiles/r06/soak_paired.txt (round 6) and profiles/r04, r05/fuzz_soak.txt.
 *  1. DECISIONS.  s == 0 (relax, F08:275), the capacity and dependence drops (F08:301-345), slot numbers, list order,
 *     free-list order and num_vec equal the reference's after EVERY call: compared with == in every test and every soak
 *     sequence (none has ever differed).  With reference-order sums that holds by construction; with the fast passes a
 *     pivot within rounding distance of vtol^2 could in principle fall the other way.
 *  2. REFERENCE-ORDER SUMS (NKA_HIP_SUMS_REFERENCE_ORDER; the default up to n = 64): the returned f -- and h, c, every
 *     stored vector -- carry the BITS of the reference flavour the handle runs, at any n, on one rank or sharded.
 *  3. FAST PASSES (the default beyond n = 64, both forms): sums in blocks with fma -- other last bits than the reference's
 *     sequential sums, and closer to f_exact than those from n ~ 1e3 up.  Over a call sequence
 *         TYPICAL  max err(f_device) <= max(base, F * max err(f_reference)),  base = 1e-12 (1e-10 for n >= 1e7),
 *                  F = 2 for n > 512, 4 within one tile of the kernels (n <= 512): within the stated tolerance wherever
 *                  the reference is, else within F times the reference's own worst distance from the truth on the same
 *                  calls (ill-conditioned sequences put the REFERENCE 1e-10 ... 1e-6 from it: no fixed figure can hold).
 *                  An empirical bar with a counted exceedance rate -- see item 6.
 *         HARD     the same with F = 8 for n > 512 and F = 32, base 1e-11 for n <= 512 has NEVER been exceeded by any
 *                  recorded sequence in either sum mode (item 6); every test also stops a call that exceeds it.
 *     DIRECTLY against the reference: wherever err(f_reference) <= base / 2, ||f_device - f_reference|| / ||f_in|| <= 2 base
 *     is asserted; at n = 2e7, m = 20 <= 1e-10 on every call, unscaled (tests/test_hip_fullsize.py).
 *  4. FLAVOURS.  The default flavour evaluates the combine as the C reference does, f + c*(v - w), against src-F08's
 *     (f - c*w) + c*v: last-bit differences, inside every bound above; NKA_HIP_FLAVOR_F08 gives the src-F08 statement.
 *  5. Sharded runs return the same bits on every rank (one reduction result, added in one order), and -- fast passes --
 *     bits that depend on the number of ranks like any blocked sum; the bounds of 3 hold unchanged.
 *  6. THE RECORD (profiles/r06/soak_paired.txt: the two fast modes on the SAME 8 104 random call sequences, 23 104 rank
 *     records each; profiles/r04, r05/fuzz_soak.txt: about 75 000 records of the raw-sum form).  Beyond the TYPICAL bar:
 *                               n <= 16     17..512    513..2048    > 2048      (records of 7 969 / 3 110 / 3 679 / 8 346)
 *         default (rounded)        14           0           0           1       largest ratio beyond one tile 2.26 x
 *         NKA_HIP_SUMS_BLOCKED     28           0           5           2       largest ratio beyond one tile 3.07 x
 *     -- 7 records beyond one tile against 1: why the default was changed in round 6.  The HARD line was exceeded by no
 *     record of either mode (largest ratios: 27.9 within one tile at err > 1e-11; 5.5 x beyond, abstract-vector flavour,
 *     round 5).  Exceedances beyond one tile are fixtures (tests/golden/soak_cases.json) replayed by the suite.
 */

/* ---- lifecycle --------------------------------------------------------- */

/* Replaces  call a%init(vlen, mvec)  (F08:185-200)  /  nka_init(vlen, mvec, vtol, dp)  (C .h:4, .c:211-258).
 * vlen_local is THIS rank's slice length (>= 0), mvec > 0, vtol > 0 (the Fortran default is 0.01, F08:160).  `device` is
 * the HIP device ordinal; `stream` a hipStream_t (NULL = HIP's default stream).  Allocates 2*(mvec+1) slot vectors on the
 * device.  Any mvec, like the reference: the scalar step runs on one wavefront up to mvec = 62, on one lane in LDS up to
 * 140, beyond that on one lane in global memory (correct and slow; practical subspaces are 5..20 vectors). */
int nka_hip_create(nka_hip_t *out, int64_t vlen_local, int32_t mvec, double vtol,
                   int32_t flavor, int32_t device, void *stream);
/* Replaces nka_delete (C .h:5, .c:261-282) / automatic deallocation (F08). */
int nka_hip_destroy(nka_hip_t a);
/* Replaces the intrinsic assignment  b = a  of the reference's type, a DEEP copy (allocatable components, F08:154-168):
 * *out becomes an independent accelerator with the same vectors, lists, factor, flags, vtol, flavour, device and stream.
 * User hooks (set_allreduce, set_host_dot) are carried over like the procedure pointer dp (F08:161); the built-in RCCL
 * communicator and peer-to-peer mailboxes are NOT (they belong to src): the copy refuses to update until it has its own. */
int nka_hip_clone(nka_hip_t src, nka_hip_t *out);
/* Rebind the handle to another hipStream_t (NULL = default stream); work already enqueued stays ordered before. */
int nka_hip_set_stream(nka_hip_t a, void *stream);

/* ---- the hot path ------------------------------------------------------ */

/* Replaces  call a%accel_update(f)  (F08:249-419; F08V:219-397; C .h:6, .c:285-447).  f_dev: vlen_local doubles in
 * device memory, updated in place.  The object keeps copies, never a reference (F08:361,404).  If the all-reduce of a
 * sharded handle fails the call returns NKA_HIP_ECOMM with the update NOT done (only scratch sums were written). */
int nka_hip_accel_update(nka_hip_t a, double *f_dev);
/* Host-array compatibility entry (the reference signature takes host memory, F08:252): H2D copy, update, D2H copy,
 * stream synchronised on return. */
int nka_hip_accel_update_host(nka_hip_t a, double *f_host);
/* Replaces  call a%restart()  (F08:422-436; C .h:7).  */
int nka_hip_restart(nka_hip_t a);
/* Replaces  call a%relax()    (F08:439-457; C .h:8).  */
int nka_hip_relax(nka_hip_t a);
/* Replaces  call a%set_vec_tol(vtol)  (F08:202-207). */
int nka_hip_set_vec_tol(nka_hip_t a, double vtol);

/* How the inner products are summed (part of the numerical contract; the long note is in nka_hip_ext.h):
 *   NKA_HIP_SUMS_AUTO (default)   reference order where it costs nothing (one rank, n <= 64), otherwise _BLOCKED_ROUNDED
 *   NKA_HIP_SUMS_REFERENCE_ORDER  every sum as the reference forms it: accel_update returns the reference's BITS at any n,
 *                                 sharded too (needs nka_hip_set_shard); validation speed beyond a few thousand elements
 *   NKA_HIP_SUMS_BLOCKED_ROUNDED  the fast passes with the norm first (a short pass of its own, a second exchange when
 *                                 sharded), then every other sum in one pure-read pass on the ROUNDED w1' = fl(d/s): the
 *                                 Gram row is the inner product of the STORED vector, as the reference defines it
 *                                 (F08:282-290).  What every front end runs since round 6 (contract item 6)
 *   NKA_HIP_SUMS_BLOCKED          opt-in FAST mode: ONE pure-read pass forms every sum of an update, the Gram row of the
 *                                 normalised difference is taken from raw sums, fl(<d,w_k>/s): 2 words per element and one
 *                                 exchange less (5-9 % faster), the same typical distance from the truth, a heavier tail
 * A user dot product (nka_hip_set_host_dot) overrides them all.  Can be changed between updates.  The environment variable
 * NKA_HIP_SUMS = auto | rounded | blocked | reference sets what a new handle starts with (for callers that cannot call
 * this function: the reference's own programs relinked against the front ends). */
enum { NKA_HIP_SUMS_AUTO = 0, NKA_HIP_SUMS_REFERENCE_ORDER = 1, NKA_HIP_SUMS_BLOCKED = 2, NKA_HIP_SUMS_BLOCKED_ROUNDED = 3 };
int nka_hip_set_sum_order(nka_hip_t a, int32_t order);

/* ---- queries (synchronise the stream) ---------------------------------- */

int nka_hip_num_vec(nka_hip_t a);      /* F08:221-231, C .h:9  ; <0 on error */
int nka_hip_max_vec(nka_hip_t a);      /* F08:233-236, C .h:10 */
int64_t nka_hip_vec_len(nka_hip_t a);  /* F08:238-241, C .h:11 (local length) */
double nka_hip_vec_tol(nka_hip_t a);   /* F08:243-246, C .h:12 ; -1 (and last_error) on a NULL handle */
int nka_hip_defined(nka_hip_t a);      /* F08:460-524 ; 1 = well defined */
int nka_hip_flavor(nka_hip_t a);       /* NKA_HIP_FLAVOR_* this handle runs (DEFAULT resolved) ; <0 on error */
const char *nka_hip_last_error(void);  /* text of the last failure on this thread */
/* "gfx950"-style name of the device the handle runs on, CU count. */
int nka_hip_device_info(nka_hip_t a, char *name64, int32_t *num_cu);

/* ---- the user's dot product --------------------------------------------- */

/* Source compatibility with  call a%set_dot_prod(dot_prod)  (F08:209-219), the dp argument of nka_init (C .h:4,
 * .c:196,227-231) and the optional dp dummy of the F95 nka_accel_update (src-F95/nka_type.F90:278-291): a user dot
 * product over HOST arrays that returns the GLOBAL dot product (F08:58-64; no hook is applied on top).  With fn installed
 * an update makes EXACTLY the reference's calls, in its order, on its operands (host copies of this rank's slices):
 * fn(d,d) with d = w1 - f; if s != 0 the Gram row fn(w1', w_k) for every older list entry in list order (F08:286-290);
 * then, after the device has taken the drop decisions, fn(f, w_j) for j = first ... last (F08:371) -- while the scalar
 * step, the combine and the ring stores stay on the device.  Results are bit-identical to the reference with the same dp.
 * 2+L vectors cross PCIe per update and the call synchronises: a compatibility path.  fn = NULL restores the device sums. */
typedef double (*nka_hip_host_dot_fn)(void *ctx, int64_t n, const double *x, const double *y);
int nka_hip_set_host_dot(nka_hip_t a, nka_hip_host_dot_fn fn, void *ctx);

/* ---- distribution (contiguous n-slices, one handle per rank; F08:58-64) --- */

/* The device form of  set_dot_prod  for a sharded vector: the local partial sums already live on the device, so the hook
 * is the global SUM of `count` doubles at device address `buf`, in place, enqueued on `stream` (hipStream_t).  Must return
 * 0 on success and give bit-identical results on every rank.  Called ONCE per accel_update (count 2+2*mvec: the norm and
 * both Gram rows; twice with NKA_HIP_SUMS_BLOCKED_ROUNDED).  NULL restores the single-rank default (no reduction). */
typedef int (*nka_hip_allreduce_fn)(void *ctx, double *buf, int32_t count, void *stream);
int nka_hip_set_allreduce(nka_hip_t a, nka_hip_allreduce_fn fn, void *ctx);
/* Built-in hook: RCCL all-reduces over xGMI on the handle's stream (per update: 8 B + 328 B at mvec = 20; one of 336 B in the
 * fast sum mode).  nka_hip_comm_unique_id fills 128 bytes (an
 * ncclUniqueId) on one rank; the caller broadcasts it by any means; every rank then calls comm_init_rank (collective,
 * blocking).  RCCL is bound at first use (dlopen): the librccl.so.1 already mapped into the process if there is one, else
 * the ROCm installation's; nka_hip_comm_library reports the file.  comm_info: what the communicator itself reports
 * (*nranks = 0, *rank = -1 without one).  nka_hip_comm_destroy drops the communicator and the hook. */
int nka_hip_comm_unique_id(void *id128);
int nka_hip_comm_init_rank(nka_hip_t a, const void *id128, int32_t nranks, int32_t rank);
int nka_hip_comm_destroy(nka_hip_t a);
int nka_hip_comm_library(char *path, int32_t len);
int nka_hip_comm_info(nka_hip_t a, int32_t *nranks, int32_t *rank);
/* Position of this rank's slice in the global vector: slice `rank` of `nranks`, slices laid out in rank order (what
 * nka_amd/dist.py and every front end of this build use).  Only the sharded reference-order sums read it;
 * nka_hip_comm_init_rank sets it by itself. */
int nka_hip_set_shard(nka_hip_t a, int32_t rank, int32_t nranks);
/* Run the installed all-reduce hook once on `count` doubles at device address buf_dev, on the handle's stream (no-op
 * without a hook): lets a launcher prove the communicator before the first update. */
int nka_hip_allreduce_now(nka_hip_t a, double *buf_dev, int32_t count);
/* 64-bit FNV-1a digest of the device-resident scalar state (flags, lists, h, c, the reduced sums of the last update).  In
 * a sharded run that state is replicated: every rank must report the same digest after the same call sequence -- the
 * check SURVEY.md 8(e) asks for, since s == 0, the drops and the slot choices are taken independently per rank. */
int nka_hip_state_digest(nka_hip_t a, uint64_t *digest);

#ifdef __cplusplus
}
#endif
#endif /* NKA_HIP_H */
