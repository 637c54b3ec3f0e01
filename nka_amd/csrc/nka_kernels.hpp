// nka_kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the NKA accel_update hot path.
//
// One update is TWO HBM-streaming passes around one scalar step:
//   PA  k_dots    : pure-read pass.  While w1 (the raw previous f), f and the L
//                   stored w's stream past once it accumulates ALL the inner
//                   products the update needs:  sum d^2 with d = w1 - f
//                   (F08:266-267), <f,d>, <d,w_k> (F08:286-290) and <f,w_k>
//                   (F08:371).  The Gram row of the normalised w1' = d/s follows
//                   by one scalar division per entry in k_solve.  Reads (2+L)n
//                   words, writes nothing -- a kernel without stores streams at
//                   ~6.8 TB/s on MI355X, one with any store at ~5.3 TB/s
//                   (tools/hbm_probe.hip), so every store of the update is
//                   concentrated in PB.
//   --  k_finalize_dots (fixed-order sums => bitwise reproducible), [one
//       all-reduce of 2+2*mvec doubles], k_solve (s, s == 0 -> relax F08:275,
//       Cholesky with drops F08:295-351, both substitutions F08:369-392, list
//       surgery) on ONE wavefront.
//   PB  k_combine : w1' = (w1-f)/s and v1' = v1/s (F08:282-283) formed in
//                   registers and stored, f <- f - sum c_k w_k + sum c_k v_k in
//                   list order (F08:395-399), and the two ring stores w_new = f_in
//                   (F08:361), v_new = f_out (F08:404).  Reads (1+2k)n, writes 5n;
//                   in the C flavour's compact storage (see k_combine) (2+k)n.
// That moves 8n(8+L+2k) bytes per update -- 8n(9+L+k) compact -- against the
// 8n(11+L+2k) of the three-pass schedule of SURVEY.md 8(d), with ONE
// synchronisation point.
// Round 5: lists longer than 32 run PA and PB as balanced passes of the same window kernels (`base` into the plans);
// with several ranks the final sums can go straight into every rank's mailbox and the scalar step gathers them (struct
// P2P: no communication kernel); the reference-order pass k_dots_ordered continues the running sums from rank to rank.
//
// Everything is fp64 and bandwidth bound (0.29 flop/byte): no MFMA.  Vectors are
// slot-major, each slot contiguous and 256-B aligned, read with 16-B/lane
// non-temporal loads (1 KiB per wave instruction).  Compiled with
// -ffp-contract=off: the elementwise statements are rounded exactly like the
// reference expressions; the dot products use explicit fma().
//
// F08 = /root/reference/src-F08/nka_type.F90.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "nka_kernels.hpp is written for gfx950 (CDNA4) only: v_permlane32_swap / v_permlane16_swap reductions, 160 KiB LDS, tile shapes measured on MI355X.  Build with --offload-arch=gfx950."
#endif

#ifndef NKA_NT_LOADS
#define NKA_NT_LOADS 1      // streaming reads: non-temporal (nt) loads
#endif
#ifndef NKA_F_TEMPORAL
// != 0 (bit 0: in PA, bit 1: in PB) = the vectors BOTH passes of an update read (f and the raw w of the pending pair: PA reads them, PB reads them again
// a fraction of a millisecond later) are loaded with the default cache policy instead of nt, in the hope that the 256 MiB
// Infinity Cache still holds them for PB at shard sizes (n_local <= 1.25e7: 200 MB).  Measured in round 5 (in-process A/B
// of two builds, tools/ab_libs.py; profiles/r05/ab_mall_reuse.txt) together with PB walking its tiles in the reverse of
// PA's order (kPbReverse); 0 = every streaming load nt, the product.
#define NKA_F_TEMPORAL 0
#endif
#ifndef NKA_DEAD_SLOT_TILE0
// A launch wider than the list (the host's bound is one too high in the update that takes a dependence drop, and too high
// by more for a caller that never synchronises) has DEAD ring slots.  In PB they re-read f: 1 = always its first tile (4 KiB
// that stay in the caches), 0 = the tile at hand, as before round 4 -- half of which came from HBM again (PMC: 19.55 words
// per element where the list needs 19; PB -2.5 % with the first tile, neutral without dead slots).
#define NKA_DEAD_SLOT_TILE0 1
#endif
#ifndef NKA_STORE_POLICY
#define NKA_STORE_POLICY 1  // 0 plain, 1 nt (default: +2-3% on the mixed pass), 2 write-through "sc0 sc1 nt" (inline asm)
#endif

namespace nka {

constexpr int kBlock = 256;  // 4 wavefronts of 64
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kMaxGrid = 4096;  // upper bound on persistent grid size (partials buffer)

// ---- indices into the small device-resident control arrays -----------------
// int32 control block
enum {
  IC_SUBSPACE = 0,
  IC_PENDING = 1,
  IC_FIRST = 2,
  IC_LAST = 3,
  IC_FREE = 4,
  IC_NEW = 5,          // slot that receives (f_in, f_out) in the current update
  IC_NCOMB = 6,        // number of (slot, coefficient) pairs in the combine plan
  IC_PLAN_PENDING = 7, // plan for PA of the NEXT update: `pending` at its entry
  IC_PLAN_FIRST = 8,   //   slot holding the pending pair
  IC_PLAN_NOLDER = 9,  //   number of list entries to dot against
  IC_NRELAX = 10,      // count of s == 0 events (diagnostic)
  IC_NORMED = 11,      // this update normalises the pending pair (pending && s != 0)
  IC_HEADER = 16
};
// double control block
enum { DC_VTOL = 0, DC_S = 1, DC_HEADER = 2 };
// pointer control block (Ctl::pc)
enum {
  PC_FIRST_W = 0,  // w of the pending pair at the entry of the NEXT update (PA)
  PC_NEW_W = 1,    // w, v buffers of the slot that receives (f_in, f_out) in the current update (PB)
  PC_NEW_V = 2,
  PC_OLD_W = 3,    // what an out-of-place update displaced from that slot: handed to the caller /
  PC_OLD_V = 4,    //   kept as the library's next spare
  PC_HEADER = 8
};

constexpr int kMaxPerPass = 32;  // largest MAXL / MAXK instantiated

constexpr int kStamps = 16;      // s_memtime stamps of the scalar step, written only when NKA_SOLVE_STAMPS is defined
#ifdef NKA_SOLVE_STAMPS
#define NKA_STAMP(ctl, i) do { if (threadIdx.x == 0) (ctl).stamps()[i] = (double)__builtin_amdgcn_s_memtime(); } while (0)
#define NKA_STAMP0(ctl, i) do { if (blockIdx.x == 0 && threadIdx.x == 0) (ctl).stamps()[i] = (double)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define NKA_STAMP(ctl, i) do { } while (0)
#define NKA_STAMP0(ctl, i) do { } while (0)
#endif

struct Ctl {
  int32_t *ic;         // header, then next[M1+1], prev[M1+1], plan_slots[M1+pad], comb_slots[M1+pad]
  double *dc;          // header, then h[(M1+1)^2], c[M1+1], comb_c[M1+pad], red[2+2*mvec]
  int32_t mvec;
  // LIST WORD: one 64-bit word in pinned host memory (nullptr: none) that block 0 of PB overwrites with
  // (number of this update << kListWordLenBits | list length at its exit), see list_word_publish; `seq` = the
  // number the host gave this update.  The host reads it WITHOUT synchronising to learn that dependence drops
  // (F08:326-345) have made the list shorter than its own bookkeeping says (nka_hip.hip: list_bound_now).
  unsigned long long *hw;
  unsigned long long seq;
  // ADDRESS CONTROL BLOCK.  The streaming passes take the ADDRESSES of the stored vectors from here, not slot numbers:
  // wtab / vtab map slot -> buffer (at creation slot k -> (k-1)*stride of the two slot-major allocations; the out-of-place
  // entry nka_hip_accel_update_swap exchanges entries with buffers of the caller), and the scalar kernels, which alone know
  // the slots, resolve them when they write the plans.  Every entry is an OFFSET IN DOUBLES FROM Vecs::w (any buffer of the
  // device, the caller's included, is some 64-bit offset from it): a pointer read from memory carries no address space and
  // the compiler would reach it with FLAT loads -- one counter for LDS and memory, every wait a wait for everything (the
  // first version of this block did: PB -13 %) -- while vs.w + offset is a global address like any kernel argument.
  long long *pc;
  __host__ __device__ long long *plan_w() const { return pc + PC_HEADER; }          // w of PA's older entries [m1p]
  __host__ __device__ long long *comb_w() const { return plan_w() + m1p(); }        // w of PB's pairs [m1p]
  __host__ __device__ long long *comb_v() const { return comb_w() + m1p(); }        // v of PB's pairs [m1p]
  __host__ __device__ long long *wtab() const { return comb_v() + m1p(); }          // slot -> w buffer [m1+1], 1-based
  __host__ __device__ long long *vtab() const { return wtab() + (m1() + 1); }       // slot -> v buffer [m1+1]
  __host__ __device__ int pc_count() const { return PC_HEADER + 3 * m1p() + 2 * (m1() + 1); }
  // plan_slots / comb_slots / comb_c are padded by one pass width: the unrolled
  // kernels read (and ignore) entries up to the end of their last pass.
  __host__ __device__ int m1() const { return mvec + 1; }
  __host__ __device__ int m1p() const { return mvec + 1 + kMaxPerPass; }
  __host__ __device__ int32_t *next() const { return ic + IC_HEADER; }
  __host__ __device__ int32_t *prev() const { return next() + (m1() + 1); }
  __host__ __device__ int32_t *plan_slots() const { return prev() + (m1() + 1); }
  __host__ __device__ int32_t *comb_slots() const { return plan_slots() + m1p(); }
  __host__ __device__ int ic_count() const { return IC_HEADER + 2 * (m1() + 1) + 2 * m1p(); }
  __host__ __device__ double *h() const { return dc + DC_HEADER; }
  __host__ __device__ double *c() const { return h() + (m1() + 1) * (m1() + 1); }
  __host__ __device__ double *comb_c() const { return c() + (m1() + 1); }
  __host__ __device__ double *red() const { return comb_c() + m1p(); }
  __host__ __device__ int red_count() const { return 2 + 2 * mvec; }
  __host__ __device__ double *stamps() const { return red() + red_count(); }   // kStamps cycle stamps (diagnostic builds)
  __host__ __device__ int dc_count() const {
    return DC_HEADER + (m1() + 1) * (m1() + 1) + (m1() + 1) + m1p() + red_count() + kStamps;
  }
};
constexpr long long kNoBuffer = (long long)0x8000000000000000ull;      // "no buffer" among the offsets of Ctl::pc
constexpr int kListWordLenBits = 20;        // mvec + 1 <= 2^17 + 1 (nka_hip_create)
// PB, first thread of block 0, before its first tile: the store is posted while the pass streams, so it costs the
// update nothing and has landed long before the pass ends (a caller that synchronises once per iteration -- every
// solver reads a residual norm -- sees the word of the update it has just waited for).  ncomb + 1 = the combined
// entries plus the new pending pair = the list length at the exit of this update.
// Words 1..3 of the record belong to the out-of-place updates: the buffers the update displaced (PC_OLD_W / PC_OLD_V),
// written BEFORE the number of that update, which is stored with release semantics; other updates leave them alone.
__device__ __forceinline__ void list_word_publish(const Ctl &ctl, int ncomb, int swapping) {
  if (ctl.hw != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    if (swapping) {      // an out-of-place update: what it displaced (words 1, 2), then its number (word 3)
      ctl.hw[1] = (unsigned long long)ctl.pc[PC_OLD_W];       // (offsets from Vecs::w, like everything in the block)
      ctl.hw[2] = (unsigned long long)ctl.pc[PC_OLD_V];
      __hip_atomic_store(ctl.hw + 3, ctl.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(ctl.hw, (ctl.seq << kListWordLenBits) | (unsigned long long)(ncomb + 1), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// red[] layout (raw sums of PA, d = w1 - f NOT yet divided by s):
//   [0] sum d^2, [1] <f,d>, [2+p] <d,w_older(p)>, [2+mvec+p] <f,w_older(p)>

struct Vecs {
  double *v, *w;       // slot k (1-based) at base + (k-1)*stride
  int64_t stride;      // in doubles, multiple of 32 (256 B)
  int64_t n;           // local vector length
};

// ---- PEER-TO-PEER EXCHANGE of the 2 + 2 mvec sums (round 5, opt-in: nka_hip_p2p_attach) -----------------------------------
// The one exchange of a sharded update is an all-reduce of 336 bytes: latency, not bandwidth.  Through RCCL it is a kernel of
// its own between the final sums and the scalar step.  Here every rank owns a MAILBOX in fine-grained device memory that its
// peers map through hipIpc: the final-sums kernel of rank p writes each sum it forms straight into row p of EVERY rank's
// mailbox (value, then -- released at system scope -- the number of the exchange as that entry's flag), and the scalar step of
// rank q starts by waiting, entry by entry, for the N flags and adding the N rows IN RANK ORDER: the same additions in the same
// order on every rank, hence the same bits -- no communication kernel, two kernel boundaries fewer.
//   mailbox of one rank: val[2][N][cap] doubles, then flag[2][N][cap] 64-bit words; slot = exchange number & 1.  Two slots
//   suffice: a rank can start exchange x+2 only after its scalar step of x+1 has seen EVERY peer's row of x+1, and a peer sends
//   x+1 only after its own scalar step has consumed x (stream order).
//   `xseq` (device memory of this rank): number of the NEXT exchange, advanced by whoever gathers; device-resident so that a
//   captured update replays correctly.  Peers' mailboxes are reached as BYTE OFFSETS from this rank's own (`off[q]`): a pointer
//   read from memory has no address space and would be accessed with FLAT instructions (see Ctl::pc).
//   A wait is bounded (`timeout_ticks` of the 100 MHz wall clock): a peer that never sends makes the gather store NaNs, raise
//   `status` and go on, so that the grid always drains; the host reports NKA_HIP_ECOMM at its next synchronising call.
struct P2P {
  char *base;                    // this rank's mailbox (nullptr: no peer-to-peer exchange)
  const long long *off;          // [n] byte offset of rank q's mailbox from `base` (device memory)
  unsigned long long *xseq;      // number of the next exchange (device memory, starts at 1)
  int *status;                   // != 0: a wait timed out
  int n, me, cap;
  long long timeout_ticks;
  __device__ double *val(int q, int slot, int src, int e) const {
    return reinterpret_cast<double *>(base + off[q]) + ((size_t)slot * n + src) * cap + e;
  }
  __device__ unsigned long long *flag(int q, int slot, int src, int e) const {
    return reinterpret_cast<unsigned long long *>(base + off[q]) + (size_t)2 * n * cap + ((size_t)slot * n + src) * cap + e;
  }
};
__host__ __device__ inline size_t p2p_mailbox_bytes(int n, int cap) { return (size_t)2 * n * cap * 16; }

// One lane sends entry e of exchange `seq` to rank q: the value, then the flag released at system scope.
__device__ __forceinline__ void p2p_send_one(const P2P &x, int q, unsigned long long seq, int e, double v) {
  const int slot = (int)(seq & 1ull);
  __hip_atomic_store(x.val(q, slot, x.me, e), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(x.flag(q, slot, x.me, e), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// A whole wavefront sends entry e (lanes q = lane, lane + 64, ... < n each serve one peer); v is uniform.
__device__ __forceinline__ void p2p_send_wave(const P2P &x, unsigned long long seq, int e, double v) {
  for (int q = threadIdx.x & 63; q < x.n; q += 64) p2p_send_one(x, q, seq, e, v);
}
// Entry e of exchange `seq`, summed over the ranks in rank order (one lane).  Bounded wait.
__device__ __forceinline__ double p2p_gather_one(const P2P &x, unsigned long long seq, int e) {
  const int slot = (int)(seq & 1ull);
  double acc = 0.0;
  bool late = false;
  const long long t0 = wall_clock64();
  for (int r = 0; r < x.n; r++) {
    unsigned long long *fl = x.flag(x.me, slot, r, e);
    while (!late && __hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > x.timeout_ticks) late = true;
    }
    const double v = __hip_atomic_load(x.val(x.me, slot, r, e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    acc = (r == 0) ? v : acc + v;
  }
  if (late) {
    *x.status = 1;
    acc = __builtin_nan("");
  }
  return acc;
}
// The gather at the head of the scalar step: red[e] <- sum over ranks, e < count, by the threads of ONE workgroup; then the
// exchange number moves on.  (Each thread reads back only entries it wrote itself or after the barrier.)
__device__ __forceinline__ void p2p_gather_block(const P2P &x, double *red, int count) {
  const unsigned long long seq = *x.xseq;
  for (int e = threadIdx.x; e < count; e += blockDim.x) red[e] = p2p_gather_one(x, seq, e);
  __syncthreads();
  if (threadIdx.x == 0) *x.xseq = seq + 1;
}
// The generic form (the hook behind nka_hip_allreduce_now, the self-test and the reference-order chain): one workgroup sends
// its `count` values to every rank, then gathers.
static __global__ __launch_bounds__(128) __attribute__((unused)) void k_p2p_allreduce(P2P x, double *buf, int count) {
  const unsigned long long seq = *x.xseq;
  for (int i = threadIdx.x; i < count * x.n; i += blockDim.x) p2p_send_one(x, i % x.n, seq, i / x.n, buf[i / x.n]);
  __syncthreads();
  p2p_gather_block(x, buf, count);
}

__device__ __forceinline__ double readlane_f64(double x, int src_lane_uniform) {
  union { double d; int i[2]; } u;
  u.d = x;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src_lane_uniform);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src_lane_uniform);
  return u.d;
}

// ---- reductions ---------------------------------------------------------------
// Sum over the wavefront, valid in LANE 0: the butterfly x += x[lane + off], off = 32, 16, 8, 4, 2, 1 -- the tree
// __shfl_down builds, hence the same bits -- but through REGISTERS: gfx950's v_permlane32_swap / v_permlane16_swap
// for the two steps that cross a row of 16 lanes, DPP row_shl for the four inside row 0 (after the step with
// offset 16 only lanes 0..15 carry partial sums that reach lane 0).  __shfl_down is two ds_bpermute_b32 and an
// LDS wait per step: the 42 sums of a PA block took 22 k cycles (~9.5 us of a 15 us launch at n = 1e5) that way.
__device__ __forceinline__ double swap_sum32(double A, double B);
__device__ __forceinline__ double swap_sum16(double A, double B);
template <int N> __device__ __forceinline__ double row_shl_sum(double x);
__device__ __forceinline__ double wave_sum(double x) {
  x = swap_sum32(x, x);
  x = swap_sum16(x, x);
  x = row_shl_sum<8>(x);
  x = row_shl_sum<4>(x);
  x = row_shl_sum<2>(x);
  return row_shl_sum<1>(x);
}

// The first two butterfly steps for TWO sums at once.  swap_sum32(A, B): lanes 0..31 get A[i] + A[i+32], lanes
// 32..63 get B[i-32] + B[i]; swap_sum16(A, B), row by row of 16 lanes: (A.r0 + A.r1, B.r0 + B.r1, A.r2 + A.r3,
// B.r2 + B.r3).  The same pairs the butterfly of wave_sum adds, parked in the half / row that the butterfly
// leaves idle.
__device__ __forceinline__ double swap_sum32(double A, double B) {
  union U { double d; unsigned u[2]; } a, b;
  a.d = A;
  b.d = B;
#pragma unroll
  for (int w = 0; w < 2; w++) {
    const auto r = __builtin_amdgcn_permlane32_swap(a.u[w], b.u[w], false, false);
    a.u[w] = r[0];
    b.u[w] = r[1];
  }
  return a.d + b.d;
}
__device__ __forceinline__ double swap_sum16(double A, double B) {
  union U { double d; unsigned u[2]; } a, b;
  a.d = A;
  b.d = B;
#pragma unroll
  for (int w = 0; w < 2; w++) {
    const auto r = __builtin_amdgcn_permlane16_swap(a.u[w], b.u[w], false, false);
    a.u[w] = r[0];
    b.u[w] = r[1];
  }
  return a.d + b.d;
}
// x[i] + x[i+N] inside every row of 16 lanes (lanes whose partner is outside the row keep x + x: never used)
template <int N> __device__ __forceinline__ double row_shl_sum(double x) {
  union U { double d; unsigned u[2]; } a, b;
  a.d = x;
  b.u[0] = __builtin_amdgcn_update_dpp(a.u[0], a.u[0], 0x100 + N, 0xf, 0xf, false);
  b.u[1] = __builtin_amdgcn_update_dpp(a.u[1], a.u[1], 0x100 + N, 0xf, 0xf, false);
  return a.d + b.d;
}

// Sum NACC per-thread accumulators over the block (fixed order: lanes by butterfly, then waves 0..3) and store
// column a at partials[a*G + block].  Every sum is the tree of wave_sum -- the same bits -- but the NACC
// butterflies share their steps: the step with offset 32 folds accumulators k and k + H1 into one register
// (lower / upper half of the wavefront), the step with offset 16 folds registers k and k + H2 (even / odd rows),
// the four steps inside a row then serve four accumulators each.  ~NACC/4 x 6 exchange-and-add groups instead of
// NACC x 6 (42 sums of PA at m = 20: 9.9 k -> ~3 k cycles; with __shfl_down 22 k).
template <int NACC>
__device__ __forceinline__ void block_reduce_store(const double (&acc)[NACC], double *partials, int G) {
  __shared__ double sm[kWavesPerBlock][NACC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int H1 = (NACC + 1) / 2, H2 = (H1 + 1) / 2;
  double r1[H1], r2[H2];
#pragma unroll
  for (int k = 0; k < H1; k++) r1[k] = swap_sum32(acc[k], acc[k + H1 < NACC ? k + H1 : k]);
#pragma unroll
  for (int k = 0; k < H2; k++) r2[k] = swap_sum16(r1[k], r1[k + H2 < H1 ? k + H2 : k]);
  const int row = lane >> 4;
#pragma unroll
  for (int k = 0; k < H2; k++) {
    double x = r2[k];
    x = row_shl_sum<8>(x);
    x = row_shl_sum<4>(x);
    x = row_shl_sum<2>(x);
    x = row_shl_sum<1>(x);
    // row 0: accumulator k; row 1: k + H2 (a register of the second half); rows 2, 3: the same + H1
    const int reg = k + (row & 1) * H2;
    const int a = reg + (row >> 1) * H1;
    if ((lane & 15) == 0 && reg < H1 && a < NACC) sm[wv][a] = x;
  }
  __syncthreads();
  for (int a = threadIdx.x; a < NACC; a += kBlock) {
    double r = sm[0][a];
#pragma unroll
    for (int q = 1; q < kWavesPerBlock; q++) r += sm[q][a];
    partials[(size_t)a * G + blockIdx.x] = r;
  }
}

// ---- 8-B / 16-B per lane streaming accesses ---------------------------------------
typedef double d2 __attribute__((ext_vector_type(2)));
template <int VEC> struct VecT;
template <> struct VecT<1> { using type = double; };
template <> struct VecT<2> { using type = d2; };

template <int VEC> __device__ __forceinline__ typename VecT<VEC>::type ld(const double *p);
template <> __device__ __forceinline__ double ld<1>(const double *p) {
#if NKA_NT_LOADS
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <> __device__ __forceinline__ d2 ld<2>(const double *p) {
#if NKA_NT_LOADS
  return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p));
#else
  return *reinterpret_cast<const d2 *>(p);
#endif
}
// loads of the vectors that PA and PB both read (see NKA_F_TEMPORAL: bit 0 = in PA, bit 1 = in PB)
template <int VEC, int PASS_BIT> __device__ __forceinline__ typename VecT<VEC>::type ld_keep(const double *p) {
  if constexpr ((NKA_F_TEMPORAL & PASS_BIT) != 0) return *reinterpret_cast<const typename VecT<VEC>::type *>(p);
  else return ld<VEC>(p);
}
__device__ __forceinline__ void st(double *p, double x) {
#if NKA_STORE_POLICY == 1
  __builtin_nontemporal_store(x, p);
#else
  *p = x;
#endif
}
__device__ __forceinline__ void st(double *p, d2 x) {
#if NKA_STORE_POLICY == 1
  __builtin_nontemporal_store(x, reinterpret_cast<d2 *>(p));
#elif NKA_STORE_POLICY == 2
  // write-through streaming store; a store needs no later wait in this wave
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(x) : "memory");
#else
  *reinterpret_cast<d2 *>(p) = x;
#endif
}

__device__ __forceinline__ double ex(double x, int) { return x; }
__device__ __forceinline__ double ex(d2 x, int i) { return x[i]; }
__device__ __forceinline__ void setc(double &x, int, double val) { x = val; }
__device__ __forceinline__ void setc(d2 &x, int i, double val) { x[i] = val; }

// ---- PA: every inner product of the update in one pure-read pass --------------------
// MAXL stored vectors per pass; entries beyond the actual count re-read f (cache
// hit) into accumulators that are discarded, which keeps every load of a tile
// unconditional so that all MAXL+2 of them are in flight together.
// acc: [0] sum d^2, [1] <f,d>, [2+j] <d,w_j>, [2+MAXL+j] <f,w_j>.
// `normed` (round 5, NKA_HIP_SUMS_BLOCKED_ROUNDED): the norm is already known -- red[0] holds the GLOBAL sum d^2 of a pass of
// its own (k_norm_diff) -- and the sums are formed on the ROUNDED w1' = fl(d/s) (bit 1 of `normed`: fl((1/s)*d), the
// F08-vector flavour), the value PB stores: acc[1] = <f,w1'>, acc[2+j] = <w1',w_j> as the reference defines them (F08:286-290,
// 371), in blocks and with fma.  The scalar step then takes them as they are (kSolvePrenorm).
__device__ __forceinline__ double pa_operand(double d, int normed, double s, double rs) {
  if (normed == 0) return d;
  if (s == 0.0) return 0.0;                       // (the scalar step relaxes, F08:275: these sums are dead)
  return (normed & 2) ? rs * d : d / s;
}

template <int MAXL, int VEC>
__global__ __launch_bounds__(kBlock) void k_dots(Ctl ctl, Vecs vs, const double *__restrict__ f,
                                                 double *__restrict__ partials, int pass, int normed) {
  using V = typename VecT<VEC>::type;
  constexpr int NACC = 2 * MAXL + 2;
  const int G = gridDim.x;
  const double s_n = normed ? sqrt(ctl.red()[0]) : 1.0, rs_n = 1.0 / s_n;
  const int pending = ctl.ic[IC_PLAN_PENDING];
  const int nolder = ctl.ic[IC_PLAN_NOLDER];
  const int base = pass * MAXL;
  // no pending pair: d = f - f = 0 and its sums are discarded by k_finalize_dots
  const double *w1 = pending ? vs.w + ctl.pc[PC_FIRST_W] : f;
  const long long *pw = ctl.plan_w();
  const double *wk[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; j++) {
    const int p = base + j;
    wk[j] = (p < nolder) ? vs.w + pw[p] : f;
  }
  double acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; a++) acc[a] = 0.0;

  const int64_t ntile = vs.n / (kBlock * VEC);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    const V fv = ld<VEC>(f + e);
    const V w1v = ld<VEC>(w1 + e);
    V wkv[MAXL];
#pragma unroll
    for (int j = 0; j < MAXL; j++) wkv[j] = ld<VEC>(wk[j] + e);
    // keep all MAXL+2 loads of the tile in flight: no FMA may be scheduled
    // between them (hipcc otherwise serialises load/wait/use to save registers)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < VEC; q++) {
      const double fq = ex(fv, q);
      const double d = pa_operand(ex(w1v, q) - fq, normed, s_n, rs_n);      // F08:266 ((-1)*f + w1 in F08V:237: same bits)
      acc[0] = fma(d, d, acc[0]);
      acc[1] = fma(fq, d, acc[1]);
#pragma unroll
      for (int j = 0; j < MAXL; j++) {
        acc[2 + j] = fma(d, ex(wkv[j], q), acc[2 + j]);
        acc[2 + MAXL + j] = fma(fq, ex(wkv[j], q), acc[2 + MAXL + j]);
      }
    }
  }
  if ((int)blockIdx.x == G - 1) {  // ragged tail, scalar
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < vs.n; i += kBlock) {
      const double fq = f[i];
      const double d = pa_operand(w1[i] - fq, normed, s_n, rs_n);
      acc[0] = fma(d, d, acc[0]);
      acc[1] = fma(fq, d, acc[1]);
#pragma unroll
      for (int j = 0; j < MAXL; j++) {
        const double x = wk[j][i];
        acc[2 + j] = fma(d, x, acc[2 + j]);
        acc[2 + MAXL + j] = fma(fq, x, acc[2 + MAXL + j]);
      }
    }
  }
  block_reduce_store<NACC>(acc, partials, G);
}

// The norm pass of NKA_HIP_SUMS_BLOCKED_ROUNDED: sum d^2 with d = w1 - f (F08:266-267) over this rank's slice, two streams,
// per-block partial sums in column 0 of `partials` (k_norm_fin adds them in a fixed order).
static __global__ __launch_bounds__(kBlock) __attribute__((unused)) void k_norm_diff(Ctl ctl, Vecs vs, const double *__restrict__ f,
                                                                                         double *__restrict__ partials) {
  const int G = gridDim.x;
  const double *w1 = vs.w + ctl.pc[PC_FIRST_W];
  const bool v2 = (reinterpret_cast<uintptr_t>(f) % 16) == 0;      // (slot bases are 256-byte aligned)
  double acc = 0.0;
  int64_t done = 0;
  if (v2) {
    const int64_t ntile = vs.n / (kBlock * 2);
    int64_t t = blockIdx.x;
    // Round 6 (the pass runs in every update since it became the default): kNormAhead tiles' loads go out before the first
    // of them is consumed -- two loads in flight per thread kept one block per CU at 4.0 TB/s; the accumulation visits the
    // tiles in the same order as the plain loop below, so the partial sums carry the same bits
    constexpr int kNormAhead = 8;
    for (; t + (int64_t)(kNormAhead - 1) * G < ntile; t += (int64_t)kNormAhead * G) {
      d2 fv[kNormAhead], wv[kNormAhead];
#pragma unroll
      for (int u = 0; u < kNormAhead; u++) {
        const int64_t e = (t + (int64_t)u * G) * (kBlock * 2) + threadIdx.x * 2;
        fv[u] = ld<2>(f + e);
        wv[u] = ld<2>(w1 + e);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kNormAhead; u++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const double d = wv[u][q] - fv[u][q];
          acc = fma(d, d, acc);
        }
    }
    for (; t < ntile; t += G) {
      const int64_t e = t * (kBlock * 2) + threadIdx.x * 2;
      const d2 fv = ld<2>(f + e), wv = ld<2>(w1 + e);
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const double d = wv[q] - fv[q];
        acc = fma(d, d, acc);
      }
    }
    done = ntile * (kBlock * 2);
  }
  // scalar path: the ragged tail (last block), or everything when f is not 16-byte aligned (grid-stride over elements)
  if (v2) {
    if ((int)blockIdx.x == G - 1)
      for (int64_t i = done + threadIdx.x; i < vs.n; i += kBlock) {
        const double d = w1[i] - f[i];
        acc = fma(d, d, acc);
      }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < vs.n; i += (int64_t)G * kBlock) {
      const double d = w1[i] - f[i];
      acc = fma(d, d, acc);
    }
  }
  const double one[1] = {acc};
  block_reduce_store<1>(one, partials, G);
}
// ... and its final sum, one wavefront, into red[0] (zero without a pending pair: nothing stale reaches the exchange)
static __global__ __launch_bounds__(64) __attribute__((unused)) void k_norm_fin(Ctl ctl, const double *__restrict__ partials, int G) {
  double r = 0.0;
  for (int b = threadIdx.x; b < G; b += 64) r += partials[b];
  r = wave_sum(r);
  if (threadIdx.x == 0) ctl.red()[0] = ctl.ic[IC_PLAN_PENDING] ? r : 0.0;
}

// PA with a SMALL ROLLING WINDOW of loads.  tools/hbm_probe (mode f) showed that a
// pure-read kernel with the arithmetic of this pass runs at 7.15 TB/s when each wave
// keeps only ~6 loads in flight and re-issues one as soon as one has been consumed,
// against 6.4-6.7 TB/s for k_dots with all 22 loads of a tile in flight: fewer streams
// are open at any moment (DRAM page locality) and the fp64 FMAs interleave with the
// load issue.  Here the MAXL stored vectors of a tile go through a ring of W registers
// (slot j mod W holds vector j; when vector j has been accumulated its slot is re-loaded
// with vector j+W of this tile or vector j+W-MAXL of the block's next tile), and f, w1
// of the next tile are requested as soon as this tile's copies are in d / fq.  Same
// products, same per-thread accumulation order => same bits as k_dots.  Single pass, VEC = 2.
// `base` (round 5): the first plan entry of this launch.  A list longer than kMaxPerPass is served by several launches of
// BALANCED exact widths (33 = 17 + 16: enqueue_pa), each on its own part of the plan; only the launch with base == 0 has its
// first two sums (d^2, <f,d>) used (k_finalize_dots).
template <int MAXL, int W>
__global__ __launch_bounds__(kBlock) void k_dots_win(Ctl ctl, Vecs vs, const double *__restrict__ f,
                                                     double *__restrict__ partials, int base, int normed) {
  constexpr int VEC = 2;
  using V = typename VecT<VEC>::type;
  constexpr int NACC = 2 * MAXL + 2;
  static_assert(MAXL % W == 0, "the ring must divide the stored vectors of a tile");
  NKA_STAMP0(ctl, 10);
  const int G = gridDim.x;
  const int pending = ctl.ic[IC_PLAN_PENDING];
  const double s_n = normed ? sqrt(ctl.red()[0]) : 1.0, rs_n = 1.0 / s_n;      // (see k_dots: sums on the rounded w1')
  const int nolder = ctl.ic[IC_PLAN_NOLDER] - base;        // entries of the plan from `base` on (<= 0: none, every slot dead)
  const long long *pw = ctl.plan_w() + base;
  const double *w1p = vs.w + ctl.pc[PC_FIRST_W];           // (read whether pending or not: no branch around a load)
  const double *w1 = pending ? w1p : f;
  // every plan slot is requested at once, whether the list reaches it or not (the plan array is longer than any
  // width): written as `j < nolder ? slots[j] ...` each slot became a branch around its own s_load + s_waitcnt --
  // twenty serial scalar round trips, 4.2 k cycles of prologue at m = 20 against 2 k at m = 5
  long long sl[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; j++) sl[j] = pw[j];
  const double *wk[MAXL];
#pragma unroll
  for (int j = 0; j < MAXL; j++) wk[j] = (j < nolder) ? vs.w + sl[j] : f;
  double acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; a++) acc[a] = 0.0;

  const int64_t ntile = vs.n / (kBlock * VEC);
  // (dead ring slots -- a launch wider than the list -- re-read f at the tile at hand here.  Sending them to f's first
  //  tile, as PB does (NKA_DEAD_SLOT_TILE0), was measured in this pass too: 25 more VGPRs for the per-slot offsets and
  //  +2...5 % of PA with NO dead slot, which is every launch of a caller that synchronises once per iteration, since PA
  //  then runs at exactly the list length; profiles/r04/ab_dead_slot.txt)
#define DEAD_OFF(live, off) (off)
  V fv, w1v, ring[W];
  int64_t t = blockIdx.x;
  if (t < ntile) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    fv = ld_keep<VEC, 1>(f + e);
    w1v = ld_keep<VEC, 1>(w1 + e);
#pragma unroll
    for (int j = 0; j < W; j++) ring[j] = ld<VEC>(wk[j] + (DEAD_OFF(j < nolder, e)));
  }
  NKA_STAMP0(ctl, 11);
  for (; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    const int64_t tn = (t + G < ntile) ? t + G : t;     // the last iteration prefetches its own tile again
    const int64_t en = tn * (kBlock * VEC) + threadIdx.x * VEC;
    double dq[VEC], fq[VEC];
#pragma unroll
    for (int q = 0; q < VEC; q++) {
      fq[q] = ex(fv, q);
      dq[q] = pa_operand(ex(w1v, q) - fq[q], normed, s_n, rs_n);      // F08:266 (and F08:283 when the norm is known)
      acc[0] = fma(dq[q], dq[q], acc[0]);
      acc[1] = fma(fq[q], dq[q], acc[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    fv = ld_keep<VEC, 1>(f + en);
    w1v = ld_keep<VEC, 1>(w1 + en);
#pragma unroll
    for (int j = 0; j < MAXL; j++) {
      const V x = ring[j % W];
      __builtin_amdgcn_sched_barrier(0);
      if (j + W < MAXL) ring[j % W] = ld<VEC>(wk[j + W] + (DEAD_OFF(j + W < nolder, e)));
      else ring[j % W] = ld<VEC>(wk[j + W - MAXL] + (DEAD_OFF(j + W - MAXL < nolder, en)));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < VEC; q++) {
        acc[2 + j] = fma(dq[q], ex(x, q), acc[2 + j]);
        acc[2 + MAXL + j] = fma(fq[q], ex(x, q), acc[2 + MAXL + j]);
      }
    }
  }
  if ((int)blockIdx.x == G - 1) {  // ragged tail, scalar
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < vs.n; i += kBlock) {
      const double fq = f[i];
      const double d = pa_operand(w1[i] - fq, normed, s_n, rs_n);
      acc[0] = fma(d, d, acc[0]);
      acc[1] = fma(fq, d, acc[1]);
#pragma unroll
      for (int j = 0; j < MAXL; j++) {
        const double x = wk[j][i];
        acc[2 + j] = fma(d, x, acc[2 + j]);
        acc[2 + MAXL + j] = fma(fq, x, acc[2 + MAXL + j]);
      }
    }
  }
  NKA_STAMP0(ctl, 12);
  block_reduce_store<NACC>(acc, partials, G);
  NKA_STAMP0(ctl, 13);
#undef DEAD_OFF
}

// Final sums of one PA pass scattered into red[] (layout above).  One wavefront
// per column (grid = 2*MAXL+2 blocks of 64): each lane sums its strided share in
// order, then a butterfly -- the same bits on every run.
constexpr int kFinThreads = 64;
// `ncover` = entries of each row that the passes of this update write (npass*MAXL):
// pass 0 zeroes the rest, so red[] never carries stale sums into the all-reduce.
// `x.base` != nullptr: the sums do not stay here -- each one goes to row `me` of every rank's mailbox (P2P above) and the
// scalar step gathers them; red[] is then written by that gather.
template <int MAXL>
__global__ __launch_bounds__(kFinThreads) void k_finalize_dots(Ctl ctl, const double *__restrict__ partials, int G,
                                                               int pass, int ncover, int base, P2P x, int keep0 = 0) {
  const int lane = threadIdx.x;
  const int c = blockIdx.x;
  const bool p2p = x.base != nullptr;
  const unsigned long long xs = p2p ? *x.xseq : 0ull;
  // the column's partial sums are requested BEFORE the plan is known (whether the column is live only decides
  // if its sum or a zero is stored): the two memory round trips overlap instead of following one another
  double r = 0.0;
  for (int b = lane; b < G; b += kFinThreads) r += partials[(size_t)c * G + b];
  r = wave_sum(r);
  const int nolder = ctl.ic[IC_PLAN_NOLDER];
  const int pending = ctl.ic[IC_PLAN_PENDING];
  // (`base` = first plan entry of this pass: pass * MAXL for the passes of equal width, the running sum of the widths
  //  for the balanced passes of the window kernels)
  if (pass == 0 && c == 0)
    for (int p = ncover + lane; p < ctl.mvec; p += kFinThreads) {
      if (p2p) {
        for (int q = 0; q < x.n; q++) {
          p2p_send_one(x, q, xs, 2 + p, 0.0);
          p2p_send_one(x, q, xs, 2 + ctl.mvec + p, 0.0);
        }
      } else {
        ctl.red()[2 + p] = 0.0;
        ctl.red()[2 + ctl.mvec + p] = 0.0;
      }
    }
  int dst = -1;
  bool live = false;
  if (c < 2) {
    if (pass == 0 && !(keep0 && c == 0)) { dst = c; live = pending != 0; }      // (keep0: red[0] holds the norm of a pass of its own)
  } else if (c < 2 + MAXL) {
    const int p = base + (c - 2);
    if (p < ctl.mvec) { dst = 2 + p; live = pending && p < nolder; }
  } else {
    const int p = base + (c - 2 - MAXL);
    if (p < ctl.mvec) { dst = 2 + ctl.mvec + p; live = p < nolder; }
  }
  if (dst < 0) return;
  if (p2p) {
    p2p_send_wave(x, xs, dst, live ? readlane_f64(r, 0) : 0.0);
    return;
  }
  if (lane == 0) ctl.red()[dst] = live ? r : 0.0;
}

// ---- PA in the REFERENCE'S ORDER: every sum of the update as the reference forms it ---------------
// One workgroup.  The reference's inner products are sequential sums of rounded products (its default dot product:
// C .c:200-208; `dot_product(x, y)` in F08:216-219, compiled without contraction as oracle/Makefile does), the norm first
// (F08:267), then -- with w1' = d/s (F08:283; (1/s)*d in the vector flavour, F08V:256) already ROUNDED -- the Gram row
// <w1',w_k> (F08:286-290) and the projections <f,w_k>, <f,w1'> (F08:371).  This kernel forms exactly those: element after
// element, one rounding per product and per addition (no fma).  A chunk of f, of the normalised w1' and of every older w is
// staged in LDS by the whole workgroup (coalesced); then thread r walks the chunk for sum r, so that the only serial
// thing per element is the addition itself.  Rows are `chunk + 1` apart (the lanes of a wavefront read different rows at
// the same element: an odd stride spreads them over the banks).  red[] then holds  [0] sum d^2, [1] <f,w1'>,
// [2+p] <w1',w_p>, [2+mvec+p] <f,w_p>  -- the scalar step takes [1] and the Gram row as they are (kSolvePrenorm) -- and,
// the scalar step and PB's statements being bit-exact given their inputs, the update returns THE REFERENCE'S BITS.
// Cost: two chains of n dependent additions (the norm, then the sums on w1'), 20-30 ns per element on an otherwise idle
// MI355X (tools/sum_order_cost.py, profiles/r04/sum_order_cost.txt): on par with the blocked PA and its final sums up to
// n = 64 (the default there), +12-18 us per update at n = 512, 40 ms per update at n = 1e6.  Single rank only: the Gram row needs the GLOBAL norm
// first, i.e. a second exchange (nka_hip_set_sum_order).
constexpr int kOrdThreads = 256;
constexpr int kOrdChunkMax = 512;
constexpr int kOrdMaxMvec = 250;                     // two sums per thread and eight elements per LDS row at least
constexpr int kOrdAutoMax = 64;                      // NKA_HIP_SUMS_AUTO sums in the reference's order up to this length (where it costs nothing)
constexpr int kOrdLdsDoubles = 16000;                // 125 KiB of dynamic LDS (one workgroup; 160 KiB per CU on gfx950) ...
constexpr int kOrdLdsPad = 16;                       // ... plus what ord_sum may read past the last row
__host__ __device__ inline int ord_chunk(int rows) {
  int c = kOrdLdsDoubles / (rows < 1 ? 1 : rows) - 1;
  return c > kOrdChunkMax ? kOrdChunkMax : (c < 8 ? 8 : c);
}
__host__ __device__ inline size_t ord_lds_bytes(int rows) {
  return sizeof(double) * ((size_t)rows * (ord_chunk(rows) + 1) + kOrdLdsPad);
}
// a + x[0]*y[0] + x[1]*y[1] + ... in THAT order, one rounding per product and per addition; the LDS reads of the next eight
// elements are in flight while the eight additions of this batch wait for one another (rows are padded: reading up to
// seven elements past `len` stays inside the allocation; those products are not added).
__device__ __forceinline__ double ord_sum(double a, const double *x, const double *y, int len) {
#pragma clang fp contract(off)      // products and additions stay separate roundings whatever the build's flags
  double xb[8], yb[8];
#pragma unroll
  for (int u = 0; u < 8; u++) { xb[u] = x[u]; yb[u] = y[u]; }
  for (int i0 = 0; i0 < len; i0 += 8) {
    double xn[8], yn[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { xn[u] = x[i0 + 8 + u]; yn[u] = y[i0 + 8 + u]; }
    if (i0 + 8 <= len) {
#pragma unroll
      for (int u = 0; u < 8; u++) a = a + xb[u] * yb[u];
    } else {
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (i0 + u < len) a = a + xb[u] * yb[u];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) { xb[u] = xn[u]; yb[u] = yn[u]; }
  }
  return a;
}
// Chunk [c0, c0+len) of the older w's into LDS rows 2..: eight rows at a time, two elements of each row per thread (a chunk
// has at most 512 elements): sixteen loads in flight per thread, the row addresses uniform (scalar loads of the plan).
__device__ __forceinline__ void ord_load_older(double *sh, int S, const Vecs &vs, const long long *pw, int nolder, int64_t c0, int len) {
  const int i0 = threadIdx.x, i1 = threadIdx.x + kOrdThreads;
  static_assert(kOrdChunkMax <= 2 * kOrdThreads, "two elements of a row per thread cover a chunk");
  for (int p0 = 0; p0 < nolder; p0 += 8) {
    double v0[8], v1[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int p = p0 + u < nolder ? p0 + u : nolder - 1;      // (the last group repeats a row: the loads stay unconditional)
      const double *wp = vs.w + pw[p] + c0;
      v0[u] = i0 < len ? wp[i0] : 0.0;
      v1[u] = i1 < len ? wp[i1] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (p0 + u >= nolder) continue;
      double *dst = sh + (size_t)(2 + p0 + u) * S;
      if (i0 < len) dst[i0] = v0[u];
      if (i1 < len) dst[i1] = v1[u];
    }
  }
}

// SHARDED (round 5): the reference's sum over the GLOBAL vector is one chain of additions through the slices in rank
// order, so rank r CONTINUES the running sums of rank r-1: `carry` != 0 starts every accumulator from the value red[]
// holds (the prefix over the ranks before this one) instead of 0, and the update is made in two kinds of rounds (nka_hip.hip,
// ordered_chain): phase 1 = the norm only (red[0]); phase 2 = with s from the GLOBAL red[0], the sums on the rounded w1'
// and on f (red[1..]).  phase 0 = both in one launch, the single-rank form described above.
enum { kOrdAll = 0, kOrdNorm = 1, kOrdRows = 2 };
static __global__ __launch_bounds__(kOrdThreads) __attribute__((unused)) void k_dots_ordered(Ctl ctl, Vecs vs,
                                                                                              const double *__restrict__ f,
                                                                                              int rcp, int chunk, int phase,
                                                                                              int carry) {
  extern __shared__ double ord_sh[];
  __shared__ double sum_dd;
  const int t = threadIdx.x;
  const int pending = ctl.ic[IC_PLAN_PENDING];
  const int nolder = ctl.ic[IC_PLAN_NOLDER];
  const int mvec = ctl.mvec;
  const int64_t n = vs.n;
  const int S = chunk + 1;                           // row stride in LDS
  const double *w1 = pending ? vs.w + ctl.pc[PC_FIRST_W] : f;
  const long long *pw = ctl.plan_w();
  double *red = ctl.red();
  double *row_f = ord_sh, *row_w1 = ord_sh + S;      // rows 2.. : the older w's
  const bool single = n <= chunk && phase == kOrdAll;   // the whole vectors fit: every global load of the update is issued ONCE

  // The sums on the ROUNDED w1' wait for the norm, those on f alone do not: they live in DIFFERENT wavefronts, so that the
  // second kind is summed while thread 0 sums the norm.  Threads 0..127 own the w1' sums  r = t, t + 128  (r = 0: <f,w1'>;
  // 1 <= r <= nolder: <w1',w_(r-1)>), threads 128..255 the sums on f  p = t - 128, t  (<f,w_p>, p < nolder).
  double acc[2] = {0.0, 0.0};
  const double *xr[2] = {row_f, row_f}, *yr[2] = {row_f, row_f};
  int dst[2] = {-1, -1};                             // where the sum goes in red[]
  const bool on_w1 = t < kOrdThreads / 2;
  for (int q = 0; q < 2; q++) {
    if (on_w1) {
      const int r = t + q * (kOrdThreads / 2);
      if (r > nolder) continue;
      if (r == 0) { xr[q] = row_f; yr[q] = row_w1; dst[q] = 1; }
      else { xr[q] = row_w1; yr[q] = ord_sh + (size_t)(2 + (r - 1)) * S; dst[q] = 2 + (r - 1); }
    } else {
      const int p = (t - kOrdThreads / 2) + q * (kOrdThreads / 2);
      if (p >= nolder) continue;
      xr[q] = row_f; yr[q] = ord_sh + (size_t)(2 + p) * S; dst[q] = 2 + mvec + p;
    }
  }

  if (carry)
    for (int q = 0; q < 2; q++)
      if (dst[q] >= 0) acc[q] = red[dst[q]];           // the running sums of the ranks before this one
  // ---- first pass: the norm (F08:266-267); with everything resident also the sums on f alone, on the other threads ----
  double s = 0.0;
  if (phase == kOrdRows) {
    if (pending) s = sqrt(red[0]);                     // the GLOBAL sum d^2 of the norm rounds
    if (t == 0) sum_dd = red[0];
    __syncthreads();
  } else {
    double dd = (carry && t == 0) ? red[0] : 0.0;
    for (int64_t c0 = 0; c0 < n; c0 += chunk) {
      const int len = (int)(n - c0 < chunk ? n - c0 : chunk);
      if (!pending && !single) break;
      for (int i = t; i < len; i += kOrdThreads) {
        const double fi = f[c0 + i];
        row_f[i] = fi;
        row_w1[i] = w1[c0 + i] - fi;                 // d (F08:266; (-1)*f + w1 in F08V:237: same bits)
      }
      if (single) ord_load_older(ord_sh, S, vs, pw, nolder, c0, len);
      __syncthreads();
      if (t == 0 && pending) dd = ord_sum(dd, row_w1, row_w1, len);
      if (single && !on_w1)
        for (int q = 0; q < 2; q++)
          if (dst[q] >= 0) acc[q] = ord_sum(acc[q], xr[q], yr[q], len);
      if (!single) __syncthreads();
    }
    if (t == 0) sum_dd = dd;
    __syncthreads();
    if (pending) s = sqrt(sum_dd);
    if (phase == kOrdNorm) {                           // a norm round of a sharded update: red[0] and nothing else
      if (t == 0 && pending) red[0] = sum_dd;
      return;
    }
  }
  const bool normed = pending && s != 0.0;           // (s == 0: the scalar step relaxes, F08:268-275; the w1' sums are dead)
  const double rs = 1.0 / s;

  // ---- second pass: the sums on the ROUNDED w1' (and, if the vectors did not fit, those on f alone) ----
  if (single) {
    if (normed) {
      for (int i = t; i < (int)n; i += kOrdThreads) row_w1[i] = rcp ? rs * row_w1[i] : row_w1[i] / s;   // the value PB stores as w1'
      __syncthreads();
      if (on_w1)
        for (int q = 0; q < 2; q++)
          if (dst[q] >= 0) acc[q] = ord_sum(acc[q], xr[q], yr[q], (int)n);
    }
  } else {
    for (int64_t c0 = 0; c0 < n; c0 += chunk) {
      const int len = (int)(n - c0 < chunk ? n - c0 : chunk);
      for (int i = t; i < len; i += kOrdThreads) {
        const double fi = f[c0 + i];
        row_f[i] = fi;
        if (normed) {
          const double d = w1[c0 + i] - fi;
          row_w1[i] = rcp ? rs * d : d / s;          // the value PB stores as w1'
        }
      }
      ord_load_older(ord_sh, S, vs, pw, nolder, c0, len);
      __syncthreads();
      if (normed || !on_w1)
        for (int q = 0; q < 2; q++)
          if (dst[q] >= 0) acc[q] = ord_sum(acc[q], xr[q], yr[q], len);
      __syncthreads();
    }
  }
  // red[]: zero what this update does not cover (nothing stale reaches a later reader), then the sums
  __syncthreads();
  if (!carry) {                                        // (a continuing rank keeps the prefix of the sums it does not own: zeros)
    for (int i = t + (phase == kOrdRows ? 1 : 0); i < 2 + 2 * mvec; i += kOrdThreads) red[i] = 0.0;
  }
  __syncthreads();
  if (t == 0 && pending && phase == kOrdAll) red[0] = sum_dd;
  for (int q = 0; q < 2; q++)
    if (dst[q] >= 0 && (normed || !on_w1)) red[dst[q]] = acc[q];
}

// ---- The same sums, ONE WORKGROUP PER SUM (round 5, long vectors) ----------------------------------
// k_dots_ordered walks every sum on one compute unit and waits for each chunk's loads before it adds: 40 ns per element.
// A sequential sum is a chain of n dependent roundings whatever is done, but (1) the 2 + 2L sums of an update are
// independent chains once the norm is known, (2) the products are not part of any chain and (3) -- see chain_block_summary --
// while the running sum stays inside one binade its roundings are roundings to a FIXED grid, which is integer
// arithmetic and therefore associative.  Here sum c has workgroup c to itself: its eight wavefronts load the next group of
// 8 192 elements, round the products into LDS and summarise one block of 1 024 each, and wavefront 0 takes the group
// through the chain.  Two launches per update: set kChainNorm = the norm (block 0) and, with `with_f`, the sums on f alone (blocks
// 1..ub); set kChainRows = with s from red[0], <f,w1'> (block 0), the Gram row on the ROUNDED w1' (blocks 1..ub) and,
// with `with_f`, the sums on f alone (blocks ub+1..2ub) -- the sharded rounds of ordered_chain take the second form (their
// norm rounds hold red[0] only).  EVERY chain starts from the value red[] holds: the host zeroes red[] where no prefix of
// other ranks is to be continued (0 + p == p: the same bits as starting at 0).
constexpr int kChainWaves = 8;
constexpr int kChainThreads = 64 * kChainWaves;
constexpr int kChainLaneElems = 16;                               // consecutive elements of a block per lane
constexpr int kChainBlock = 64 * kChainLaneElems;                 // elements per step of the chain
constexpr int kChainGroupBlocks = kChainWaves;                    // one block of a group per wavefront
constexpr int kChainGroup = kChainGroupBlocks * kChainBlock;      // elements whose products are in LDS at a time
constexpr int kChainPerThread = kChainGroup / kChainThreads;
constexpr int kChainRow = 2 * 64 + 4;                             // doubles between the PAIR rows of a block in LDS: row k holds
                                                                  // elements 2k, 2k+1 of every lane, lane after lane -- the
                                                                  // lanes of a wavefront read 16 bytes each, side by side
constexpr int kChainBlockLds = (kChainLaneElems / 2) * kChainRow;
constexpr int kChainGroupLds = kChainGroupBlocks * kChainBlockLds;
constexpr size_t kChainLdsBytes = sizeof(double) * kChainGroupLds;   // (dynamic: beyond the 64 KiB of static LDS)
enum { kChainNorm = 0, kChainRows = 1, kChainProbe = 2 };   // (probe: <f, probe> from red[2 + mvec], diagnostic entry)
enum { kChainKindNorm = 0, kChainKindFW1 = 1, kChainKindW1W = 2, kChainKindFW = 3 };
// where element i of a group lies in LDS
__device__ __forceinline__ int chain_idx(int i) {
  const int blk = i / kChainBlock, ib = i % kChainBlock;
  const int lane = ib / kChainLaneElems, j = ib % kChainLaneElems;
  return blk * kChainBlockLds + (j / 2) * kChainRow + 2 * lane + (j & 1);
}
// the 16 products of lane `lane` of a block, in order
__device__ __forceinline__ void chain_lane_read(double (&p)[kChainLaneElems], const double *blk, int lane) {
  using V2 = typename VecT<2>::type;
#pragma unroll
  for (int k = 0; k < kChainLaneElems / 2; k++) {
    const V2 v = *reinterpret_cast<const V2 *>(blk + k * kChainRow + 2 * lane);
    p[2 * k] = v.x;
    p[2 * k + 1] = v.y;
  }
}

// a + p[0] + p[1] + ... + p[len-1] in THAT order, one rounding per addition: the chain as it stands (every lane does the
// same additions on the same LDS words: no divergence, the sum stays wave-uniform).  One lane's worth (16 products) per
// step, read while the additions of the step before wait for one another, two steps per trip (no register copies): the
// loop is the chain of dependent v_add_f64 and little else (2.3 ns each, tools/micro/dep_add.hip).
__device__ __forceinline__ double chain_block_serial(double a, const double *blk, int len, int g = 0) {   // (from lane g on)
#pragma clang fp contract(off)
  const int ng = len / kChainLaneElems;                 // whole lanes
  if (g < ng) {
    double A[kChainLaneElems], B[kChainLaneElems];
    chain_lane_read(A, blk, g);
    for (; g + 2 <= ng; g += 2) {
      chain_lane_read(B, blk, g + 1);
      __builtin_amdgcn_sched_barrier(0);                 // (the reads of the NEXT lane go out before this lane's additions)
#pragma unroll
      for (int j = 0; j < kChainLaneElems; j++) a = a + A[j];
      __builtin_amdgcn_sched_barrier(0);
      chain_lane_read(A, blk, g + 3 <= ng ? g + 2 : g);   // (the last trip re-reads: the loads stay unconditional)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < kChainLaneElems; j++) a = a + B[j];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (g < ng) {                                        // (A holds lane g whenever a whole lane remains)
#pragma unroll
      for (int j = 0; j < kChainLaneElems; j++) a = a + A[j];
      g++;
    }
  }
  for (int i = g * kChainLaneElems; i < len; i++) a = a + blk[chain_idx(i)];
  return a;
}

// reductions and one scan over the 64 lanes through DPP (row shifts, then the row broadcasts of gfx9)
template <int CTRL, int RM> __device__ __forceinline__ float dpp_f32(float x, float old) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, RM, 0xf, false));
}
template <int CTRL, int RM> __device__ __forceinline__ int dpp_i32(int x, int old) {
  return __builtin_amdgcn_update_dpp(old, x, CTRL, RM, 0xf, false);
}
template <int CTRL, int RM> __device__ __forceinline__ double dpp_f64(double x, double old) {
  union { double d; int u[2]; } a, o, r;
  a.d = x; o.d = old;
  r.u[0] = __builtin_amdgcn_update_dpp(o.u[0], a.u[0], CTRL, RM, 0xf, false);
  r.u[1] = __builtin_amdgcn_update_dpp(o.u[1], a.u[1], CTRL, RM, 0xf, false);
  return r.d;
}
// lane l: the operation over lanes 0..l  (OP 0: +, 1: min, 2: max; `idn` the operation's identity: what lanes without a
// source take)
#define NKA_WAVE_SCAN(T, DPP)                                                                        \
  x = op(x, DPP<0x111, 0xf>(x, idn)); x = op(x, DPP<0x112, 0xf>(x, idn)); x = op(x, DPP<0x114, 0xf>(x, idn)); \
  x = op(x, DPP<0x118, 0xf>(x, idn)); x = op(x, DPP<0x142, 0xa>(x, idn)); x = op(x, DPP<0x143, 0xc>(x, idn)); \
  return x;
template <int OP> __device__ __forceinline__ float wave_scan_f32(float x, const float idn) {
  auto op = [](float a, float b) { return OP == 0 ? a + b : OP == 1 ? fminf(a, b) : fmaxf(a, b); };
  NKA_WAVE_SCAN(float, dpp_f32)
}
__device__ __forceinline__ int wave_scan_add_i32(int x) {
  auto op = [](int a, int b) { return a + b; };
  const int idn = 0;
  NKA_WAVE_SCAN(int, dpp_i32)
}
__device__ __forceinline__ double wave_scan_add_f64(double x) {
  auto op = [](double a, double b) { return a + b; };
  const double idn = 0.0;
  NKA_WAVE_SCAN(double, dpp_f64)
}
#undef NKA_WAVE_SCAN

// One step of a chain over a FULL block of products without walking it element after element.
// While  2^e <= |a| < 2^(e+1)  every representable neighbour of the running sum is a multiple of u = 2^(e-52), so
// fl(a + p) = u * (S + R(p/u)) with S = |a|/u an integer in [2^52, 2^53) and R the rounding of t = p/u to an integer,
// halves going to whichever neighbour makes S + R EVEN (round-to-nearest-even acts on the sum's significand).  The
// additions of integers are exact and associative; the only thing a step inherits from its predecessors is the PARITY of
// S, and only a halfway case reads it (after which the sum is even whatever it was).  So each lane takes 16 consecutive
// products: r = rne(t) (t + 1.5*2^52 - 1.5*2^52), the halfway flag |t - r| == 0.5, the plain sum of the r's, its own
// parity map and the corrections (+-1) a halfway case owes under either incoming parity; ballots carry the parity from
// lane to lane and one prefix sum places every lane's excursion.  That SUMMARY of a block depends on the running sum only
// through its sign and exponent (chain_block_summary), so the wavefronts of the workgroup take one block each under the
// exponent the group starts with; wavefront 0 then walks the summaries (chain_block_apply): a block is accepted iff it
// was summarised under the sum's present sign and exponent, every prefix provably stays inside the binade and every
// lane's sum of |r| < 2^51 (r exact, lane sums exact; NaN and Inf fail the comparison).  The prefix bounds are kept in SINGLE precision,
// rounded to nearest: they are off by < 2^33 units, and the acceptance window leaves 2^34 units (2^-18 of the binade)
// at either end -- which also covers the corrections (<= 1024) and the one inexact case (sums beyond 2^53 are only ever
// formed in blocks that leave the window by far more than their error).  Otherwise the block is summarised again under
// the present exponent or, failing that, walked (chain_block_serial).  Same bits as the walk by construction;
// tests/test_chain_sums_gpu.py holds the two to each other and to numpy's sequential accumulate on adversarial inputs
// (halfway cases under both parities, binade crossings, cancellation, zeros, subnormals, overflow, NaN).
// what one lane makes of its 16 consecutive products under the scale of the sum's binade
struct ChainLane {
  double base, absl;         // sum of the r's; sum of their magnitudes (< 2^51: every r and every partial sum exact)
  double pmin, pmax;         // least / greatest prefix sum inside the lane, the empty one (0) included
  int par, differ;           // parity of the lane's sum if it starts even; whether starting odd still flips it (no halfway case met)
  int adj0, adj1;            // corrections the halfway cases owe if the lane starts even / odd
};
__device__ __forceinline__ ChainLane chain_lane_pass(const double (&pl)[kChainLaneElems], double scale) {
#pragma clang fp contract(off)
  constexpr double M = 6755399441055744.0;                         // 1.5 * 2^52
  ChainLane ln;
  ln.base = ln.absl = ln.pmin = ln.pmax = 0.0;
  ln.differ = 1; ln.adj0 = ln.adj1 = 0;
  int parw = 0;
  bool halfway = false;
#pragma unroll
  for (int j = 0; j < kChainLaneElems; j++) {
    const double t = pl[j] * scale;                                // exact (a power of two), |t| tiny if it underflows
    const double tm = t + M;                                       // rounds t to an integer, halves to even
    const double r = tm - M;
    const double diff = t - r;                                     // exact
    halfway |= fabs(diff) == 0.5;
    parw ^= __double2loint(tm);
    ln.base = ln.base + r;
    ln.absl = ln.absl + fabs(r);
    ln.pmin = fmin(ln.pmin, ln.base);
    ln.pmax = fmax(ln.pmax, ln.base);
  }
  if (__any(halfway)) {                                            // the parity bookkeeping in its own pass
    parw = 0;
#pragma unroll
    for (int j = 0; j < kChainLaneElems; j++) {
      const double t = pl[j] * scale;
      const double tm = t + M;
      const double diff = t - (tm - M);
      if (fabs(diff) == 0.5) {                                     // halfway: r is the EVEN neighbour of t, r + 2 diff the odd one
        const int tau = diff > 0.0 ? 1 : -1;
        // r being even, S + r has the parity of S: an odd S takes the other neighbour, and the sum is even either way
        if (parw & 1) ln.adj0 += tau;
        if ((parw ^ ln.differ) & 1) ln.adj1 += tau;
        parw = 0; ln.differ = 0;
      } else {
        parw ^= __double2loint(tm);
      }
    }
  }
  ln.par = parw & 1;
  return ln;
}
// the parity each lane starts from if the block starts EVEN (q), and whether a block starting odd flips it (no halfway
// case in any lane before this one)
__device__ __forceinline__ void chain_lane_parity(const ChainLane &ln, int lane, int &q, bool &flips) {
  const unsigned long long T = __ballot(ln.differ == 0), A = __ballot(ln.par);
  const unsigned long long lt = (1ull << lane) - 1ull, Tl = T & lt;
  flips = Tl == 0;
  if (flips) q = __popcll(A & lt) & 1;
  else {
    const int h = 63 - __clzll(Tl);                                // the last lane before this one that met a halfway case
    q = __popcll(A & lt & ~((1ull << h) - 1ull)) & 1;
  }
}

struct ChainSummary {
  double total;              // sum of the r's
  float gmin, gmax;          // least / greatest prefix bound
  int adj;                   // corrections if S starts even (low half) / odd (high half), each biased by kChainAdjBias
  int hi;                    // sign and exponent word the summary assumed; 0: none, or a product out of range
};
constexpr int kChainAdjBias = 64 * kChainLaneElems;
__device__ __forceinline__ bool chain_scalable(double a) {
  const int ef = (__double2hiint(a) >> 20) & 0x7ff;
  return ef >= 1023 - 900 && ef <= 1023 + 900;                    // not zero, subnormal, Inf, NaN; scale factors in range
}
__device__ __forceinline__ ChainSummary chain_block_summary(double a, const double *blk) {
#pragma clang fp contract(off)
  ChainSummary sm;
  sm.hi = 0; sm.total = 0.0; sm.gmin = sm.gmax = 0.f; sm.adj = 0;
  if (!chain_scalable(a)) return sm;
  const int lane = threadIdx.x & 63;
  const int hi = __double2hiint(a);
  const int e = ((hi >> 20) & 0x7ff) - 1023;
  const double scale = __hiloint2double((hi & (int)0x80000000) | ((1023 + 52 - e) << 20), 0);    // +-2^(52-e): S > 0
  double pl[kChainLaneElems];
  chain_lane_read(pl, blk, lane);
  const ChainLane ln = chain_lane_pass(pl, scale);
  const bool bad = !(ln.absl < 0x1p51);                            // some |t| >= 2^51 / Inf / NaN: r = rne(t) and the lane's sums are exact below that
  // where the lane's excursion lies: the prefix before it + its own least / greatest prefix, in single precision
  const float basef = (float)ln.base;
  const float exclf = wave_scan_f32<0>(basef, 0.f) - basef;
  const float lo = wave_scan_f32<1>(exclf + (float)ln.pmin, __builtin_inff());
  const float up = wave_scan_f32<2>(exclf + (float)ln.pmax, -__builtin_inff());
  int q;
  bool flips;
  chain_lane_parity(ln, lane, q, flips);
  const int qo = flips ? q ^ 1 : q;
  const int packed = ((q ? ln.adj1 : ln.adj0) + kChainLaneElems) | (((qo ? ln.adj1 : ln.adj0) + kChainLaneElems) << 16);
  const int adjs = wave_scan_add_i32(packed);
  const double tot = wave_scan_add_f64(ln.base);
  sm.total = readlane_f64(tot, 63);
  sm.gmin = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo), 63));
  sm.gmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(up), 63));
  sm.adj = __builtin_amdgcn_readlane(adjs, 63);
  sm.hi = __any(bad) ? 0 : (hi & (int)0xfff00000);
  return sm;
}
// The running sum through summarised blocks.  S = |a| / u as long as blocks are accepted (ChainRun); leaving that form
// gives the double back.
struct ChainRun {
  double S, unscale;
  int hi;                    // sign and exponent word of the sum S stands for; 0: none (a is authoritative)
};
__device__ __forceinline__ void chain_run_enter(ChainRun &run, double a) {
  run.hi = 0;
  if (!chain_scalable(a)) return;
  const int hi = __double2hiint(a);
  const int e = ((hi >> 20) & 0x7ff) - 1023;
  const double scale = __hiloint2double((hi & (int)0x80000000) | ((1023 + 52 - e) << 20), 0);
  run.unscale = __hiloint2double((hi & (int)0x80000000) | ((1023 - 52 + e) << 20), 0);
  run.S = a * scale;                                               // exact, an integer in [2^52, 2^53)
  run.hi = hi & (int)0xfff00000;
}
__device__ __forceinline__ bool chain_block_apply(ChainRun &run, const ChainSummary &sm) {
#pragma clang fp contract(off)
  if (sm.hi == 0 || run.hi != sm.hi) return false;
  constexpr double kEdge = 0x1p34;
  if (!(run.S + (double)sm.gmin >= 0x1p52 + kEdge) || !(run.S + (double)sm.gmax <= 0x1p53 - kEdge)) return false;
  const int odd = __double2loint(run.S) & 1;                       // the parity of S: the last bit of the significand
  const int adj = ((odd ? sm.adj >> 16 : sm.adj) & 0xffff) - kChainAdjBias;
  run.S = run.S + (sm.total + (double)adj);
  return true;
}

// The loop of one sum: `load(g0)` brings the operands of the group that starts at element g0 into the caller's registers
// (wavefront w: block w of the group), `store()` rounds their products into the wavefront's block of `prod`.  Returns the
// sum (valid in thread 0).  The whole workgroup calls it.
struct ChainStamps {
#ifdef NKA_CHAIN_STAMPS
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // 10 ns ticks of wavefront 0: load issue, summary, wait, apply, wait, store
  unsigned long long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // blocks: in a run / on their own / summarised again / walked
#endif
};
template <class Load, class Store>
__device__ __forceinline__ double chain_drive(double a, int64_t n, double *prod, ChainSummary *summ, double *sh_a_p, int walk,
                                              ChainStamps &stamps, Load load, Store store) {
#pragma clang fp contract(off)
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  double *myblk = prod + wave * kChainBlockLds;
  if (t == 0) *sh_a_p = a;
  if (n > 0) { load(0); store(); }
  __syncthreads();
#ifdef NKA_CHAIN_STAMPS
  unsigned long long tk = wall_clock64(), tn;
#define NKA_CHAIN_STAMP(i) tn = wall_clock64(); stamps.st[i] += tn - tk; tk = tn;
#define NKA_CHAIN_COUNT(i, v) stamps.cnt[i] += (v);
#else
#define NKA_CHAIN_STAMP(i)
#define NKA_CHAIN_COUNT(i, v)
#endif
  for (int64_t g0 = 0; g0 < n; g0 += kChainGroup) {
    const bool more = g0 + kChainGroup < n;
    if (more) load(g0 + kChainGroup);                  // in flight while this group goes through the chain
    NKA_CHAIN_STAMP(0)
    const int glen = (int)(n - g0 < kChainGroup ? n - g0 : kChainGroup);
    const int nfull = glen / kChainBlock, nblk = (glen + kChainBlock - 1) / kChainBlock;
    // every wavefront summarises its block under the sign and exponent the group starts with ...
    const double a0 = *sh_a_p;
    if (wave < nfull && !walk) {
      const ChainSummary sm = chain_block_summary(a0, myblk);
      if (lane == 0) summ[wave] = sm;
    }
    NKA_CHAIN_STAMP(1)
    __syncthreads();
    NKA_CHAIN_STAMP(2)
    // ... and wavefront 0 takes the running sum through them: lane k holds the summary of block k
    if (wave == 0) {
      a = a0;
      ChainRun run;
      chain_run_enter(run, a);
      ChainSummary mine = summ[lane < kChainGroupBlocks ? lane : 0];
      int k = 0;
      while (k < nblk) {
        if (run.hi != 0 && !walk && k < nfull) {
          // every block from k on that applies whatever the parity of the sum: their totals as one prefix sum, each
          // checked against the sum it would start from; the longest run of acceptable blocks goes in at once
          const bool cand = lane >= k && lane < nfull;
          const int ae = (mine.adj & 0xffff) - kChainAdjBias, ao = ((mine.adj >> 16) & 0xffff) - kChainAdjBias;
          const bool usable = cand && mine.hi == run.hi;
          // the parity each block starts from, if every block before it goes in: block i flips it by the parity of its
          // total plus the correction it takes under the parity it meets (a short scalar chain over two bit masks)
          const int tpar = __double2loint(fabs(mine.total) + 0x1p52) & 1;     // (|total| < 2^52 in any block that goes in)
          const unsigned pe = (unsigned)__ballot(usable && ((tpar ^ ae) & 1)), po = (unsigned)__ballot(usable && ((tpar ^ ao) & 1));
          unsigned odd = 0;
          {
            unsigned p = (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(run.S)) & 1u;   // (scalar: the chain runs on the SALU)
            for (int i = k; i < nfull; i++) {
              odd |= p << i;
              p ^= ((p ? po : pe) >> i) & 1u;
            }
          }
          const double tk_ = usable ? mine.total + (double)(((odd >> lane) & 1u) ? ao : ae) : 0.0;
          double incl = tk_;
          incl = incl + dpp_f64<0x111, 0xf>(incl, 0.0);
          incl = incl + dpp_f64<0x112, 0xf>(incl, 0.0);
          incl = incl + dpp_f64<0x114, 0xf>(incl, 0.0);
          const double Sk = run.S + (incl - tk_);
          constexpr double kEdge = 0x1p34;
          const bool ok = usable && (Sk + (double)mine.gmin >= 0x1p52 + kEdge) && (Sk + (double)mine.gmax <= 0x1p53 - kEdge);
          const unsigned long long need = ((1ull << nfull) - 1ull) & ~((1ull << k) - 1ull);
          const unsigned long long failm = need & ~__ballot(ok);
          const int F = failm ? __ffsll((long long)failm) - 1 : nfull;
          if (F > k) {
            run.S = run.S + readlane_f64(incl, F - 1);
            NKA_CHAIN_COUNT(0, F - k)
            k = F;
            continue;
          }
        }
        // block k on its own: under the parity of the sum, or summarised again under its present exponent, or walked
        const double *bk = prod + k * kChainBlockLds;
        const int len = glen - k * kChainBlock < kChainBlock ? glen - k * kChainBlock : kChainBlock;
        bool done = false;
        if (len == kChainBlock && !walk) {
          ChainSummary sm = summ[k];
          done = chain_block_apply(run, sm);
          if (done) { NKA_CHAIN_COUNT(1, 1) }
          if (!done && run.hi != 0 && run.hi != sm.hi) {           // another binade by now: summarise under the present one
            sm = chain_block_summary(run.S * run.unscale, bk);
            done = chain_block_apply(run, sm);
            if (done) { NKA_CHAIN_COUNT(2, 1) }
          }
        }
        if (!done) {
          if (run.hi != 0) a = run.S * run.unscale;
          if (len == kChainBlock && !walk) { NKA_CHAIN_COUNT(3, 1) }
          // (walked whole: accepting its first lanes and taking the rest again under the next exponent was measured -- a
          //  sum that meets an end of its binade hovers there, a round costs what walking 14 lanes costs and gained 5: a loss)
          a = chain_block_serial(a, bk, len);
          chain_run_enter(run, a);
        }
        k++;
      }
      if (run.hi != 0) a = run.S * run.unscale;
      if (t == 0) *sh_a_p = a;
    }
    NKA_CHAIN_STAMP(3)
    __syncthreads();
    NKA_CHAIN_STAMP(4)
    if (more) store();                                 // (each wavefront into its own block, which it alone summarises)
    NKA_CHAIN_STAMP(5)
  }
  return a;
#undef NKA_CHAIN_STAMP
#undef NKA_CHAIN_COUNT
}

// One sum of an update as the chain kernels see it: which vectors, which rounding of their product, where the sum goes.
struct ChainSum {
  int kind, dst;             // kChainKind...; index into red[].  kind < 0: this sum does not exist in this update
  const double *f, *w1, *wk; // f; the pending w (d = w1 - f); the older w of the sum (kinds W1W, FW)
  double s, rs;              // the norm of d and its reciprocal (kinds FW1, W1W)
  int rcp;                   // w1' = (1/s) * d (vector flavour) instead of d / s
  int64_t n;
  bool vec16;                // every base address allows 16-byte loads
};
// sum number b of a launch: set kChainNorm = the norm (b = 0) and, with `with_f`, the sums on f alone (b = 1..ub);
// set kChainRows = <f,w1'> (b = 0), the Gram row on the rounded w1' (b = 1..ub) and, with `with_f`, the sums on f alone
// (b = ub+1..2ub), s from red[0]; set kChainProbe = <f, probe> into red[2 + mvec] (diagnostic entry)
__device__ __forceinline__ ChainSum chain_decode(const Ctl &ctl, const Vecs &vs, const double *f, int rcp, int set, int with_f,
                                                 int ub, int b, const double *probe) {
  ChainSum cs;
  const int pending = ctl.ic[IC_PLAN_PENDING];
  const int nolder = ctl.ic[IC_PLAN_NOLDER];
  const int mvec = ctl.mvec;
  const long long *pw = ctl.plan_w();
  cs.kind = -1; cs.dst = 0;
  cs.f = f; cs.w1 = pending ? vs.w + ctl.pc[PC_FIRST_W] : f; cs.wk = f;
  cs.s = 0.0; cs.rcp = rcp; cs.n = vs.n;
  if (set == kChainProbe) {
    cs.kind = kChainKindFW; cs.dst = 2 + mvec; cs.wk = probe;
  } else if (set == kChainNorm) {
    if (b == 0) {
      if (pending) { cs.kind = kChainKindNorm; cs.dst = 0; }
    } else {
      const int p = b - 1;
      if (with_f && p < nolder) { cs.kind = kChainKindFW; cs.dst = 2 + mvec + p; cs.wk = vs.w + pw[p]; }
    }
  } else {
    if (pending) cs.s = sqrt(ctl.red()[0]);           // the GLOBAL sum d^2 (F08:267)
    const bool normed = pending && cs.s != 0.0;       // (s == 0: the scalar step relaxes, F08:268-275; the w1' sums are dead)
    if (b == 0) {
      if (normed) { cs.kind = kChainKindFW1; cs.dst = 1; }
    } else if (b <= ub) {
      const int k = b - 1;
      if (normed && k < nolder) { cs.kind = kChainKindW1W; cs.dst = 2 + k; cs.wk = vs.w + pw[k]; }
    } else {
      const int p = b - 1 - ub;
      if (with_f && p < nolder) { cs.kind = kChainKindFW; cs.dst = 2 + mvec + p; cs.wk = vs.w + pw[p]; }
    }
  }
  cs.rs = 1.0 / cs.s;
  cs.vec16 = ((reinterpret_cast<uintptr_t>(cs.f) | reinterpret_cast<uintptr_t>(cs.w1) | reinterpret_cast<uintptr_t>(cs.wk)) & 15) == 0;
  return cs;
}
// The operands of one block (1024 elements from e0 on) in a wavefront's registers: pair j*64 + lane of the block per load,
// i.e. 1 KiB per wave instruction ...
constexpr int kChainPairs = kChainLaneElems / 2;       // 16-byte loads per thread and vector
static_assert(kChainPairs == 8, "the pair mapping of chain_load_block / chain_store_block assumes 16 elements per lane");
// xf = f; xb = the SECOND operand of the kind -- the older w (kind FW) or the pending w1 (every other kind); xc = the older w
// of kind W1W.  (Round 6: three arrays named after the vectors, each written under its own branch, made the compiler sink the
// stores of two branches into one store through a pointer phi -- the arrays then lived partly in scratch, 48-80 bytes per
// lane in k_chain_sums / k_chain_blocks / k_chain_apply.  One destination per load, the ADDRESS selected instead.)
struct ChainBlockRegs {
  typename VecT<2>::type xf[kChainPairs], xb[kChainPairs], xc[kChainPairs];
};
__device__ __forceinline__ void chain_load_block(const ChainSum &cs, ChainBlockRegs &r, int64_t e0, int lane, bool full) {
  using V2 = typename VecT<2>::type;
  const int64_t n = cs.n;
  const bool vec16 = cs.vec16;
  auto ldpair = [&](const double *p, int64_t i) -> V2 {
    V2 v;
    if (full && vec16) v = *reinterpret_cast<const V2 *>(p + i);
    else if (full) { v.x = p[i]; v.y = p[i + 1]; }
    else { v.x = i < n ? p[i] : 0.0; v.y = i + 1 < n ? p[i + 1] : 0.0; }
    return v;
  };
  const int64_t i0 = e0 + 2 * lane;
  const double *const second = cs.kind == kChainKindFW ? cs.wk : cs.w1;
#pragma unroll
  for (int j = 0; j < kChainPairs; j++) { r.xf[j] = ldpair(cs.f, i0 + j * 128); r.xb[j] = ldpair(second, i0 + j * 128); }
  if (cs.kind == kChainKindW1W) {
#pragma unroll
    for (int j = 0; j < kChainPairs; j++) r.xc[j] = ldpair(cs.wk, i0 + j * 128);
  }
}
// ... and their rounded products where the lane that owns them reads them (blk: the block's kChainBlockLds doubles of LDS)
__device__ __forceinline__ void chain_store_block(const ChainSum &cs, const ChainBlockRegs &r, double *blk, int lane) {
#pragma clang fp contract(off)      // products and additions stay separate roundings whatever the build's flags
  using V2 = typename VecT<2>::type;
  const double s = cs.s, rs = cs.rs;
#pragma unroll
  for (int j = 0; j < kChainPairs; j++) {
    V2 p;
    if (cs.kind == kChainKindFW) { p.x = r.xf[j].x * r.xb[j].x; p.y = r.xf[j].y * r.xb[j].y; }
    else {
      const double d0 = r.xb[j].x - r.xf[j].x, d1 = r.xb[j].y - r.xf[j].y;   // F08:266 ((-1)*f + w1 in F08V:237: same bits)
      if (cs.kind == kChainKindNorm) { p.x = d0 * d0; p.y = d1 * d1; }
      else {
        const double n0 = cs.rcp ? rs * d0 : d0 / s, n1 = cs.rcp ? rs * d1 : d1 / s;   // the value PB stores as w1' (F08:283; F08V:256)
        if (cs.kind == kChainKindFW1) { p.x = r.xf[j].x * n0; p.y = r.xf[j].y * n1; }
        else { p.x = n0 * r.xc[j].x; p.y = n1 * r.xc[j].y; }
      }
    }
    // pair j*64 + lane of the block = elements 2 (j*64 + lane), +1: lane (j*64 + lane) / 8 of the chain, pair row lane % 8
    *reinterpret_cast<V2 *>(blk + (lane % kChainPairs) * kChainRow + 2 * (j * (64 / kChainPairs) + lane / kChainPairs)) = p;
  }
}

static __global__ __launch_bounds__(kChainThreads) __attribute__((unused)) void k_chain_sums(Ctl ctl, Vecs vs,
                                                                                              const double *__restrict__ f,
                                                                                              int rcp, int set, int with_f, int ub,
                                                                                              int walk, const double *probe) {
  extern __shared__ __attribute__((aligned(16))) double prod[];   // kChainLdsBytes
  __shared__ ChainSummary summ[kChainGroupBlocks];
  __shared__ double sh_a;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const ChainSum cs = chain_decode(ctl, vs, f, rcp, set, with_f, ub, blockIdx.x, probe);
  if (cs.kind < 0) return;
  // wavefront w loads, multiplies and summarises block w of a group (elements [1024 w, 1024 w + 1024))
  ChainBlockRegs regs;
  double *myblk = prod + wave * kChainBlockLds;
  auto load = [&](int64_t g0) { chain_load_block(cs, regs, g0 + wave * kChainBlock, lane, g0 + kChainGroup <= cs.n); };
  auto store = [&]() { chain_store_block(cs, regs, myblk, lane); };
  double *red = ctl.red();
  ChainStamps stamps;
  const double a = chain_drive(red[cs.dst], cs.n, prod, summ, &sh_a, walk, stamps, load, store);
  if (t == 0) red[cs.dst] = a;
#ifdef NKA_CHAIN_STAMPS
  if (t == 0 && set == kChainProbe)
    for (int i = 0; i < 8; i++) { ctl.stamps()[i] = (double)stamps.st[i]; ctl.stamps()[8 + i] = (double)stamps.cnt[i]; }
  // (10 ns ticks: load issue, summary, wait, apply, wait, store; then the block counts)
#endif
}

// ---- The same sums, MANY compute units per sum (round 5, the longest vectors) ------------------------------------
// k_chain_sums gives a sum one compute unit: its blocks are summarised eight at a time and the summaries applied in
// between.  But a block's summary needs only the SIGN AND EXPONENT of the running sum at its start -- and those follow from
// an ordinary blocked prefix sum (off by rounding noise, i.e. wrong only when the sum is within ~1e-13 of a power of two,
// which the apply step notices).  So the summaries of ALL blocks of ALL sums are made by the whole device:
//   k_chain_blocks(mode 0)   one wavefront per (block, sum): the block's products, summed any way -> pred[sum][block]
//   k_chain_predict          per sum: exclusive prefix of those, from red[dst] -> the predicted running sum at each block
//   k_chain_blocks(mode 1)   one wavefront per (block, sum): the summary under the predicted sign and exponent
//   k_chain_apply            per sum, ONE wavefront: 64 summaries at a time -- one scalar parity chain, one prefix sum, each
//                            block checked against the sum it would start from, the longest run of acceptable blocks goes
//                            in at once; a block that does not is loaded, summarised again if the exponent was
//                            mispredicted, else walked (chain_block_serial), and the run goes on behind it.
// Same functions, same acceptance rule, same bits as k_chain_sums; what remains sequential is ~10 ns per accepted block and
// the walk of the blocks that meet an end of their binade.
static __global__ __launch_bounds__(kChainThreads) __attribute__((unused)) void k_chain_blocks(Ctl ctl, Vecs vs,
                                                                                                const double *__restrict__ f,
                                                                                                int rcp, int set, int with_f, int ub,
                                                                                                int nsum, long long nfull,
                                                                                                double *__restrict__ pred,
                                                                                                ChainSummary *__restrict__ summ,
                                                                                                int mode, const double *probe) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double prod[];   // kChainLdsBytes: one block per wavefront
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const long long item = (long long)blockIdx.x * kChainWaves + wave;      // (no barrier in this kernel: wavefronts are on their own)
  if (item >= nfull * nsum) return;
  const int b = (int)(item % nsum);                                        // the sums of one block side by side: f and w1 come from L2
  const long long blk = item / nsum;
  const ChainSum cs = chain_decode(ctl, vs, f, rcp, set, with_f, ub, b, probe);
  if (cs.kind < 0) return;
  ChainBlockRegs regs;
  double *myblk = prod + wave * kChainBlockLds;
  chain_load_block(cs, regs, blk * kChainBlock, lane, true);
  chain_store_block(cs, regs, myblk, lane);
  if (mode == 0) {
    double pl[kChainLaneElems];
    chain_lane_read(pl, myblk, lane);
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < kChainLaneElems; j++) acc = acc + pl[j];
    const double tot = wave_scan_add_f64(acc);
    if (lane == 63) pred[(long long)b * nfull + blk] = tot;
  } else {
    const ChainSummary sm = chain_block_summary(pred[(long long)b * nfull + blk], myblk);
    if (lane == 0) summ[(long long)b * nfull + blk] = sm;
  }
}

constexpr int kChainPredictThreads = 256;
static __global__ __launch_bounds__(kChainPredictThreads) __attribute__((unused)) void k_chain_predict(Ctl ctl, Vecs vs,
                                                                                                        const double *f, int rcp,
                                                                                                        int set, int with_f, int ub,
                                                                                                        long long nfull,
                                                                                                        double *__restrict__ pred,
                                                                                                        const double *probe) {
  __shared__ double seg_sum[kChainPredictThreads];
  const int t = threadIdx.x, b = blockIdx.x;
  const ChainSum cs = chain_decode(ctl, vs, f, rcp, set, with_f, ub, b, probe);
  if (cs.kind < 0) return;
  double *p = pred + (long long)b * nfull;
  const long long seg = (nfull + kChainPredictThreads - 1) / kChainPredictThreads;
  const long long lo = t * seg, hi = lo + seg < nfull ? lo + seg : nfull;
  double acc = 0.0;
  for (long long i = lo; i < hi; i++) acc += p[i];
  seg_sum[t] = acc;
  __syncthreads();
  if (t == 0) {
    double run = ctl.red()[cs.dst];                    // the sum the chain starts from
    for (int i = 0; i < kChainPredictThreads; i++) { const double v = seg_sum[i]; seg_sum[i] = run; run += v; }
  }
  __syncthreads();
  double run = seg_sum[t];
  for (long long i = lo; i < hi; i++) { const double v = p[i]; p[i] = run; run += v; }
}

static __global__ __launch_bounds__(64) __attribute__((unused)) void k_chain_apply(Ctl ctl, Vecs vs, const double *__restrict__ f,
                                                                                  int rcp, int set, int with_f, int ub,
                                                                                  long long nfull,
                                                                                  const ChainSummary *__restrict__ summ, int walk,
                                                                                  const double *probe) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) double blk_lds[kChainBlockLds];
  const int lane = threadIdx.x, b = blockIdx.x;
  const ChainSum cs = chain_decode(ctl, vs, f, rcp, set, with_f, ub, b, probe);
  if (cs.kind < 0) return;
  double *red = ctl.red();
  double a = red[cs.dst];
  ChainRun run;
  chain_run_enter(run, a);
  ChainBlockRegs regs, ahead;                          // the operands of the block in hand; of the block behind it (see below)
  long long ahead_of = -1;                             // which block `ahead` holds
  const ChainSummary *mysum = summ + (long long)b * nfull;
  ChainSummary none;
  none.hi = 0; none.total = 0.0; none.gmin = none.gmax = 0.f; none.adj = 0;
#ifdef NKA_CHAIN_STAMPS
  unsigned long long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tw = 0, t0_;   // blocks: in runs / singly / summarised again / walked
#define NKA_APPLY_COUNT(i, v) cnt[i] += (v);
#else
#define NKA_APPLY_COUNT(i, v)
#endif
  ChainSummary next = (lane < nfull && !walk) ? mysum[lane] : none;       // (the summaries of the batch after this one are in
  for (long long k0 = 0; k0 < nfull; k0 += 64) {                          //  flight while this one is applied)
    const int nb = (int)(nfull - k0 < 64 ? nfull - k0 : 64);
    const ChainSummary mine = next;
    next = (k0 + 64 + lane < nfull && !walk) ? mysum[k0 + 64 + lane] : none;
    int k = 0;
    bool try_run = true;                               // (right behind a block that did not go in, its successor is tried alone first)
    while (k < nb) {
      if (run.hi != 0 && !walk && !try_run) {
        ChainSummary one;
        one.total = readlane_f64(mine.total, k);
        one.gmin = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.gmin), k));
        one.gmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.gmax), k));
        one.adj = __builtin_amdgcn_readlane(mine.adj, k);
        one.hi = __builtin_amdgcn_readlane(mine.hi, k);
        if (chain_block_apply(run, one)) { k++; try_run = true; NKA_APPLY_COUNT(1, 1) continue; }
      }
      if (run.hi != 0 && !walk && try_run) {
        const bool cand = lane >= k && lane < nb;
        const int ae = (mine.adj & 0xffff) - kChainAdjBias, ao = ((mine.adj >> 16) & 0xffff) - kChainAdjBias;
        const bool usable = cand && mine.hi == run.hi;
        // the parity each block starts from, if every block before it goes in (a scalar chain over two ballot masks)
        const int tpar = __double2loint(fabs(mine.total) + 0x1p52) & 1;     // (|total| < 2^52 in any block that goes in)
        const unsigned long long pe = __ballot(usable && ((tpar ^ ae) & 1)), po = __ballot(usable && ((tpar ^ ao) & 1));
        unsigned long long odd = 0;
        if (pe == po) {                                // (no block's flip depends on the parity it meets: a prefix XOR)
          unsigned long long x = pe;
          x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;
          odd = ((__builtin_amdgcn_readfirstlane(__double2loint(run.S)) & 1) ? ~0ull : 0ull) ^ (x << 1);
        } else {
          unsigned long long p = (unsigned long long)(__builtin_amdgcn_readfirstlane(__double2loint(run.S)) & 1);
          for (int i = k; i < nb; i++) {
            odd |= p << i;
            p ^= ((p ? po : pe) >> i) & 1ull;
          }
        }
        const double tk_ = usable ? mine.total + (double)(((odd >> lane) & 1ull) ? ao : ae) : 0.0;
        const double incl = wave_scan_add_f64(tk_);
        const double Sk = run.S + (incl - tk_);
        constexpr double kEdge = 0x1p34;
        const bool ok = usable && (Sk + (double)mine.gmin >= 0x1p52 + kEdge) && (Sk + (double)mine.gmax <= 0x1p53 - kEdge);
        const unsigned long long need = (nb == 64 ? ~0ull : (1ull << nb) - 1ull) & ~((1ull << k) - 1ull);
        const unsigned long long failm = need & ~__ballot(ok);
        const int F = failm ? __ffsll((long long)failm) - 1 : nb;
        if (F > k) {
          run.S = run.S + readlane_f64(incl, F - 1);
          NKA_APPLY_COUNT(0, F - k)
          k = F;
          continue;
        }
      }
      // block k0 + k on its own: its products into LDS; summarised under the sum's present exponent if the prediction
      // missed it, else (or failing that) walked.  A sum that meets an end of its binade stays there for a while: the
      // operands of the NEXT block are requested before this one is walked, so that its walk, if it comes to that, finds them.
      const long long kb = k0 + k;
      if (ahead_of == kb) regs = ahead;
      else chain_load_block(cs, regs, kb * kChainBlock, lane, true);
      if (kb + 1 < nfull) { chain_load_block(cs, ahead, (kb + 1) * kChainBlock, lane, true); ahead_of = kb + 1; }
      chain_store_block(cs, regs, blk_lds, lane);
      bool done = false;
      const int hi_k = __builtin_amdgcn_readlane(mine.hi, k);
      if (!walk && run.hi != 0 && run.hi != hi_k) {
        const ChainSummary sm = chain_block_summary(run.S * run.unscale, blk_lds);
        done = chain_block_apply(run, sm);
        if (done) { NKA_APPLY_COUNT(2, 1) }
      }
      if (!done) {
        NKA_APPLY_COUNT(3, 1)
#ifdef NKA_CHAIN_STAMPS
        t0_ = wall_clock64();
#endif
        if (run.hi != 0) a = run.S * run.unscale;
        a = chain_block_serial(a, blk_lds, kChainBlock);
        chain_run_enter(run, a);
#ifdef NKA_CHAIN_STAMPS
        tw += wall_clock64() - t0_;
#endif
      }
      k++;
      try_run = false;
    }
  }
  if (run.hi != 0) a = run.S * run.unscale;
  const long long e0 = nfull * kChainBlock;
  if (e0 < cs.n) {                                     // the last, partial block
    chain_load_block(cs, regs, e0, lane, false);
    chain_store_block(cs, regs, blk_lds, lane);
    a = chain_block_serial(a, blk_lds, (int)(cs.n - e0));
  }
  if (lane == 0) red[cs.dst] = a;
#ifdef NKA_CHAIN_STAMPS
  if (lane == 0 && set == kChainProbe) {
    for (int i = 0; i < 8; i++) ctl.stamps()[8 + i] = (double)cnt[i];
    ctl.stamps()[3] = (double)tw;                       // (10 ns ticks spent walking)
  }
#endif
#undef NKA_APPLY_COUNT
}

// ---- PB: normalise the pending pair, combine, and all five stores -----------------
// COMB 0: x/s          ; (f - c*w) + c*v       F08:282-283, 397
// COMB 1: (1/s)*x      ; ((-c)*w + c*v) + f    F08V:255-256 scale(1/s), :374 update3_
//                                              (grid_vector_type.F90:117,151)
// COMB 2: x/s          ; f + c*(v - w)         C .c:317-320, 423
// The k loop runs in list order with the reference's association, so given the
// same coefficients the result is bit-identical to the reference's k passes.
//
// COMPACT storage (COMB 2 only).  The C reference combines with the DIFFERENCE
// v_k - w_k (f += c*(v - w), .c:423).  For a normalised pair that difference
// never changes, so this flavour stores u_k = fl(v_k' - w_k') in the v array
// when the pair is normalised and reads ONE vector per pair ever after:
// f + c*u_k is bit-identical to the C statement, and PB reads k+2 vectors
// instead of 2k+1.  (The pending slot still holds the raw w = f_in, v = f_out.)
//
// MAXK (slot, coefficient) pairs per pass, fully unrolled: offsets and
// coefficients sit in SGPRs and all loads of a tile are issued together.  Pairs
// beyond the actual count re-read f and are not applied.  Pass 0 stores
// w_new = f_in and, if this update normalises (IC_NORMED), treats pair 0 -- the
// pending slot, still holding the raw previous f and update -- as
// w1' = (w1-f)/s, v1' = v1/s formed in registers and stored back.  The last pass
// stores v_new = f_out.
enum { kPbNoStoreW = 1, kPbNoStoreF = 2,     // `flags` of PB in an out-of-place update (nka_hip_accel_update_swap)
       kPbNotFirst = 8, kPbNotLast = 16,      // rolling-window PB over a list longer than kMaxPerPass: not the first / not the
                                              // last of its passes (enqueue_pb; in place only)
       kPbReverse = 4 };                      // rolling-window PB: walk the tiles from the END of the vectors, i.e. in the
                                              // reverse of PA's order (diagnostic builds only, see NKA_F_TEMPORAL)

template <int COMB>
__device__ __forceinline__ double comb1(double x, double c, double w, double v) {
  if (COMB == 0) return (x - c * w) + c * v;
  if (COMB == 1) return ((-c) * w + c * v) + x;
  return x + c * (v - w);
}

template <int MAXK, int VEC, int COMB>
__global__ __launch_bounds__(kBlock) void k_combine(Ctl ctl, Vecs vs, double *f, int pass, int last_pass, int flags) {
  using V = typename VecT<VEC>::type;
  constexpr bool RCP = (COMB == 1);
  constexpr bool COMPACT = (COMB == 2);
  constexpr int NW = COMPACT ? 1 : MAXK;   // w vectors loaded per tile
  const int G = gridDim.x;
  const int ncomb = ctl.ic[IC_NCOMB];
  double *wnew = vs.w + ctl.pc[PC_NEW_W], *vnew = vs.w + ctl.pc[PC_NEW_V];
  const long long *cw = ctl.comb_w(), *cv = ctl.comb_v();
  const double *cc = ctl.comb_c();
  const int base = pass * MAXK;
  // Out-of-place update (kPbNoStoreW / kPbNoStoreF): the caller's buffer f IS w_new and must keep f_in, and f_out goes
  // to v_new only -- so between the passes of a long list the running value lives in v_new, never in f.
  const bool oop = (flags & kPbNoStoreF) != 0;
  const double *src = (oop && pass > 0) ? vnew : f;
  const bool store_w = (pass == 0) && !(flags & kPbNoStoreW), store_v = (last_pass != 0) || oop;
  const bool store_f = !oop && (last_pass != 0 ? (ncomb > 0) : true);  // nothing to combine: f stays as it is
  const bool norm0 = (pass == 0) && ctl.ic[IC_NORMED];
  const double s = ctl.dc[DC_S];
  const double rs = 1.0 / s;
  if (pass == 0) list_word_publish(ctl, ncomb, flags & kPbNoStoreW);

  double *wk[MAXK], *vk[MAXK];
  double ck[MAXK];
#pragma unroll
  for (int j = 0; j < MAXK; j++) {
    const int k = base + j;
    const bool live = k < ncomb;
    wk[j] = live ? vs.w + cw[k] : f;
    vk[j] = live ? vs.w + cv[k] : f;
    ck[j] = cc[k];
  }

  const int64_t ntile = vs.n / (kBlock * VEC);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    const V fin = ld<VEC>(src + e);
    V wv[NW], vv[MAXK];
    if (!COMPACT || norm0) wv[0] = ld<VEC>(wk[0] + e); else wv[0] = fin;
#pragma unroll
    for (int j = 1; j < NW; j++) wv[j] = ld<VEC>(wk[j] + e);
#pragma unroll
    for (int j = 0; j < MAXK; j++) vv[j] = ld<VEC>(vk[j] + e);
    __builtin_amdgcn_sched_barrier(0);  // every load of the tile in flight before any arithmetic
    if (norm0) {
#pragma unroll
      for (int q = 0; q < VEC; q++) {
        const double d = ex(wv[0], q) - ex(fin, q);
        const double wn = RCP ? rs * d : d / s;
        const double vn = RCP ? rs * ex(vv[0], q) : ex(vv[0], q) / s;
        setc(wv[0], q, wn);
        setc(vv[0], q, COMPACT ? vn - wn : vn);
      }
      st(wk[0] + e, wv[0]);
      st(vk[0] + e, vv[0]);
    }
    V x = fin;
#pragma unroll
    for (int j = 0; j < MAXK; j++) {
      if (base + j < ncomb) {
#pragma unroll
        for (int q = 0; q < VEC; q++) {
          if (COMPACT) setc(x, q, ex(x, q) + ck[j] * ex(vv[j], q));
          else setc(x, q, comb1<COMB>(ex(x, q), ck[j], ex(wv[j < NW ? j : 0], q), ex(vv[j], q)));
        }
      }
    }
    if (store_w) st(wnew + e, fin);
    if (store_v) st(vnew + e, x);
    if (store_f) st(f + e, x);
  }
  if ((int)blockIdx.x == G - 1) {  // ragged tail, scalar
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < vs.n; i += kBlock) {
      const double fin = src[i];
      double x = fin;
#pragma unroll
      for (int j = 0; j < MAXK; j++) {
        if (base + j < ncomb) {
          double v = vk[j][i];
          double w = (!COMPACT || (j == 0 && norm0)) ? wk[j][i] : 0.0;
          if (j == 0 && norm0) {
            const double d = w - fin;
            w = RCP ? rs * d : d / s;
            v = RCP ? rs * v : v / s;
            if (COMPACT) v = v - w;
            wk[0][i] = w;
            vk[0][i] = v;
          }
          x = COMPACT ? x + ck[j] * v : comb1<COMB>(x, ck[j], w, v);
        }
      }
      if (store_w) wnew[i] = fin;
      if (store_v) vnew[i] = x;
      if (store_f) f[i] = x;
    }
  }
}

// ---- PB with a SMALL ROLLING WINDOW of loads ---------------------------------------
// tools/hbm_probe (mode m): a kernel reading 22 streams and writing 5 moves 5.5 TB/s with
// every load of a tile in flight, 5.7 software-pipelined, 5.9 when each wave keeps only a
// ring of FOUR loads in flight and re-issues a slot the moment it has been consumed
// (pure reads: 7.25 against 7.0 TB/s) -- fewer streams are open in the DRAMs at any
// moment, and the load issue never stops for the arithmetic or the stores.  Here the MAXK
// pairs of a tile go through a ring of W (pair j in slot j mod W; a consumed slot is
// re-loaded with pair j+W of this tile or pair j+W-MAXK of the block's next tile); f
// and, with compact storage, the raw w of the pending pair are requested one tile ahead.
// Same arithmetic in the same order => same bits as k_combine.  Single pass, VEC = 2.
//
// TILE TICKETS (`tickets` != nullptr).  With the static mapping (tile t -> block t mod G) the
// blocks of a mixed read/write pass drift apart -- by 5 % of the launch, i.e. ~40 tiles, at
// n = 1e8 -- and the chip then works on a ~40 MB window of each of the 27 streams at once.
// tools/hbm_probe (modes d, e, g; profiles/r02/hbm_probe_tile_tickets.txt) shows the same
// streams moving 8-14 % faster when every block takes its next tile from ONE global counter:
// the blocks then advance as a compact front (all end within 5 us of each other) and the DRAMs
// see one narrow window per stream.  A block's first two tiles are static (b, b + G); thread 0
// requests the tile after next with a returning atomic at the top of an iteration and publishes it
// in LDS at the end (one workgroup barrier per tile): the OTHER three waves never wait for the
// atomic, wave 0 does (see ticket_request).  `ng` counters (128 B apart), counter g serving the blocks with b % ng == g and
// the tiles = g (mod ng): a single counter saturates near 60-75 tickets/us, which short lists
// exceed.  The last block to finish resets the counters (a second counter, `done`), so a launch
// always finds them zero.  Elementwise pass: which block handles a tile changes no bit.
constexpr int kTicketStride = 32;                 // uint32 words between counters (128 B)
constexpr int kTicketGroupsMax = 8;
constexpr int kTicketWords = kTicketStride * (kTicketGroupsMax + 1);   // ng counters + `done`
constexpr unsigned kNoTicket = 0xffffffffu;

// thread 0: the tile after next of this block's group (returning atomic; the value is used a tile later).
// hipcc's atomic optimiser broadcasts the result with v_readfirstlane right behind the instruction, so
// wave 0 does wait for the atomic here (s_waitcnt vmcnt(0)); measured against an inline-asm request
// whose result is only read at the end of the tile, that costs 0-2 % of PB -- and the asm form needs a
// hand-counted s_waitcnt that turned out NOT to be safe: stores retire out of order with respect to the
// atomic, a short tile read its ticket too early (tests caught it).  The plain form stays.
__device__ __forceinline__ unsigned ticket_request(unsigned *group_counter, unsigned base, unsigned ng, unsigned grp) {
  return (atomicAdd(group_counter, 1u) + base) * ng + grp;
}
// end of a tile: thread 0 publishes what it was given, every thread learns the block's tile after next
// (two LDS words used alternately: a word is rewritten only after another barrier)
__device__ __forceinline__ int64_t ticket_publish(unsigned *s_next, unsigned &par, unsigned claimed, int64_t ntile) {
  if (threadIdx.x == 0) s_next[par] = claimed;
  __syncthreads();
  const unsigned nx = s_next[par];
  par ^= 1u;
  return nx == kNoTicket ? ntile : (int64_t)nx;
}
// end of the kernel: every ticket request of this block has returned; the block that arrives last
// resets the counters, so the next launch finds them zero
__device__ __forceinline__ void ticket_finish(unsigned *tickets, int ng, int G) {
  if (threadIdx.x != 0) return;
  unsigned *const done = tickets + kTicketGroupsMax * kTicketStride;
  if (atomicAdd(done, 1u) == (unsigned)G - 1u) {
    for (int g = 0; g < ng; g++) atomicExch(tickets + g * kTicketStride, 0u);
    atomicExch(done, 0u);
  }
}

// PASSES (round 5).  A list longer than kMaxPerPass pairs is combined by several launches of balanced exact widths, each on
// the pairs [base, base + MAXK) of the plan, f carrying the running value in between (the k loop of F08:395-399 cut into
// consecutive pieces: same statements in the same order, same bits).  The FIRST pass (no kPbNotFirst) normalises the pending
// pair -- pair 0 of the plan -- and stores w_new = f_in; the LAST (no kPbNotLast) stores v_new = f_out; every pass stores f.
template <int MAXK, int COMB, int W, int T = 1>
__global__ __launch_bounds__(kBlock) void k_combine_win(Ctl ctl, Vecs vs, double *f, unsigned *tickets, int ng, int flags,
                                                        int base) {
  // T = 16-byte pieces per thread, stream and tile (tile = 512*T elements, 4*T KiB per stream and
  // block): T = 2 halves the ticket rate, which is what lets SHORT lists use one counter (a
  // single counter saturates near 60-75 tickets/us; tools/hbm_probe mode i: 12 + 5 streams move
  // 6.3-6.4 TB/s with T = 2 and one counter against 5.7 with T = 1 and two, 5.4 static).
  constexpr int VEC = 2;
  using V = typename VecT<VEC>::type;
  constexpr bool RCP = (COMB == 1);
  constexpr bool COMPACT = (COMB == 2);
  constexpr int TILE = kBlock * VEC * T;
  static_assert(MAXK % W == 0, "the ring must divide the pairs of a tile");
  NKA_STAMP0(ctl, 14);
  __shared__ unsigned s_next[2];
  // The ragged tail (n mod TILE elements, scalar) has a block of its own, the LAST of the grid, launched only when
  // there is a tail: appended to the last tile block's work it made that block -- and so the launch -- one memory
  // round trip longer (2.3 us of a 13 us launch at n = 1e5).  Elementwise pass: who handles an element changes no bit.
  const int64_t ntile = vs.n / TILE;
  const bool has_tail = ntile * TILE < vs.n;
  const int G = (int)gridDim.x - (has_tail ? 1 : 0);       // tile blocks
  const bool tail_block = has_tail && (int)blockIdx.x == G;
  const int ncomb_all = ctl.ic[IC_NCOMB];
  const int ncomb = ncomb_all - base;                      // pairs of the plan from `base` on (this launch applies the first MAXK)
  double *wnew = vs.w + ctl.pc[PC_NEW_W], *vnew = vs.w + ctl.pc[PC_NEW_V];
  const long long *cw = ctl.comb_w() + base, *cv = ctl.comb_v() + base;
  const double *cc = ctl.comb_c() + base;
  // (uniform: out-of-place update.  The two scalar branches around the stores cost the in-place path nothing measurable:
  //  interleaved A/B against a build with unconditional stores, profiles/r04/ab_pb_flags.txt)
  const bool first_pass = !(flags & kPbNotFirst), last_pass = !(flags & kPbNotLast);
  const bool store_w = !(flags & kPbNoStoreW) && first_pass, store_f = !(flags & kPbNoStoreF), store_v = last_pass;
  const bool norm0 = first_pass && ctl.ic[IC_NORMED] != 0;
  const double s = ctl.dc[DC_S];
  const double rs = 1.0 / s;

  double *wk[MAXK], *vk[MAXK];
  double ck[MAXK];
  long long slw[MAXK], slv[MAXK];
#pragma unroll
  for (int j = 0; j < MAXK; j++) {     // all addresses and coefficients in one batch of scalar loads (see k_dots_win)
    if (!COMPACT || j == 0) slw[j] = cw[j];      // (compact storage reads w of the pending pair only)
    slv[j] = cv[j];
    ck[j] = cc[j];
  }
#pragma unroll
  for (int j = 0; j < MAXK; j++) {
    const bool live = j < ncomb;
    wk[j] = (live && (!COMPACT || j == 0)) ? vs.w + slw[COMPACT ? 0 : j] : f;
    vk[j] = live ? vs.w + slv[j] : f;
  }
  // compact storage reads w only for the pending pair that is normalised now
  const double *w0src = norm0 ? wk[0] : f;

  const int lane_off = threadIdx.x * VEC;                  // piece q of a tile starts q*512 elements further
#if NKA_DEAD_SLOT_TILE0
#define DEAD_OFF(live, off) ((live) ? (off) : (int64_t)lane_off)      // (dead ring slots: see NKA_DEAD_SLOT_TILE0)
#else
#define DEAD_OFF(live, off) (off)
#endif
  V finv[T], w0v[T], rw[COMPACT ? 1 : W][T], rv[W][T];
  // logical tile t -> the elements it covers (kPbReverse: counted from the end; elementwise pass, same bits either way)
  const bool rev = (flags & kPbReverse) != 0;
  const int64_t tlast = ntile - 1;
#define TILE_ELEM(t) (((rev) ? tlast - (t) : (t)) * TILE + lane_off)
  // pair 0 is the pending pair, whose raw w PA has just read (NKA_F_TEMPORAL): the other ring loads stream
#define LD_W(j, p) ((j) == 0 ? ld_keep<VEC, 2>(p) : ld<VEC>(p))
  int64_t t = tail_block ? ntile : (int64_t)blockIdx.x;
  if (t < ntile) {
    const int64_t e = TILE_ELEM(t);
#pragma unroll
    for (int q = 0; q < T; q++) {
      finv[q] = ld_keep<VEC, 2>(f + e + q * (kBlock * VEC));
      if (COMPACT) w0v[q] = ld_keep<VEC, 2>(w0src + e + q * (kBlock * VEC));
    }
#pragma unroll
    for (int j = 0; j < W; j++)
#pragma unroll
      for (int q = 0; q < T; q++) {
        if (!COMPACT) rw[j][q] = LD_W(j, wk[j] + DEAD_OFF(j < ncomb, e) + q * (kBlock * VEC));
        rv[j][q] = ld<VEC>(vk[j] + DEAD_OFF(j < ncomb, e) + q * (kBlock * VEC));
      }
  }
  if (first_pass) list_word_publish(ctl, ncomb_all, flags & kPbNoStoreW);      // (behind the first ring of loads: nothing waits for it)
  // ticket counter of this block's group; ticket k of group g is tile (k + 2G/ng)*ng + g
  const unsigned grp = tickets ? blockIdx.x % (unsigned)ng : 0u;
  unsigned *const my_ticket = tickets ? tickets + grp * kTicketStride : nullptr;
  const unsigned ticket_base = tickets ? 2u * (unsigned)G / (unsigned)ng : 0u;
  int64_t tnext = t + G;
  unsigned par = 0;
  while (t < ntile) {
    const int64_t e = TILE_ELEM(t);
    const bool more = tnext < ntile;
    unsigned claimed = kNoTicket;
    if (tickets && more && threadIdx.x == 0) claimed = ticket_request(my_ticket, ticket_base, (unsigned)ng, grp);
    const int64_t tn = more ? tnext : t;                 // the last iteration prefetches its own tile again
    const int64_t en = TILE_ELEM(tn);
    V fin[T], w0[T], x[T];
#pragma unroll
    for (int q = 0; q < T; q++) {
      fin[q] = finv[q];
      w0[q] = COMPACT ? w0v[q] : fin[q];
      x[q] = fin[q];
      if (store_w) st(wnew + e + q * (kBlock * VEC), fin[q]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < T; q++) {
      finv[q] = ld_keep<VEC, 2>(f + en + q * (kBlock * VEC));
      if (COMPACT) w0v[q] = ld_keep<VEC, 2>(w0src + en + q * (kBlock * VEC));
    }
#pragma unroll
    for (int j = 0; j < MAXK; j++) {
      V wj[T], vj[T];
#pragma unroll
      for (int q = 0; q < T; q++) {
        wj[q] = COMPACT ? w0[q] : rw[COMPACT ? 0 : j % W][q];
        vj[q] = rv[j % W][q];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < T; q++) {
        if (j + W < MAXK) {
          if (!COMPACT) rw[COMPACT ? 0 : j % W][q] = LD_W(j + W, wk[j + W] + DEAD_OFF(j + W < ncomb, e) + q * (kBlock * VEC));
          rv[j % W][q] = ld<VEC>(vk[j + W] + DEAD_OFF(j + W < ncomb, e) + q * (kBlock * VEC));
        } else {
          if (!COMPACT) rw[COMPACT ? 0 : j % W][q] = LD_W(j + W - MAXK, wk[j + W - MAXK] + DEAD_OFF(j + W - MAXK < ncomb, en) + q * (kBlock * VEC));
          rv[j % W][q] = ld<VEC>(vk[j + W - MAXK] + DEAD_OFF(j + W - MAXK < ncomb, en) + q * (kBlock * VEC));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (j == 0 && norm0) {
#pragma unroll
        for (int q = 0; q < T; q++) {
#pragma unroll
          for (int c = 0; c < VEC; c++) {
            const double d = ex(wj[q], c) - ex(fin[q], c);
            const double wn = RCP ? rs * d : d / s;
            const double vn = RCP ? rs * ex(vj[q], c) : ex(vj[q], c) / s;
            setc(wj[q], c, wn);
            setc(vj[q], c, COMPACT ? vn - wn : vn);
          }
          st(wk[0] + e + q * (kBlock * VEC), wj[q]);
          st(vk[0] + e + q * (kBlock * VEC), vj[q]);
        }
      }
      if (j < ncomb) {
#pragma unroll
        for (int q = 0; q < T; q++)
#pragma unroll
          for (int c = 0; c < VEC; c++) {
            if (COMPACT) setc(x[q], c, ex(x[q], c) + ck[j] * ex(vj[q], c));
            else setc(x[q], c, comb1<COMB>(ex(x[q], c), ck[j], ex(wj[q], c), ex(vj[q], c)));
          }
      }
    }
#pragma unroll
    for (int q = 0; q < T; q++) {
      if (store_v) st(vnew + e + q * (kBlock * VEC), x[q]);
      if (store_f) st(f + e + q * (kBlock * VEC), x[q]);
    }
#ifdef NKA_SOLVE_STAMPS
    if (t == (int64_t)blockIdx.x) NKA_STAMP0(ctl, 15);       // block 0: its first tile is done
#endif
    const int64_t t2 = tickets ? ticket_publish(s_next, par, claimed, ntile) : tnext + G;
    t = tnext;
    tnext = t2;
  }
  if (tickets && !tail_block) ticket_finish(tickets, ng, G);
  if (tail_block) {  // ragged tail, scalar
    for (int64_t i = ntile * TILE + threadIdx.x; i < vs.n; i += kBlock) {
      const double fin = f[i];
      double x = fin;
#pragma unroll
      for (int j = 0; j < MAXK; j++) {
        if (j < ncomb) {
          double v = vk[j][i];
          double w = (!COMPACT || (j == 0 && norm0)) ? wk[j][i] : 0.0;
          if (j == 0 && norm0) {
            const double d = w - fin;
            w = RCP ? rs * d : d / s;
            v = RCP ? rs * v : v / s;
            if (COMPACT) v = v - w;
            wk[0][i] = w;
            vk[0][i] = v;
          }
          x = COMPACT ? x + ck[j] * v : comb1<COMB>(x, ck[j], w, v);
        }
      }
      if (store_w) wnew[i] = fin;
      if (store_v) vnew[i] = x;
      if (store_f) f[i] = x;
    }
  }
#undef DEAD_OFF
#undef TILE_ELEM
#undef LD_W
}

// ---- scalar kernels: list surgery + Cholesky + substitutions on one wavefront ----
// Working copy of the control arrays in LDS (indices as in the Fortran: slots
// 1..M1, 0 = end of list).
struct Lst {
  int32_t *next, *prev;
  double *h;   // h[i*(M1+1)+j] == reference h(i,j)
  double *c;
  int first, last, free_, subspace, pending, m1, mvec;
  double vtol;
  __device__ double &H(int i, int j) { return h[i * (m1 + 1) + j]; }
};

// F08:439-457
__device__ inline void lst_relax(Lst &L) {
  if (!L.pending) return;
  const int dropped = L.first;
  L.first = L.next[dropped];
  if (L.first == 0) L.last = 0; else L.prev[L.first] = 0;
  L.next[dropped] = L.free_;
  L.free_ = dropped;
  L.pending = 0;
}

// F08:422-436
__device__ inline void lst_restart(Lst &L) {
  L.subspace = 0;
  L.pending = 0;
  L.first = 0;
  L.last = 0;
  L.free_ = 1;
  for (int k = 1; k < L.m1; k++) L.next[k] = k + 1;
  L.next[L.m1] = 0;
}

// F08:295-351.  Row-by-row Cholesky of the Gram matrix in list order; capacity
// drop of the last entry; dependence drop when the pivot hkk <= vtol^2.  The
// subtraction order of the inner loop (i ascending in list order) is preserved:
// with equal dot products the decisions equal the reference's bit for bit.
__device__ inline void lst_factor(Lst &L) {
  L.H(L.first, L.first) = 1.0;
  int k = L.next[L.first];
  int nvec = 1;
  while (k != 0) {
    nvec++;
    if (nvec > L.mvec) {
      L.next[L.last] = L.free_;
      L.free_ = k;
      L.last = L.prev[k];
      L.next[L.last] = 0;
      break;
    }
    double hkk = 1.0;
    for (int j = L.first; j != k; j = L.next[j]) {
      double hkj = L.H(j, k);
      for (int i = L.first; i != j; i = L.next[i]) hkj = hkj - L.H(k, i) * L.H(j, i);
      hkj = hkj / L.H(j, j);
      hkk = hkk - hkj * hkj;
      L.H(k, j) = hkj;
    }
    if (hkk > L.vtol * L.vtol) {
      L.H(k, k) = sqrt(hkk);
    } else {
      const int p = L.prev[k], nx = L.next[k];
      L.next[p] = nx;
      if (nx == 0) L.last = p; else L.prev[nx] = p;
      L.next[k] = L.free_;
      L.free_ = k;
      k = p;
      nvec--;
    }
    k = L.next[k];
  }
  L.subspace = 1;
  L.pending = 0;
}

// F08:369-392 (c holds the right-hand side on entry)
__device__ inline void lst_solve(Lst &L) {
  for (int j = L.first; j != 0; j = L.next[j]) {
    double cj = L.c[j];
    for (int i = L.first; i != j; i = L.next[i]) cj = cj - L.H(j, i) * L.c[i];
    L.c[j] = cj / L.H(j, j);
  }
  for (int j = L.last; j != 0; j = L.prev[j]) {
    double cj = L.c[j];
    for (int i = L.last; i != j; i = L.prev[i]) cj = cj - L.H(i, j) * L.c[i];
    L.c[j] = cj / L.H(j, j);
  }
}

// F08:406-417
__device__ inline void lst_prepend(Lst &L, int slot) {
  L.prev[slot] = 0;
  L.next[slot] = L.first;
  if (L.first == 0) L.last = slot; else L.prev[L.first] = slot;
  L.first = slot;
  L.pending = 1;
}

constexpr int kSolveThreads = 64;  // ONE wavefront

// dynamic LDS: next[M1+1], prev[M1+1] (int32) then h[(M1+1)^2], c[M1+1] (double).
// in_global != 0 (mvec > 140: the (mvec+2)^2 matrix no longer fits the 160 KiB of LDS): the working
// arrays ARE the control block in global memory -- no copy in, none back; slow (every step of the
// list-ordered loops is a dependent global access), but the reference has no limit on mvec
// (F08:185-200) and neither has this build.
__device__ inline void lst_load(Lst &L, const Ctl &ctl, unsigned char *smem, int in_global = 0) {
  const int m1 = ctl.m1(), nh = (m1 + 1) * (m1 + 1);
  L.m1 = m1;
  L.mvec = ctl.mvec;
  if (in_global) {
    L.h = ctl.h();
    L.c = ctl.c();
    L.next = ctl.next();
    L.prev = ctl.prev();
  } else {
    L.h = reinterpret_cast<double *>(smem);
    L.c = L.h + nh;
    L.next = reinterpret_cast<int32_t *>(L.c + (m1 + 1));
    L.prev = L.next + (m1 + 1);
    for (int i = threadIdx.x; i < nh; i += kSolveThreads) L.h[i] = ctl.h()[i];
    for (int i = threadIdx.x; i < m1 + 1; i += kSolveThreads) {
      L.c[i] = ctl.c()[i];
      L.next[i] = ctl.next()[i];
      L.prev[i] = ctl.prev()[i];
    }
  }
  L.subspace = ctl.ic[IC_SUBSPACE];
  L.pending = ctl.ic[IC_PENDING];
  L.first = ctl.ic[IC_FIRST];
  L.last = ctl.ic[IC_LAST];
  L.free_ = ctl.ic[IC_FREE];
  L.vtol = ctl.dc[DC_VTOL];
  __syncthreads();
}

__host__ __device__ constexpr size_t lst_smem_bytes(int mvec) {
  const int m1 = mvec + 1;
  return (size_t)((m1 + 1) * (m1 + 1) + (m1 + 1)) * sizeof(double) + 2 * (size_t)(m1 + 1) * sizeof(int32_t);
}

// Lane 0 writes the scalars and the plan for the next update; all lanes copy
// the arrays back.
__device__ inline void lst_store(Lst &L, const Ctl &ctl, int in_global = 0) {
  __syncthreads();
  const int m1 = L.m1, nh = (m1 + 1) * (m1 + 1);
  if (!in_global) {
    for (int i = threadIdx.x; i < nh; i += kSolveThreads) ctl.h()[i] = L.h[i];
    for (int i = threadIdx.x; i < m1 + 1; i += kSolveThreads) {
      ctl.c()[i] = L.c[i];
      ctl.next()[i] = L.next[i];
      ctl.prev()[i] = L.prev[i];
    }
  }
  if (threadIdx.x == 0) {
    ctl.ic[IC_SUBSPACE] = L.subspace;
    ctl.ic[IC_PENDING] = L.pending;
    ctl.ic[IC_FIRST] = L.first;
    ctl.ic[IC_LAST] = L.last;
    ctl.ic[IC_FREE] = L.free_;
    // plan for the next update's PA
    ctl.ic[IC_PLAN_PENDING] = L.pending;
    ctl.ic[IC_PLAN_FIRST] = L.first;
    int n = 0;
    int32_t *ps = ctl.plan_slots();
    const long long *wt = ctl.wtab();
    for (int k = L.pending ? L.next[L.first] : L.first; k != 0; k = L.next[k]) {
      ctl.plan_w()[n] = wt[k];       // the streaming passes get addresses, not slots (Ctl::pc)
      ps[n++] = k;
    }
    ctl.ic[IC_PLAN_NOLDER] = n;
    ctl.pc[PC_FIRST_W] = wt[L.first];      // (entry 0 of the table is a valid dummy: first == 0 without a list)
  }
}

// The slot that receives the new pair gets its buffers here.  An out-of-place update (swap_w / swap_v != kNoBuffer,
// nka_hip_accel_update_swap) EXCHANGES them: the caller's buffer, which holds f_in, becomes the slot's w -- no copy --
// and a spare buffer of the library becomes its v; what the slot held before is reported in PC_OLD_W / PC_OLD_V.
__device__ inline void assign_new_buffers(const Ctl &ctl, int slot, long long swap_w, long long swap_v) {
  long long *wt = ctl.wtab(), *vt = ctl.vtab();
  if (swap_w != kNoBuffer) { ctl.pc[PC_OLD_W] = wt[slot]; wt[slot] = swap_w; }      // (other updates leave PC_OLD_* alone:
  if (swap_v != kNoBuffer) { ctl.pc[PC_OLD_V] = vt[slot]; vt[slot] = swap_v; }      //  the host may collect them later)
  ctl.pc[PC_NEW_W] = wt[slot];
  ctl.pc[PC_NEW_V] = vt[slot];
}

static __global__ __launch_bounds__(kSolveThreads) __attribute__((unused)) void k_restart(Ctl ctl, int in_global) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Lst L;
  lst_load(L, ctl, smem, in_global);
  if (threadIdx.x == 0) lst_restart(L);
  lst_store(L, ctl, in_global);
}

static __global__ __launch_bounds__(kSolveThreads) __attribute__((unused)) void k_relax(Ctl ctl, int in_global) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Lst L;
  lst_load(L, ctl, smem, in_global);
  if (threadIdx.x == 0) lst_relax(L);
  lst_store(L, ctl, in_global);
}

// `mode` of the scalar step.  kSolveRcp: the F08-vector flavour, whose
// normalisation is a multiplication by 1/s.  kSolvePrenorm: red[1] and the Gram
// row red[2..] were already evaluated on the NORMALISED w1' (the host
// dot-product path, nka_hip_set_host_dot) and are taken as they are.
enum { kSolveRcp = 1, kSolvePrenorm = 2 };
__device__ __forceinline__ double solve_nrm(double x, double s, double rs, int mode) {
  return (mode & kSolvePrenorm) ? x : ((mode & kSolveRcp) ? rs * x : x / s);
}

// The scalar part of accel_update between PA and PB, reference loops verbatim on one lane.
// `phase`: 0 = the whole step.  The user-dot-product path (nka_hip_set_host_dot) runs it in two halves so that
// the host can ask the user's dp for the projection row AFTER the drop decisions, as the reference does (F08:371
// comes behind F08:295-347): 1 = norm, s == 0 -> relax, Gram row, factorisation with drops; 2 = new slot, the
// substitutions on the right-hand side the host has put into c[] BY SLOT, combine plan, prepend.
static __global__ __launch_bounds__(kSolveThreads) __attribute__((unused)) void k_solve(Ctl ctl, int mode, int in_global,
                                                                                       int phase, long long swap_w,
                                                                                       long long swap_v, P2P x) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (x.base != nullptr) p2p_gather_block(x, ctl.red(), ctl.red_count());      // the sums of all ranks, in rank order
  Lst L;
  lst_load(L, ctl, smem, in_global);
  if (threadIdx.x == 0) {
    const double *red = ctl.red();
    const int32_t *ps = ctl.plan_slots();
    const int nolder = ctl.ic[IC_PLAN_NOLDER];
    const int entry_first = L.first;
    bool normed = (phase == 2) ? ctl.ic[IC_NORMED] != 0 : false;
    double s = (phase == 2) ? ctl.dc[DC_S] : 0.0;
    if (phase != 2) {
      if (L.pending) {
        s = sqrt(red[0]);                       // F08:267
        ctl.dc[DC_S] = s;
        if (s == 0.0) {                         // F08:275
          lst_relax(L);
          ctl.ic[IC_NRELAX] += 1;
        }
      }
      const double rs = 1.0 / s;
      if (L.pending) {
        normed = true;
        // Gram row of w1' = d/s from the raw sums <d,w_k> of PA (F08:286-290)
        for (int p = 0; p < nolder; p++) L.H(L.first, ps[p]) = solve_nrm(red[2 + p], s, rs, mode);
        lst_factor(L);
      }
      ctl.ic[IC_NORMED] = normed ? 1 : 0;
    }
    if (phase != 1) {
      const double rs = 1.0 / s;
      const int slot = L.free_;
      L.free_ = L.next[slot];
      int ncomb = 0;
      if (L.subspace) {
        if (phase == 0) {
          if (normed) L.c[entry_first] = solve_nrm(red[1], s, rs, mode);   // <f,w1'> = <f,d>/s
          for (int p = 0; p < nolder; p++) L.c[ps[p]] = red[2 + ctl.mvec + p];
        }
        lst_solve(L);
        for (int k = L.first; k != 0; k = L.next[k]) {
          ctl.comb_slots()[ncomb] = k;
          ctl.comb_w()[ncomb] = ctl.wtab()[k];
          ctl.comb_v()[ncomb] = ctl.vtab()[k];
          ctl.comb_c()[ncomb] = L.c[k];
          ncomb++;
        }
      }
      ctl.ic[IC_NCOMB] = ncomb;
      ctl.ic[IC_NEW] = slot;
      assign_new_buffers(ctl, slot, swap_w, swap_v);     // (before lst_store resolves the next plan through the tables)
      lst_prepend(L, slot);
    }
  }
  lst_store(L, ctl, in_global);
}

// ---- the same scalar step with the O(m^3) arithmetic spread over ONE wavefront ----
// k_solve above runs the reference loops verbatim on one lane (~130 us at m=20: every step waits
// on an LDS-resident linked list).  k_solve_rows linearises the list, gathers the Gram entries
// into a dense position-indexed matrix and factorises it RIGHT-LOOKING: when column i is reached
// its pivot is final (decide keep / drop exactly as F08:326), the column is scaled by one division
// per row (lanes = rows) and every trailing entry gets
//      a(p,q) <- a(p,q) - a(p,i)*a(q,i)
// Each entry thus receives the same subtractions in the same (ascending i) order as the
// reference's inner loop F08:316-319, then the same division F08:320 -- bit-identical results,
// ~m sequential steps instead of ~m^3/3.  The right-hand side rides along as one more row (forward
// substitution, F08:369-379); the back-substitution (F08:382-392) is column-oriented in the same
// way.  List surgery (drops, free-list pushes in list order, new slot, prepend) is O(m) and done
// redundantly by all lanes on private copies of the scalars.  Requires mvec+1 <= kSolveWaveMax;
// larger subspaces use k_solve.
//
// What sets the run time of a LONE wavefront (tools/solve_phases.py, s_memtime stamps;
// profiles/r02 and r03/solve_phases*.txt for the forms tried) is, in this order: serial round trips
// (LDS ~130 cycles, global ~500-2000), taken branches (~20 cycles each), and the issue rate of
// dependent fp64 instructions (~6 cycles each; sqrt and divide are ~15-instruction chains).  So:
//   * ONE global round trip at entry: every load of the state, sums and plan issued before the first
//     is waited for (as plain loops the compiler emits load -> wait -> ds_write per iteration);
//   * the list order comes from the plan (no walk); lane p holds row p of the matrix IN REGISTERS,
//     pivots and column entries travel by v_readlane with a uniform source lane (a scalar
//     broadcast): no LDS access and no barrier inside the factorisation;
//   * constant trip counts and guards instead of break / continue, no per-entry branches: both
//     loops unroll completely and the inner one is straight-line code;
//   * `alive` as a 64-bit mask: drops are visited by count-trailing-zeros, compaction by
//     population count, in parallel;
//   * the factor goes to LDS once (the back-substitution reads COLUMNS of it, requested as one
//     batch) and to the stored matrix by slot with branch-free predicated addresses; the plan of
//     the next update is written in parallel.
// History: k_solve_wave (LDS-resident, round 2: ~66 k cycles at m = 20), k_solve_wave2 (pairs dealt
// to lanes, LDS column exchange with two barriers per column: 42.9 k), a ds_bpermute variant
// (50 k, not kept), this one: 26.4 k.  The rare path without a new pair (after relax / s == 0)
// keeps the first version's gather-and-substitute code.
constexpr int kSolveWaveMax = 63;     // (round 5: 48 -> 63, one more instantiation: the lanes were there; mvec <= 62)

__host__ __device__ inline size_t solve_wave_smem_bytes(int mvec) {
  const int nl = mvec + 1;
  size_t b = (lst_smem_bytes(mvec) + 15) / 16 * 16;
  b += (size_t)((nl + 1) * (nl + 1) + 3 * nl + (2 + 2 * mvec)) * sizeof(double);
  b += (size_t)(3 * nl + 1) * sizeof(int32_t);                // (+1: keeps the pointer tables behind them 8-byte aligned)
  b = (b + 7) / 8 * 8;
  b += (size_t)(2 * (nl + 1)) * sizeof(long long);            // slot -> buffer tables (Ctl::wtab / vtab)
  return b;
}

// NLMAX >= list length is a template parameter: both loops of the factorisation are fully unrolled so
// that the row a[] stays in registers (62 VGPRs at NLMAX = 21, 116 at 48; no scratch).  Column i:
// l_p = a_p[i] / L_ii on every lane, then for q = i+1 .. (uniform loop) a_p[q] -= l_p * l_q with
// l_q = readlane(l, q) -- lanes p <= q update entries nobody reads.
template <int NLMAX>
__global__ __launch_bounds__(kSolveThreads) void k_solve_rows(Ctl ctl, int mode, long long swap_w, long long swap_v,
                                                              long long id_stride, long long id_vbase, P2P x) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  // peer-to-peer exchange: wait for the rows of all ranks and add them in rank order (P2P above); red[] then holds the
  // global sums like after an all-reduce (lane e reads back below what it has just written itself)
  if (x.base != nullptr) p2p_gather_block(x, ctl.red(), ctl.red_count());
  const int m1 = ctl.m1(), NL = m1, LDA = NL + 1, M = ctl.mvec, nh = (m1 + 1) * (m1 + 1);
  Lst L;
  L.m1 = m1;
  L.mvec = M;
  L.h = reinterpret_cast<double *>(smem);
  L.c = L.h + nh;
  L.next = reinterpret_cast<int32_t *>(L.c + (m1 + 1));
  L.prev = L.next + (m1 + 1);
  double *A = reinterpret_cast<double *>(smem + (lst_smem_bytes(ctl.mvec) + 15) / 16 * 16);
  double *bb = A + (NL + 1) * LDA + 2 * NL;   // (the 2*NL doubles in between are unused)
  double *redL = bb + NL;
  int32_t *ord = reinterpret_cast<int32_t *>(redL + (2 + 2 * M));
  int32_t *psL = ord + 2 * NL;
  long long *wtL = reinterpret_cast<long long *>((reinterpret_cast<uintptr_t>(psL + NL + 1) + 7) / 8 * 8);   // [m1 + 1]
  long long *vtL = wtL + (m1 + 1);                                                                            // [m1 + 1]
  NKA_STAMP(ctl, 0);
  // ---- one global round trip: EVERY load is issued before the first is waited for (written as plain
  //      loops the compiler emits load -> s_waitcnt vmcnt(0) -> ds_write per iteration: eleven serial
  //      round trips of ~500 cycles at mvec = 20).  m1 + 1 <= kSolveWaveMax + 1 <= 64 and
  //      2 + 2 mvec <= 128, so the lists, sums and plan take one or two loads a lane; the Gram
  //      matrix kHB a lane (mvec <= 21), the rest of a larger one in the old loop.
  const int nolder = ctl.ic[IC_PLAN_NOLDER];
  {
    constexpr int kHB = 8;
    static_assert(kSolveWaveMax + 1 <= kSolveThreads, "lists: one load a lane");
    const double *gh = ctl.h(), *gred = ctl.red();
    double hreg[kHB], rreg[2], creg = 0.0;
    long long wtreg = 0, vtreg = 0;
    int32_t nreg = 0, preg = 0, psreg = 0;
#pragma unroll
    for (int k = 0; k < kHB; k++) hreg[k] = (lane + kSolveThreads * k < nh) ? gh[lane + kSolveThreads * k] : 0.0;
#pragma unroll
    for (int k = 0; k < 2; k++) rreg[k] = (lane + kSolveThreads * k < 2 + 2 * M) ? gred[lane + kSolveThreads * k] : 0.0;
    if (lane < m1 + 1) {
      creg = ctl.c()[lane];
      nreg = ctl.next()[lane];
      preg = ctl.prev()[lane];
      // id_stride != 0: the slot -> buffer tables are still the ones of creation (no out-of-place update yet): slot k at
      // (k-1)*stride of the two slot-major allocations -- computed, not loaded (two more cache lines in the entry round
      // trip cost the one-wavefront step ~1 us at small n)
      if (id_stride != 0) {
        wtreg = (long long)(lane - 1) * id_stride;
        vtreg = id_vbase + (long long)(lane - 1) * id_stride;
      } else {
        wtreg = ctl.wtab()[lane];
        vtreg = ctl.vtab()[lane];
      }
    }
    if (lane < M) psreg = ctl.plan_slots()[lane];        // (bounded by mvec, not by the count still on its way)
#pragma unroll
    for (int k = 0; k < kHB; k++)
      if (lane + kSolveThreads * k < nh) L.h[lane + kSolveThreads * k] = hreg[k];
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (lane + kSolveThreads * k < 2 + 2 * M) redL[lane + kSolveThreads * k] = rreg[k];
    if (lane < m1 + 1) {
      L.c[lane] = creg;
      L.next[lane] = nreg;
      L.prev[lane] = preg;
      wtL[lane] = wtreg;
      vtL[lane] = vtreg;
    }
    if (lane < nolder) psL[lane] = psreg;
    for (int i = lane + kSolveThreads * kHB; i < nh; i += kSolveThreads) L.h[i] = gh[i];
  }
  L.subspace = ctl.ic[IC_SUBSPACE];
  L.pending = ctl.ic[IC_PENDING];
  L.first = ctl.ic[IC_FIRST];
  L.last = ctl.ic[IC_LAST];
  L.free_ = ctl.ic[IC_FREE];
  L.vtol = ctl.dc[DC_VTOL];
  const int entry_pending = L.pending;
  __syncthreads();
  NKA_STAMP(ctl, 1);
  const double vtol2 = L.vtol * L.vtol;

  // ---- phase 0: norm, s == 0 -> relax, Gram row of w1' = d/s, right-hand side
  const int entry_first = L.first;
  int normed = 0;
  double s = 0.0;
  if (L.pending) {
    s = sqrt(redL[0]);                        // F08:267
    if (s == 0.0) lst_relax(L);               // F08:275
  }
  if (L.pending) normed = 1;
  {
    const double rs = 1.0 / s;
    for (int p = lane; p < nolder; p += kSolveThreads) {
      if (normed) L.H(L.first, psL[p]) = solve_nrm(redL[2 + p], s, rs, mode);         // F08:286-290
      L.c[psL[p]] = redL[2 + M + p];                                                 // F08:371
    }
    if (normed && lane == 0) L.c[entry_first] = solve_nrm(redL[1], s, rs, mode);     // <f,w1'> = <f,d>/s
  }
  __syncthreads();
  // list position -> slot WITHOUT walking the list: with a new pair this call the list is `first` followed
  // by the plan (the older entries in list order, which PA's sums follow too; every operation that changes
  // the list rewrites the plan).  The path without a new pair walks the list itself, below.
  const int nl = normed ? 1 + nolder : 0;
  const int myord = normed ? (lane == 0 ? entry_first : (lane <= nolder ? psL[lane - 1] : 0)) : 0;
  NKA_STAMP(ctl, 2);
  const uint64_t listmask = (nl >= 64) ? ~0ull : ((1ull << nl) - 1);
  uint64_t alive = listmask;
  int capdrop = -1;
  bool forward_done = false;
  double ddr = 1.0;    // lane p: running pivot 1 - sum l^2 of list position p
  double Ldr = 1.0;    // lane p: accepted pivot sqrt(hkk)
  double yr = 0.0;     // lane p: right-hand side / solution of list position p
  int nk = 0;

  if (normed) {
    // ---- phase 1: right-looking Cholesky with drops (F08:295-347), rows 0..nl-1,
    //      plus the right-hand side as row nl (lane p scales row p of each column).
    // Row p of the matrix is gathered into the registers of lane p (the loads are issued as a batch).
    double a[NLMAX];
#pragma unroll
    for (int q = 0; q < NLMAX; q++) {
      const int oq = __builtin_amdgcn_readlane(myord, q);      // slot of list position q (0 beyond the list)
      a[q] = (lane < nl) ? L.H(oq, myord)                      // raw <w_q,w_p>, q newer (used for q < p only)
                         : L.c[oq];                            // row nl: rhs <f,w_q>
    }
    NKA_STAMP(ctl, 3);
    int kept = 0;
    bool open_ = true;                           // false once the capacity drop has ended the list (F08:309 exit)
    // (constant trip counts, guards instead of break / continue: both loops must unroll completely)
#pragma unroll
    for (int i = 0; i < NLMAX; i++) {
      bool keep = false;
      double Lii = 1.0;
      if (open_ && i < nl) {
        if (i == 0) {
          keep = true;                           // F08:295 h(first,first) = 1
        } else if (kept + 1 > L.mvec) {
          capdrop = i;                           // F08:301-308 capacity: i is the last entry
          open_ = false;
        } else {
          const double hkk = readlane_f64(ddr, i);
          keep = hkk > vtol2;                    // F08:326
          if (keep) Lii = sqrt(hkk);
        }
        if (!keep) alive &= ~(1ull << i);
      }
      if (keep) {
        kept++;
        if (lane == i) Ldr = Lii;
        const double l = a[i] / Lii;             // F08:320 (row nl: F08:377); rows <= i: unused
        a[i] = l;
        if (lane > i && lane < nl) ddr = ddr - l * l;            // F08:321
#pragma unroll
        for (int q = i + 1; q < NLMAX; q++)                      // (no guard q < nl: straight-line code; columns
          a[q] = a[q] - l * readlane_f64(l, q);                  //  beyond the list hold values nobody reads)
                                                                 // trailing entry (p,q), p > q: F08:317 (row nl: F08:374)
      }
    }
    NKA_STAMP(ctl, 4);
    // ---- phase 2: the factor back by slot, and into LDS for the back-substitution (which reads
    //      COLUMNS of it); replay the drops in list order
    // (branch-free: a lone wavefront pays ~20 cycles for every taken branch.  Rows are written whole --
    //  the part right of the diagonal and the columns beyond the list, folded onto column nl, are
    //  read by nobody -- and an entry that must NOT reach the stored factor goes to a dump word.)
    {
      double *const dump = A + NL * LDA + NL;
      const uint64_t rowbits = (lane < nl && ((alive >> lane) & 1)) ? (alive & ((1ull << lane) - 1)) : 0ull;
      if (lane <= nl) {
#pragma unroll
        for (int q = 0; q < NLMAX; q++) A[lane * LDA + (q < nl ? q : nl)] = a[q];
      }
#pragma unroll
      for (int q = 0; q < NLMAX; q++) {
        const int oq = __builtin_amdgcn_readlane(myord, q);
        double *const dst = ((rowbits >> q) & 1) ? &L.H(myord, oq) : dump;
        *dst = a[q];
      }
    }
    __syncthreads();
    if (lane < nl && ((alive >> lane) & 1)) L.H(myord, myord) = Ldr;
    if (lane < nl) yr = A[nl * LDA + lane];    // forward-substituted right-hand side of position p
    forward_done = true;
    for (uint64_t dm = ~alive & listmask & ~1ull; dm != 0; dm &= dm - 1) {
      const int p = __builtin_ctzll(dm);
      const int k = __builtin_amdgcn_readlane(myord, p);
      if (p == capdrop) {                      // F08:303-307
        L.next[L.last] = L.free_;
        L.free_ = k;
        L.last = L.prev[k];
        L.next[L.last] = 0;
      } else {                                 // F08:331-340
        const int pv = L.prev[k], nx = L.next[k];
        L.next[pv] = nx;
        if (nx == 0) L.last = pv; else L.prev[nx] = pv;
        L.next[k] = L.free_;
        L.free_ = k;
      }
    }
    L.subspace = 1;
    L.pending = 0;
    __syncthreads();
  }

  // ---- phase 3: new slot, then the substitutions on the current list
  NKA_STAMP(ctl, 5);
  const int slot = L.free_;                    // F08:357-358
  L.free_ = L.next[slot];
  // buffers of the new pair (every lane computes the same values; lane 0 writes them): an out-of-place update exchanges
  // them (assign_new_buffers).  The new slot is never one of the entries combined below (it comes off the free list).
  const long long new_w = swap_w != kNoBuffer ? swap_w : wtL[slot], new_v = swap_v != kNoBuffer ? swap_v : vtL[slot];
  if (lane == 0) {
    if (swap_w != kNoBuffer) ctl.pc[PC_OLD_W] = wtL[slot];  // (other updates leave PC_OLD_* alone: the host may collect them later)
    if (swap_v != kNoBuffer) ctl.pc[PC_OLD_V] = vtL[slot];
    ctl.pc[PC_NEW_W] = new_w;
    ctl.pc[PC_NEW_V] = new_v;
    ctl.pc[PC_FIRST_W] = new_w;                // the new pair is the pending pair of the next update
    if (swap_w != kNoBuffer) ctl.wtab()[slot] = swap_w;
    if (swap_v != kNoBuffer) ctl.vtab()[slot] = swap_v;
  }
  if (forward_done) {
    // back-substitution F08:382-392 in position space: the factor lies in A, its
    // diagonal in Ldr, the forward-substituted right-hand side in yr
    NKA_STAMP(ctl, 6);
    // column `lane` of the factor, requested as ONE batch of LDS reads (rows / lanes beyond the list fold
    // onto row / column nl: valid addresses, values nobody uses)
    double t[NLMAX];
#pragma unroll
    for (int i = 0; i < NLMAX; i++) t[i] = A[(i < nl ? i : nl) * LDA + (lane < nl ? lane : nl)];
#pragma unroll
    for (int i = NLMAX - 1; i >= 0; i--) {
      if (i < nl && ((alive >> i) & 1)) {
        const double ci = readlane_f64(yr, i) / readlane_f64(Ldr, i);
        if (lane == i) yr = ci;
        if (lane < i && ((alive >> lane) & 1)) yr = yr - t[i] * ci;
      }
    }
    NKA_STAMP(ctl, 7);
    nk = __builtin_popcountll(alive & listmask);
    if (lane < nl && ((alive >> lane) & 1)) {
      const int r = __builtin_popcountll(alive & ((1ull << lane) - 1));   // position among the kept entries
      ctl.comb_slots()[r] = myord;
      ctl.comb_c()[r] = yr;
      ctl.plan_slots()[r] = myord;             // the next update's older entries: this list, in order
      const long long wb = wtL[myord];
      ctl.comb_w()[r] = wb;                    // ... and their addresses for the streaming passes (Ctl::pc)
      ctl.comb_v()[r] = vtL[myord];
      ctl.plan_w()[r] = wb;
      L.c[myord] = yr;
    }
  } else if (L.subspace) {
    // no new pair this call (after relax / s == 0): substitute on the stored factor
    for (int k = L.first; k != 0; k = L.next[k]) ord[nk++] = k;
    __syncthreads();
    for (int p = lane; p < nk; p += kSolveThreads) bb[p] = L.c[ord[p]];
    for (int idx = lane; idx < nk * nk; idx += kSolveThreads) {
      const int p = idx / nk, q = idx - p * nk;
      if (p >= q) A[p * LDA + q] = L.H(ord[p], ord[q]);
    }
    __syncthreads();
    for (int i = 0; i < nk; i++) {             // forward, F08:369-379
      const double ci = bb[i] / A[i * LDA + i];
      __syncthreads();
      if (lane == 0) bb[i] = ci;
      for (int j = i + 1 + lane; j < nk; j += kSolveThreads) bb[j] = bb[j] - A[j * LDA + i] * ci;
      __syncthreads();
    }
    for (int i = nk - 1; i >= 0; i--) {        // backward, F08:382-392
      const double ci = bb[i] / A[i * LDA + i];
      __syncthreads();
      if (lane == 0) bb[i] = ci;
      for (int j = lane; j < i; j += kSolveThreads) bb[j] = bb[j] - A[i * LDA + j] * ci;
      __syncthreads();
    }
    for (int p = lane; p < nk; p += kSolveThreads) {
      ctl.comb_slots()[p] = ord[p];
      ctl.comb_c()[p] = bb[p];
      ctl.plan_slots()[p] = ord[p];
      ctl.comb_w()[p] = wtL[ord[p]];
      ctl.comb_v()[p] = vtL[ord[p]];
      ctl.plan_w()[p] = wtL[ord[p]];
      L.c[ord[p]] = bb[p];
    }
  }
  __syncthreads();
  lst_prepend(L, slot);                        // F08:406-417 (every lane, same values)
  NKA_STAMP(ctl, 8);
  // ---- state back to global memory (the plan was written above: without a subspace
  //      the list was empty before the prepend, so the next update has no older entry)
  __syncthreads();
  for (int i = lane; i < nh; i += kSolveThreads) ctl.h()[i] = L.h[i];
  for (int i = lane; i < m1 + 1; i += kSolveThreads) {
    ctl.c()[i] = L.c[i];
    ctl.next()[i] = L.next[i];
    ctl.prev()[i] = L.prev[i];
  }
  if (lane == 0) {
    ctl.dc[DC_S] = s;
    if (entry_first != 0 && !normed && entry_pending) ctl.ic[IC_NRELAX] += 1;
    ctl.ic[IC_NEW] = slot;
    ctl.ic[IC_NCOMB] = nk;
    ctl.ic[IC_NORMED] = normed;
    ctl.ic[IC_SUBSPACE] = L.subspace;
    ctl.ic[IC_PENDING] = L.pending;
    ctl.ic[IC_FIRST] = L.first;
    ctl.ic[IC_LAST] = L.last;
    ctl.ic[IC_FREE] = L.free_;
    ctl.ic[IC_PLAN_PENDING] = L.pending;
    ctl.ic[IC_PLAN_FIRST] = L.first;
    ctl.ic[IC_PLAN_NOLDER] = nk;
  }
  NKA_STAMP(ctl, 9);
}

}  // namespace nka
