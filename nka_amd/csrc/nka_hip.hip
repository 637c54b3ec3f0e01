// nka_hip.hip -- host side of the C ABI declared in include/nka_hip.h.
//
// The accelerator object (reference: `type nka`, src-F08/nka_type.F90:154-181)
// keeps ALL of its state on the device: the 2*(mvec+1) slot vectors, the Gram /
// Cholesky matrix h, the linked lists and the flags.  accel_update enqueues
//   PA k_dots -> k_finalize_dots -> [all-reduce 2+2*mvec] -> k_solve -> PB k_combine
// on one HIP stream and returns; nothing is read back.  The host only tracks what
// it can know without looking: whether a pair is pending, and an upper bound on
// the list length (used to pick the unroll width of PA/PB).
#include "../../include/nka_hip.h"
#include "../../include/nka_hip_ext.h"
#include "../../include/nka_hip_vec.h"
#ifdef NKA_DIAGNOSTIC
#include "../../include/nka_hip_diag.h"
#endif
#include "nka_kernels.hpp"
#include "host_logic.hpp"

#include <hip/hip_runtime.h>
#include "rccl_dl.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

using namespace nka;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (void)hipGetLastError(); /* reported here: do not leave it for a later hipGetLastError() to find */ \
      return fail(e_ == hipErrorOutOfMemory ? NKA_HIP_ENOMEM : NKA_HIP_EHIP,                \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    }                                                                                       \
  } while (0)

}  // namespace

namespace nka_detail {
// shared with vec_ops.hip: one thread-local error string per library
int set_error(int code, const std::string &msg) { return fail(code, msg); }
std::string last_error() { return g_err; }
bool &span_check_failed() {
  thread_local bool flag = false;
  return flag;
}

// A kernel reading past a caller's buffer faults the GPU (and can take the node
// with it), so every device pointer that crosses the ABI is checked on the host
// against the allocation it lies in before any launch: it must be device
// memory and hold at least n doubles from p on.
//   * Memory handed out by this library (nka_hip_vec_alloc: every vector of the
//     device vector types) is looked up in a registry of live allocations: exact,
//     deterministic, no HIP call (the abstract-vector path passes ~66 pointers per update).
//   * Any other pointer is checked with hipMemGetAddressRange on EVERY call (2-3 us,
//     one pointer per accel_update in the array flavours): a buffer the caller freed
//     and re-allocated shorter at the same address is caught at once.
//   * NKA_HIP_CHECK_POINTERS=cached (opt-in) remembers a foreign span that passed for
//     100 ms per thread (dropped on any free through the library and by
//     nka_hip_invalidate_pointer_cache()); NKA_HIP_CHECK_POINTERS=0 switches the check off.
std::atomic<uint64_t> g_span_generation{1};
void invalidate_span_cache() { g_span_generation.fetch_add(1, std::memory_order_relaxed); }

namespace {
std::mutex g_reg_mu;
struct RegEntry { size_t bytes; const void *owner; };
std::map<uintptr_t, RegEntry> g_reg;   // base address -> (bytes, owning workspace), allocations made through this library
}  // namespace
void register_allocation(const void *p, size_t bytes, const void *owner) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_reg_mu);
  g_reg[reinterpret_cast<uintptr_t>(p)] = RegEntry{bytes, owner};
}
void unregister_allocation(const void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_reg_mu);
  g_reg.erase(reinterpret_cast<uintptr_t>(p));
}
// A workspace that goes away takes its entries with it: vectors it handed out can no longer be freed through the
// library, so an entry left behind would vouch for whatever is allocated at that address later.
void unregister_owner(const void *owner) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  for (auto it = g_reg.begin(); it != g_reg.end();) it = (it->second.owner == owner) ? g_reg.erase(it) : std::next(it);
}

static int check_device_span_impl(const void *p, int64_t n, const char *what);
int check_device_span(const void *p, int64_t n, const char *what) {
  const int rc = check_device_span_impl(p, n, what);
  if (rc) span_check_failed() = true;
  return rc;
}
static int check_device_span_impl(const void *p, int64_t n, const char *what) {
  static const int mode = [] {      // 0 off, 1 strict (default), 2 cached
    const char *e = getenv("NKA_HIP_CHECK_POINTERS");
    if (e && e[0] == '0') return 0;
    if (e && std::string(e) == "cached") return 2;
    return 1;
  }();
  if (mode == 0 || n <= 0) return 0;
  if (!p) return fail(NKA_HIP_EINVAL, std::string(what) + ": NULL device pointer");
  const size_t need = (size_t)n * sizeof(double);
  const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
  bool hit_in_registry = false;
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_reg.upper_bound(addr);
    if (it != g_reg.begin()) {
      --it;
      if (addr < it->first + it->second.bytes) {      // inside an allocation of this library
        if (addr + need > it->first + it->second.bytes)
          return fail(NKA_HIP_EINVAL, std::string(what) + ": device buffer shorter than the vector length");
        // NKA_HIP_DEBUG=1: do not take the registry's word for it (memory released behind the library's back --
        // hipFree, hipDeviceReset -- leaves a stale entry): ask the runtime as for a foreign pointer
        static const bool revalidate = [] { const char *e = getenv("NKA_HIP_DEBUG"); return e && atoi(e) != 0; }();
        if (!revalidate) return 0;
        hit_in_registry = true;
      }
    }
  }
  struct Entry { const void *p; size_t avail; uint64_t gen; int64_t t_ns; };
  thread_local Entry cache[256] = {};
  Entry *e = nullptr;
  uint64_t gen = 0;
  int64_t now = 0;
  if (mode == 2) {
    gen = g_span_generation.load(std::memory_order_relaxed);
    now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    e = &cache[(addr >> 8) * 0x9E3779B97F4A7C15ull >> 56];
    if (e->p == p && e->gen == gen && now - e->t_ns < 100000000 && need <= e->avail) return 0;
  }
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, const_cast<void *>(p)) != hipSuccess) {
    (void)hipGetLastError();
    return fail(NKA_HIP_EINVAL, std::string(what) + (hit_in_registry ? ": a vector of this library whose memory has been released behind its back"
                                                                     : ": not a device allocation"));
  }
  const char *lo = static_cast<const char *>(p), *end = static_cast<const char *>(base) + size;
  if (lo < static_cast<const char *>(base) || lo + need > end)
    return fail(NKA_HIP_EINVAL, std::string(what) + ": device buffer shorter than the vector length");
  if (e) *e = Entry{p, (size_t)(end - lo), gen, now};
  return 0;
}
}  // namespace nka_detail

namespace {

int env_int(const char *name, int dflt) {
  const char *s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

using nka_detail::rccl;   // rccl_dl.hpp: RCCL bound at first use, one copy per process

}  // namespace

struct nka_hip_state {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t n = 0;
  int32_t mvec = 0;
  double vtol = 0.01;
  int flavor = NKA_HIP_FLAVOR_F08;
  Vecs vs{};
  Ctl ctl{};
  double *partials = nullptr;
  unsigned *tickets = nullptr; // tile-ticket counters of the rolling-window PB (kTicketWords, zero between launches)
  double *f_stage = nullptr;  // device staging for the host-array entry point
  double *hd_scratch = nullptr;  // device vector for the operands of a user host dot product (set_host_dot)
  // what the host knows without reading the device back
  bool pending = false;
  int list_ub = 0;            // upper bound on the list length
  // ... and what the device tells it without being asked: the list word (Ctl::hw, nka_kernels.hpp), one 64-bit word
  // in pinned host memory that PB of update number u overwrites with (u, list length at its exit).
  unsigned long long *list_word = nullptr;
  int64_t seq = 0;            // updates enqueued so far; the next one is number seq + 1
  int64_t word_valid_after = 0;      // words of updates numbered <= this are stale (restart; a word of 0 = none yet)
  std::vector<int64_t> relaxed_after;  // relax() calls that dropped a pending pair, by the number of the update before them
  bool word_off = false;      // the handle's stream was seen capturing (replays change the list behind the word), or
                              // the diagnostic switch "list_word" = 0
  // out-of-place updates (nka_hip_accel_update_swap): the slot -> buffer tables live on the device (Ctl::pc); the host
  // knows only the two buffers it can hand out next
  bool swapped = false;       // a table entry may differ from base + (slot-1)*stride: host-side accessors read the tables
  double *spare_w = nullptr, *spare_v = nullptr;   // free buffers the host knows of (given away by the next swap update)
  bool swap_pending = false;  // the buffers the last swap update displaced have not been collected yet (record word 3)
  int64_t swap_seq = 0;       // number of that update
  const double *last_acc = nullptr;   // the accelerated f lent by the last out-of-place update (read only for the caller)
  std::vector<void *> extra_allocs;   // the two spare buffers allocated at the first swap update (freed at destroy)
  // Who holds which buffer, as far as the HOST can know it: `taken` = every buffer outside the two slot-major
  // allocations that a table entry may name (callers' buffers handed in, the two extra allocations); `lent` = the free
  // buffers handed to the caller for its NEXT input (they may lie anywhere, the slot-major allocations included).  A
  // buffer may enter an update only if the library does not hold it: it is lent, or it is foreign to both.
  nka_host::BufferBook book;          // .taken / .lent (host_logic.hpp)
  bool captured = false;      // the handle's stream was seen capturing: replays re-issue the scalar step with the ARGUMENTS of
                              // the capture, so from then on it loads the slot -> buffer tables instead of computing them
  bool solve63_raised = false;   // k_solve_rows<63> has its dynamic-LDS limit raised on this handle's device
  bool poisoned = false;      // a HIP call failed AFTER the scalar step of an update was enqueued: the lists on the device
                              // are ahead of the vectors; every later call but destroy returns NKA_HIP_ESTATE
  // launch geometry
  int num_cu = 256;
  int bpc[2] = {0, 0};        // blocks per CU of PA, PB; 0 = automatic (see grid_for)
  char devname[64] = {0};
  int pa_pipe = -1;           // form of PA: 0 = k_dots (every load of a tile in flight; any list length), 201..204 = k_dots_win
                              // (rolling window) with 1..4 blocks per CU, -1 = automatic (see enqueue_pa)
  int pb_pipe = -1;           // form of PB: 0 = k_combine, 201..204 = k_combine_win, -1 = automatic (see enqueue_pb)
  int pb_flags = 0;           // kPbNoStoreW | kPbNoStoreF while an out-of-place update is being enqueued, else 0
  int pb_tile = -1;           // tile width of the rolling-window PB for short lists: -1 automatic (double-width tiles
                              // when the tickets apply), 1 = 512 elements, 2 = 1024 elements
  int pb_tickets = -1;        // tile tickets of the rolling-window PB: -1 automatic, 0 static tile mapping,
                              // 1, 2, 4, 8 = that many ticket counters (see k_combine_win)
  double *chain_pred = nullptr;   // reference-order sums of the longest vectors (k_chain_blocks ...): per sum and block of 1024, the block
  void *chain_summ = nullptr;     // sum / the predicted running sum, and the block's summary; allocated at the first such update
  long long chain_cap = 0;        // (sums x blocks the two arrays hold)
  int chain_many = -1;            // diagnostic switch "chain_many": -1 automatic (from kChainManyMin elements on), 0 never, 1 whenever blocks exist
  int chain_walk = 0;         // diagnostic switch "chain_walk": k_chain_sums walks every block element after element (A/B of chain_block_summary / _apply)
  int pb_reverse = 0;         // diagnostic switch "pb_reverse": the rolling-window PB walks its tiles from the end (kPbReverse)
  int fail_after_solve = 0;   // diagnostic switch "fail_after_solve": the next update fails as if a HIP call behind its scalar step had
                              // (the test of the poisoned-handle path: the failure itself cannot be provoked from outside)
  int prime_pad = -1;         // list lengths 23, 29, 31 (primes: the only ring of their window kernels is the whole width -- up to 311
                              // VGPRs and scratch in PA, every load of a tile in flight in PB) run the next width with ONE dead ring
                              // slot: -1 / 1 on (automatic), 0 off.  In-process A/B at n = 1e7 (profiles/r05/multipass.txt): update
                              // -12.7 % at m = 31, -3.4 % at 29, -0.3 % at 23 (compact); -11 / -4.4 / -2.9 % in the src-F08 rounding
  bool state_in_global = false;  // mvec > 140: h, c and the links no longer fit the LDS of one CU; the one-lane
                                 // scalar kernels then work on the control block in global memory (slow, unlimited)
  bool serial_solve = false;  // NKA_HIP_SERIAL_SOLVE=1: reference loops verbatim on one lane
  int sum_order = NKA_HIP_SUMS_AUTO;   // nka_hip_set_sum_order: reference-order sums (k_dots_ordered) within one tile / always / never
  bool debug = false;         // NKA_HIP_DEBUG=1: check defined() on entry of every update, like the
                              // reference built without -DNDEBUG (F08:257); synchronises
  // distribution hook
  nka_hip_allreduce_fn allreduce = nullptr;
  void *allreduce_ctx = nullptr;
  ncclComm_t comm = nullptr;
  // peer-to-peer exchange (nka_hip_p2p_export / _attach; nka_kernels.hpp: struct P2P)
  P2P p2p{};                          // base == nullptr: none
  void *p2p_mail = nullptr;           // this rank's mailbox (fine-grained device memory, exported through hipIpc)
  std::vector<void *> p2p_opened;     // the peers' mailboxes as mapped here (hipIpcCloseMemHandle at detach)
  void *p2p_dev = nullptr;            // offsets table, exchange counter, status word
  int p2p_ranks = 0;                  // ranks the mailbox was sized for
  bool p2p_fused = false;             // the PA being enqueued sends its sums itself (update_impl)
  int pa_normed = 0;                  // the PA being enqueued sums on the ROUNDED w1' (NKA_HIP_SUMS_BLOCKED_ROUNDED): 0 no, 1 d/s, 3 (1/s)*d
  int shard_rank = -1, shard_n = 0;   // position of this rank's slice in the global vector (nka_hip_set_shard; set by
                                      // nka_hip_comm_init_rank too): only the sharded reference-order sums need it
  bool needs_comm = false;    // a deep copy of an accelerator that reduced through the built-in RCCL communicator: the
                              // communicator belongs to the original, and rank-local sums would be silently wrong
  nka_hip_host_dot_fn host_dot = nullptr;   // user dot product on host copies (compatibility path)
  void *host_dot_ctx = nullptr;
  // instrumentation
  // kTimingEvents events per update, in a ring of timing_cap updates
  int timing_cap = 0;
  int64_t timing_count = 0;   // updates recorded since set_timing
  int timing_stride = 1;      // record the events of every stride-th update only (tuning key "timing_stride"):
  int64_t update_seq = 0;     // four event records widen the kernel boundaries of an update by ~15 us
  bool timing_this = false;   // the update in progress is a recorded one
  std::vector<hipEvent_t> ev;
};

// Entries of the address block (Ctl::pc) are offsets in doubles from vs.w (nka_kernels.hpp).
static inline long long buffer_offset(const nka_hip_state *a, const double *p) {
  return (long long)((reinterpret_cast<intptr_t>(p) - reinterpret_cast<intptr_t>(a->vs.w)) / (intptr_t)sizeof(double));
}
static inline double *buffer_at(const nka_hip_state *a, long long off) {
  return reinterpret_cast<double *>(reinterpret_cast<intptr_t>(a->vs.w) + (intptr_t)off * (intptr_t)sizeof(double));
}

namespace {

// Persistent grid: one block per resident slot (occupancy of THIS instantiation
// x CU count, capped by the tunable blocks-per-CU and by the number of tiles),
// so every block is co-resident and the grid-stride loops stay balanced.
template <typename K>
int occupancy_of(K kernel) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlock, 0) != hipSuccess || nb < 1) nb = 1;
  return nb;
}

// `nloads` = 16-byte loads each thread keeps in flight per tile.  Measured on
// MI355X (n = 1.25e7 and 1e8, m = 20): ONE block per CU is fastest once a block
// has >= 22 loads per thread in flight (88 KiB per CU); more blocks per CU only
// add write/read interleaving in the mixed pass (PB 0.46 -> 0.52 ms at 4 per CU).
// Narrow instantiations get proportionally more blocks to keep ~88 KiB in flight.
int grid_for(const nka_hip_state *a, int which, int vec, int occ, int nloads) {
  const int64_t ntile = a->n / (kBlock * vec);
  const int want = a->bpc[which] > 0 ? a->bpc[which] : std::max(1, (22 + nloads - 1) / nloads);
  int64_t g = (int64_t)a->num_cu * std::min(occ, want);
  g = std::min<int64_t>(g, std::max<int64_t>(ntile, 1));
  g = std::min<int64_t>(g, kMaxGrid);
  return (int)std::max<int64_t>(g, 1);
}

int rccl_allreduce(void *ctx, double *buf, int32_t count, void *stream) {
  auto *a = static_cast<nka_hip_state *>(ctx);
  ncclResult_t r = rccl().AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, a->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return fail(NKA_HIP_ECOMM, std::string("ncclAllReduce: ") + rccl().GetErrorString(r));
  return 0;
}

// The peer-to-peer exchange as a plain all-reduce hook: one small kernel that sends and gathers (k_p2p_allreduce).  An update in
// the fast passes does not come here: its final-sums kernel sends and its scalar step gathers (update_impl).
int p2p_allreduce(void *ctx, double *buf, int32_t count, void *stream) {
  auto *a = static_cast<nka_hip_state *>(ctx);
  if (!a->p2p.base) return fail(NKA_HIP_ECOMM, "peer-to-peer exchange: not attached");
  if (count < 0 || count > a->p2p.cap) return fail(NKA_HIP_EINVAL, "peer-to-peer exchange: more values than a mailbox row holds");
  if (count == 0) return 0;
  hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(128), 0, (hipStream_t)stream, a->p2p, buf, (int)count);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NKA_HIP_ECOMM, std::string("peer-to-peer exchange: ") + hipGetErrorString(e));
  return 0;
}

// ---- kernel dispatch by unroll width ------------------------------------------
template <int MAXL, int VEC>
int launch_dots_1(const nka_hip_state *a, const double *f, int pass, int npass) {
  static const int occ = occupancy_of(k_dots<MAXL, VEC>);
  const int g = grid_for(a, 0, VEC, occ, MAXL + 2);
  hipLaunchKernelGGL((k_dots<MAXL, VEC>), dim3(g), dim3(kBlock), 0, a->stream, a->ctl, a->vs, f, a->partials, pass, a->pa_normed);
  hipLaunchKernelGGL((k_finalize_dots<MAXL>), dim3(2 * MAXL + 2), dim3(kFinThreads), 0, a->stream, a->ctl,
                     a->partials, g, pass, npass * MAXL, pass * MAXL, a->p2p_fused ? a->p2p : P2P{}, a->pa_normed ? 1 : 0);
  return g;
}

// rolling-window PA: ring of W registers, `bpc` blocks per CU
template <int MAXL, int W>
int launch_dots_win_1(const nka_hip_state *a, const double *f, int bpc, int base, int pass, int ncover) {
  static const int occ = occupancy_of(k_dots_win<MAXL, W>);
  const int64_t ntile = a->n / (kBlock * 2);
  int64_t g = (int64_t)a->num_cu * std::min(occ, std::max(1, bpc));
  g = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(g, std::max<int64_t>(ntile, 1)), kMaxGrid));
  hipLaunchKernelGGL((k_dots_win<MAXL, W>), dim3((int)g), dim3(kBlock), 0, a->stream, a->ctl, a->vs, f, a->partials, base, a->pa_normed);
  // (Round 3 measured forming these sums -- and the scalar step -- in the tail of the PA launch, by the block that
  //  finishes last: 2-4 us SLOWER per update than the launches it saves, profiles/r03/ab_small_pa_tail_not_kept.txt.)
  hipLaunchKernelGGL((k_finalize_dots<MAXL>), dim3(2 * MAXL + 2), dim3(kFinThreads), 0, a->stream, a->ctl,
                     a->partials, (int)g, pass, ncover < 0 ? MAXL : ncover, base, a->p2p_fused ? a->p2p : P2P{}, a->pa_normed ? 1 : 0);
  return (int)g;
}

// The ring size: a small divisor of the width (4, 5, 6, 3 or 7), or the width itself when it
// is prime (then every load of a tile is in flight and each is re-issued for the next tile
// as soon as it has been consumed).  Rings of 2, MAXL/4 and MAXL/2 measured slower than 4
// (profiles/r02/ab_inproc_pipelined_passes.txt).
template <int MAXL>
constexpr int win_ring() {
  return MAXL % 4 == 0 ? 4 : MAXL % 5 == 0 ? 5 : MAXL % 6 == 0 ? 6 : MAXL % 3 == 0 ? 3 : MAXL % 7 == 0 ? 7 : MAXL;
}

// The flavours that stream TWO vectors per pair (F08, F08-vector) keep the same number of loads in
// flight with half the ring: 2 pairs where the width is even (in-process A/B at m = 20: -3.5 % at
// n = 1.25e7, equal at 1e8; the compact flavour loses 13 % with a ring of 2 single loads).
template <int MAXK>
constexpr int win_ring_pairs() {
  return MAXK % 2 == 0 ? 2 : MAXK % 3 == 0 ? 3 : MAXK % 5 == 0 ? 5 : MAXK % 7 == 0 ? 7 : MAXK;
}

// Instantiated for EVERY width 1..32 so that no list length needs padding: a padded ring slot is a cache hit that starves
// the window (PA at m = 5 through the width-8 kernel: +20 %; exact against padded widths -7...-14 % at m = 5 / 10,
// profiles/r02/ab_inproc_window_matrix.txt).  A maintainer who prefers a small library builds with
// -DNKA_WINDOW_WIDTH_STEP=4 (`make WIDTH_STEP=4`): widths 4, 8, ... 32 only, every other list length padded to the next one
// -- same bits, nka_hip.o 3.4 -> 1.6 MB and 34 -> 18 s of compile time on this image's 8 cores (measured in round 5, DESIGN.md
// section 4; libnka_hip.so 5.9 -> 4.1 MB: the other 2.9 MB are the abstract-vector kernels of vec_ops.o), at that price.
#ifndef NKA_WINDOW_WIDTH_STEP
#define NKA_WINDOW_WIDTH_STEP 1
#endif
static_assert(NKA_WINDOW_WIDTH_STEP == 1 || NKA_WINDOW_WIDTH_STEP == 4, "NKA_WINDOW_WIDTH_STEP: 1 (every width) or 4");
#if NKA_WINDOW_WIDTH_STEP == 1
#define NKA_WIDTH_CASES CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) \
  CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) \
  CASE(28) CASE(29) CASE(30) CASE(31) CASE(32)
static inline int window_width(int w) { return w; }
#else
#define NKA_WIDTH_CASES CASE(4) CASE(8) CASE(12) CASE(16) CASE(20) CASE(24) CASE(28) CASE(32)
static inline int window_width(int w) { return ((std::max(w, 1) + 3) / 4) * 4; }
#endif
// (base, pass, ncover: the balanced passes of a list longer than kMaxPerPass, enqueue_pa; one launch: 0, 0, its own width)
int launch_dots_win(int width, const nka_hip_state *a, const double *f, int bpc, int base = 0, int pass = 0, int ncover = -1) {
#define CASE(L) \
  case L: return launch_dots_win_1<L, win_ring<L>()>(a, f, bpc, base, pass, ncover);
  switch (window_width(width)) {
    NKA_WIDTH_CASES
  }
#undef CASE
  return 0;
}

// The all-loads-in-flight PA serves what the rolling-window kernel does not: lists longer than 32 (passes of 32) and an
// update whose only stored vector is the pending pair (width 4, all padding).  Any other width reaches it only through
// the diagnostic switch pa_pipe = 0 and is then padded to 32 (same bits; the padding costs time, not correctness).
int launch_dots_w(int maxl, const nka_hip_state *a, const double *f, int pass, int npass) {
  if (maxl <= 4) return launch_dots_1<4, 2>(a, f, pass, npass);
  return launch_dots_1<32, 2>(a, f, pass, npass);
}

template <int MAXK, int VEC, int COMB>
int launch_combine_1(const nka_hip_state *a, double *f, int pass, int last) {
  static const int occ = occupancy_of(k_combine<MAXK, VEC, COMB>);
  const int g = grid_for(a, 1, VEC, occ, (COMB == 2 ? MAXK + 2 : 2 * MAXK + 1));
  hipLaunchKernelGGL((k_combine<MAXK, VEC, COMB>), dim3(g), dim3(kBlock), 0, a->stream, a->ctl, a->vs, f, pass, last,
                     a->pb_flags);
  return g;
}

template <int VEC, int COMB>
int launch_combine_w(int maxk, const nka_hip_state *a, double *f, int pass, int last) {
  // Compact storage (COMB 2) always takes the rolling-window kernel for lists of <= 32 pairs; the all-loads-in-flight form
  // serves it beyond that (passes of 32) -- any other width reaches it only through the diagnostic switch pb_pipe = 0
  // and is padded to 32 (same bits).  The two-vector flavours use every width below the ticket threshold (enqueue_pb).
  if constexpr (COMB == 2) {
#ifdef NKA_DIAGNOSTIC      // (the automatic rule sends every compact list of <= 32 pairs to the window kernel)
    if (maxk <= 4) return launch_combine_1<4, VEC, COMB>(a, f, pass, last);
#endif
    return launch_combine_1<32, VEC, COMB>(a, f, pass, last);
  }
#define CASE(K) \
  case K: return launch_combine_1<K, VEC, COMB>(a, f, pass, last);
  switch (maxk) {
    CASE(4) CASE(8) CASE(12) CASE(16) CASE(20) CASE(24) CASE(28) CASE(32)
  }
#undef CASE
  return 0;
}

// rolling-window PB: ring of W pairs, `bpc` blocks per CU; every width 1..32 (no padding);
// T = 16-byte pieces per thread, stream and tile (2 only for short lists, see launch_combine_win_k)
template <int MAXK, int COMB, int W, int T = 1>
int launch_combine_win_1(const nka_hip_state *a, double *f, int bpc, int base) {
  static const int occ = occupancy_of(k_combine_win<MAXK, COMB, W, T>);
  const int64_t ntile = a->n / (kBlock * 2 * T);
  int64_t g = (int64_t)a->num_cu * std::min(occ, std::max(1, bpc));
  g = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(g, std::max<int64_t>(ntile, 1)), kMaxGrid));
  // Tile tickets (k_combine_win), in-process A/B (profiles/r02/ab_inproc_tile_tickets.txt): PB -3...-10 %
  // at n = 1e8 (the slower the box's static pass, the larger the gain), -2...-4 % at 1.25e7, 0 (m = 20) to
  // -10 % (m = 10) at 1e7, +3 % at 3e6 -> from 64 tiles of 512 elements per block.  ONE counter while a tile carries >= 22
  // words per element and 512 elements (<= ~60 tickets/us), two below that (a single counter saturates).
  constexpr int words = ((COMB == 2 ? MAXK + 2 : 2 * MAXK + 1) + 5) * T;
  int ng = a->pb_tickets;
  if (ng < 0) ng = (ntile * T >= 64 * g) ? (words >= 22 ? 1 : 2) : 0;
  if (ng > 0 && (g % ng != 0 || ntile >= ((int64_t)1 << 31) - 2 * kMaxGrid || !a->tickets)) ng = 0;
  const int tail = (ntile * (kBlock * 2 * T) < a->n) ? 1 : 0;     // the ragged tail has a block of its own (k_combine_win)
  hipLaunchKernelGGL((k_combine_win<MAXK, COMB, W, T>), dim3((int)g + tail), dim3(kBlock), 0, a->stream, a->ctl, a->vs, f,
                     ng > 0 ? a->tickets : nullptr, std::max(ng, 1), a->pb_flags, base);
  return (int)g;
}

// does the automatic rule hand out the tiles of this accelerator's PB by tickets? (see launch_combine_win_1)
bool pb_tickets_apply(const nka_hip_state *a) {
  return a->pb_tickets != 0 && a->tickets && a->n / (kBlock * 2) >= (int64_t)64 * a->num_cu;
}

// One width: the shortest lists (a 512-element tile carries <= 14 words per element) exist with
// double-width tiles too, so that under tickets ONE counter serves them: in-process A/B at n = 1e8,
// compact m = 5 (12 words): 1.511 ms against 1.643 with 512-element tiles and two counters (static
// 1.733); from m = 10 (17 words) on the narrow tile with two counters is as good or better
// (2.110 vs 2.148 ms; two-vector m = 5, 16 words: 2.040 vs 2.028) -- profiles/r02/ab_inproc_tile_tickets.txt.
template <int K, int COMB>
int launch_combine_win_k(const nka_hip_state *a, double *f, int bpc, int base) {
  constexpr int W = (COMB == 2 ? win_ring<K>() : win_ring_pairs<K>());
  constexpr int words = (COMB == 2 ? K + 2 : 2 * K + 1) + 5;
  if constexpr (words <= 14) {
    if (a->pb_tile == 2 || (a->pb_tile < 0 && pb_tickets_apply(a))) return launch_combine_win_1<K, COMB, W, 2>(a, f, bpc, base);
  }
  return launch_combine_win_1<K, COMB, W, 1>(a, f, bpc, base);
}

template <int COMB>
int launch_combine_win_w(int width, const nka_hip_state *a, double *f, int bpc, int base) {
  // One vector per pair, tiles by tickets: a ring of 5 beats the ring of 4 (in-process A/B at m = 20:
  // 3.351 vs 3.406 ms at n = 1e8, 0.435 vs 0.442 at 1.25e7; ring of 10: 3.340 / 0.437), while the static
  // mapping prefers 4 (n = 1e7: 0.359 vs 0.366 ms).  Among the widths 1..32 only 20 has both divisors.
  width = window_width(width);
  if (COMB == 2 && width == 20 && pb_tickets_apply(a)) return launch_combine_win_1<20, 2, 5>(a, f, bpc, base);
#define CASE(K) \
  case K: return launch_combine_win_k<K, COMB>(a, f, bpc, base);
  switch (width) {
    NKA_WIDTH_CASES
  }
#undef CASE
  return 0;
}

int launch_combine_win(int flavor, int width, const nka_hip_state *a, double *f, int bpc, int base = 0) {
  switch (flavor) {
    case NKA_HIP_FLAVOR_F08_VECTOR: return launch_combine_win_w<1>(width, a, f, bpc, base);
    case NKA_HIP_FLAVOR_C: return launch_combine_win_w<2>(width, a, f, bpc, base);
    default: return launch_combine_win_w<0>(width, a, f, bpc, base);
  }
}

// balanced_passes / balanced_widths / heavy_prime / round_up4: host_logic.hpp (pure arithmetic, checked under sanitizers on the CPU)
using nka_host::balanced_passes;
using nka_host::balanced_widths;
using nka_host::heavy_prime;
using nka_host::round_up4;
static_assert(nka_host::kMaxPerPass == kMaxPerPass && nka_host::kListWordLenBits == kListWordLenBits, "host_logic.hpp out of step with nka_kernels.hpp");

// Optional ROCTx ranges around the phases of an update (NKA_HIP_ROCTX=1), for
// rocprofv3 --marker-trace timelines.  The library is looked up at run time so
// that libnka_hip.so has no hard dependency on the profiler SDK.
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char *e = getenv("NKA_HIP_ROCTX");
    if (!e || e[0] != '1') return;
    for (const char *name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
      }
    }
    push = nullptr;
    pop = nullptr;
  }
};
const Roctx &roctx() {
  static const Roctx r;
  return r;
}
struct RoctxRange {
  explicit RoctxRange(const char *name) { if (roctx().push) roctx().push(name); }
  ~RoctxRange() { if (roctx().pop) roctx().pop(); }
};

constexpr int kTimingEvents = 4;

// Upper bound on the list length at the entry of the NEXT update (number seq + 1), tightened -- without any
// synchronisation -- by the newest list word the device has published.  The host's own bookkeeping only counts:
// +1 per update, capped by mvec + 1, -1 per relax; it never learns of a dependence drop (F08:326-345), after which
// PA / PB used to run at the full width with dead ring slots re-reading f (+20 % at m = 5 padded to 8).  The word
// of update u says the list held `len` entries at u's exit; each later update adds at most one entry and each
// later relax that found a pending pair removes exactly one, so at the entry of update seq + 1
//     length <= len + (seq - u) - #{relax after update >= u}.
// Exact whenever the caller has synchronised with the previous update (u == seq): every solver does, once per
// iteration, to read its residual norm.  A caller that never synchronises gets the old bound.
int list_bound_now(nka_hip_state *a) {
  if (a->word_off || !a->list_word) return a->list_ub;
  const unsigned long long w = __atomic_load_n(a->list_word, __ATOMIC_ACQUIRE);
  return nka_host::list_bound_from_word(a->list_ub, w, a->seq, a->word_valid_after, a->relaxed_after);      // host_logic.hpp
}

constexpr size_t kMaxDynamicLds = 160 * 1024;   // gfx950: LDS per CU = per workgroup maximum
constexpr int kMaxMvec = 140;                   // largest mvec with lst_smem_bytes(mvec) <= kMaxDynamicLds
static_assert(lst_smem_bytes(kMaxMvec) <= kMaxDynamicLds && lst_smem_bytes(kMaxMvec + 1) > kMaxDynamicLds,
              "kMaxMvec out of step with lst_smem_bytes");

int record(nka_hip_state *a, int i) {
  if (i == 0) a->timing_this = a->timing_cap > 0 && (a->update_seq++ % a->timing_stride) == 0;
  if (!a->timing_this) return 0;
  const int slot = (int)(a->timing_count % a->timing_cap);
  HIP_TRY(hipEventRecord(a->ev[(size_t)slot * kTimingEvents + i], a->stream));
  return 0;
}

}  // namespace

extern "C" {

const char *nka_hip_last_error(void) { return g_err.c_str(); }

void nka_hip_invalidate_pointer_cache(void) { nka_detail::invalidate_span_cache(); }

int nka_hip_create(nka_hip_t *out, int64_t vlen_local, int32_t mvec, double vtol, int32_t flavor,
                   int32_t device, void *stream) {
  if (!out) return fail(NKA_HIP_EINVAL, "nka_hip_create: out is NULL");
  *out = nullptr;
  // the reference ASSERTs these (F08:190-191, 205)
  if (mvec <= 0) return fail(NKA_HIP_EINVAL, "nka_hip_create: mvec must be > 0");
  if (vlen_local < 0) return fail(NKA_HIP_EINVAL, "nka_hip_create: vlen must be >= 0");
  if (!(vtol > 0.0)) return fail(NKA_HIP_EINVAL, "nka_hip_create: vtol must be > 0");
  if (flavor == NKA_HIP_FLAVOR_DEFAULT) {
    // one rule for every front end: NKA_HIP_FLAVOR, else compact storage (include/nka_hip.h)
    flavor = NKA_HIP_FLAVOR_C;
    if (const char *e = getenv("NKA_HIP_FLAVOR")) {
      const std::string v(e);
      if (v == "f08" || v == "F08" || v == "0") flavor = NKA_HIP_FLAVOR_F08;
      else if (v == "f08vec" || v == "F08VEC" || v == "f08_vector" || v == "1") flavor = NKA_HIP_FLAVOR_F08_VECTOR;
      else if (v == "c" || v == "C" || v == "compact" || v == "2") flavor = NKA_HIP_FLAVOR_C;
      else if (!v.empty()) return fail(NKA_HIP_EINVAL, "NKA_HIP_FLAVOR: expected f08, f08vec or c, got '" + v + "'");
    }
  }
  if (flavor < 0 || flavor > 2) return fail(NKA_HIP_EINVAL, "nka_hip_create: unknown flavor");
  if (mvec > (1 << 20) / 8) return fail(NKA_HIP_EINVAL, "nka_hip_create: mvec is absurdly large");   // (mvec+2)^2 doubles of h
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(NKA_HIP_EINVAL, "nka_hip_create: no such HIP device");
  HIP_TRY(hipSetDevice(device));

  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  auto *a = new nka_hip_state();   // from here on every failure path goes through nka_hip_destroy(a)
  a->device = device;
  a->n = vlen_local;
  a->mvec = mvec;
  a->vtol = vtol;
  a->flavor = flavor;
  a->num_cu = prop.multiProcessorCount;
  snprintf(a->devname, sizeof a->devname, "%s", prop.gcnArchName);
  // The scalar kernels keep h, c and the links of all mvec+1 slots in LDS (8*(mvec+2)^2 bytes and
  // change): gfx950's 160 KiB per workgroup hold mvec <= 140.  Beyond that (the reference has no
  // limit, F08:185-200; practical subspaces are 5..20 vectors) they work in global memory.
  a->state_in_global = lst_smem_bytes(mvec) > kMaxDynamicLds;
  // (kernel variants and launch geometry are chosen automatically; the switches of nka_hip_set_tuning /
  //  nka_hip_set_grid exist for in-process A/B measurements and for the tests that hold every variant to
  //  the same bits -- since round 3 no environment variable selects a variant)
  a->debug = env_int("NKA_HIP_DEBUG", 0) != 0;
  // NKA_HIP_SUMS = auto | rounded | blocked | reference: the sum order a handle starts with, for callers that cannot call
  // nka_hip_set_sum_order (the reference's own programs relinked against the front ends); like NKA_HIP_FLAVOR
  if (const char *e = getenv("NKA_HIP_SUMS")) {
    std::string v(e);
    for (char &ch : v) ch = (char)tolower((unsigned char)ch);
    if (v == "auto" || v.empty()) a->sum_order = NKA_HIP_SUMS_AUTO;
    else if (v == "rounded") a->sum_order = NKA_HIP_SUMS_BLOCKED_ROUNDED;
    else if (v == "blocked" || v == "fast") a->sum_order = NKA_HIP_SUMS_BLOCKED;
    else if (v == "reference") a->sum_order = NKA_HIP_SUMS_REFERENCE_ORDER;
    else {
      delete a;
      return fail(NKA_HIP_EINVAL, "NKA_HIP_SUMS: expected auto, rounded, blocked or reference, got '" + v + "'");
    }
    if (a->sum_order == NKA_HIP_SUMS_REFERENCE_ORDER && mvec > kOrdMaxMvec) a->sum_order = NKA_HIP_SUMS_AUTO;
  }

  // NULL is HIP's default (null) stream, as everywhere in HIP: work is ordered
  // with whatever else the caller enqueues there.
  a->stream = (hipStream_t)stream;

  // above the default 64 KiB of dynamic LDS the kernels must opt in (mvec >= 88)
  if (lst_smem_bytes(mvec) > 64 * 1024 && !a->state_in_global) {
    const int bytes = (int)lst_smem_bytes(mvec);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_restart), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_relax), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
      delete a;
      return fail(NKA_HIP_EHIP, std::string("raising the dynamic LDS limit failed: ") + hipGetErrorString(e));
    }
  }

  {   // the reference-order pass stages whole chunks of every vector in LDS (k_dots_ordered)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dots_ordered), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(double) * (kOrdLdsDoubles + kOrdLdsPad)));
    if (e != hipSuccess) {
      delete a;
      return fail(NKA_HIP_EHIP, std::string("raising the dynamic LDS limit failed: ") + hipGetErrorString(e));
    }
  }

  {   // ... and, for long vectors, a group of products per sum (k_chain_sums) / a block per wavefront (k_chain_blocks)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_sums), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kChainLdsBytes);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_blocks), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kChainLdsBytes);
    if (e != hipSuccess) {
      delete a;
      return fail(NKA_HIP_EHIP, std::string("raising the dynamic LDS limit failed: ") + hipGetErrorString(e));
    }
  }

  // Slot-major storage (F08:163-164, 196): slot k holds n contiguous doubles;
  // the slot stride is padded to 256 B so every slot base allows 16-B loads.
  // (A skew of the slots across HBM channels was measured in round 2 and has no
  //  systematic effect, power-of-two lengths included: profiles/r02/ab_slot_pad.txt.)
  a->vs.n = vlen_local;
  a->vs.stride = ((std::max<int64_t>(vlen_local, 1) + 31) / 32) * 32;
  a->ctl.mvec = mvec;
  const size_t slot_bytes = (size_t)a->vs.stride * sizeof(double) * (size_t)(mvec + 1);
  int rc = 0;
  auto alloc = [&](void **p, size_t bytes) {
    if (rc) return;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();   // reported here: the runtime's sticky last error must not fail the NEXT, unrelated call
      rc = fail(NKA_HIP_ENOMEM, std::string("hipMalloc(") + std::to_string(bytes) + " B): " + hipGetErrorString(e));
    }
  };
  alloc((void **)&a->vs.v, slot_bytes);
  alloc((void **)&a->vs.w, slot_bytes);
  alloc((void **)&a->ctl.ic, sizeof(int32_t) * a->ctl.ic_count());
  alloc((void **)&a->ctl.dc, sizeof(double) * a->ctl.dc_count());
  alloc((void **)&a->partials, sizeof(double) * (size_t)kMaxGrid * (2 * kMaxPerPass + 2));  // NACC columns of k_dots
  alloc((void **)&a->tickets, sizeof(unsigned) * kTicketWords);
  alloc((void **)&a->ctl.pc, sizeof(long long) * a->ctl.pc_count());
  if (!rc) {
    // the list word: fine-grained pinned host memory the device writes and the host polls (no synchronisation)
    void *hw = nullptr;
    hipError_t e = hipHostMalloc(&hw, 64, hipHostMallocMapped | hipHostMallocCoherent);   // (a record of four words, see list_word_publish)
    if (e == hipSuccess) {
      memset(hw, 0, 64);
      void *dp = nullptr;
      if (hipHostGetDevicePointer(&dp, hw, 0) == hipSuccess && dp) {
        a->list_word = static_cast<unsigned long long *>(hw);
        a->ctl.hw = static_cast<unsigned long long *>(dp);
      } else {
        (void)hipGetLastError();
        hipHostFree(hw);           // no device view of it: run without the word (the bound is then the host's own)
      }
    } else {
      (void)hipGetLastError();
    }
  }
  if (rc) {
    nka_hip_destroy(a);
    return rc;
  }
  // zero the control blocks, set vtol, then the device-side restart (F08:198)
  if (hipMemsetAsync(a->tickets, 0, sizeof(unsigned) * kTicketWords, a->stream) != hipSuccess ||
      hipMemsetAsync(a->ctl.ic, 0, sizeof(int32_t) * a->ctl.ic_count(), a->stream) != hipSuccess ||
      hipMemsetAsync(a->ctl.dc, 0, sizeof(double) * a->ctl.dc_count(), a->stream) != hipSuccess ||
      hipMemcpyAsync(a->ctl.dc + DC_VTOL, &a->vtol, sizeof(double), hipMemcpyHostToDevice, a->stream) != hipSuccess) {
    nka_hip_destroy(a);
    return fail(NKA_HIP_EHIP, "initialising the control block failed");
  }
  {
    // slot -> buffer tables: slot k at (k-1)*stride of the two slot-major allocations, as offsets from vs.w (Ctl::pc)
    std::vector<long long> pc((size_t)a->ctl.pc_count(), 0);
    Ctl h = a->ctl;
    h.pc = pc.data();
    pc[PC_OLD_W] = pc[PC_OLD_V] = kNoBuffer;      // (offset 0 is a real buffer: slot 1 of w)
    for (int k = 1; k <= mvec + 1; k++) {
      h.wtab()[k] = buffer_offset(a, a->vs.w + (size_t)(k - 1) * a->vs.stride);
      h.vtab()[k] = buffer_offset(a, a->vs.v + (size_t)(k - 1) * a->vs.stride);
    }
    if (hipMemcpy(a->ctl.pc, pc.data(), sizeof(long long) * pc.size(), hipMemcpyHostToDevice) != hipSuccess) {
      nka_hip_destroy(a);
      return fail(NKA_HIP_EHIP, "initialising the pointer block failed");
    }
  }
  // the memcpy source is a->vtol (heap): safe until the stream reaches it; sync to be simple
  hipStreamSynchronize(a->stream);
  rc = nka_hip_restart(a);
  if (rc) {
    nka_hip_destroy(a);
    return rc;
  }
  *out = a;
  return 0;
}

int nka_hip_capture_safe(nka_hip_t a) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  // (the debug mode reads the state back after every update, a user dot product runs on the host, and a caller's all-reduce
  //  hook is a host callback that a replay would not call again: none of them can be captured; the built-in RCCL hook only
  //  enqueues on the stream)
  const bool user_hook = a->allreduce && a->allreduce != rccl_allreduce && a->allreduce != p2p_allreduce;
  return (a->pending && list_bound_now(a) >= a->mvec + 1 && !a->debug && !a->host_dot && !user_hook) ? 1 : 0;
}

int nka_hip_list_bound(nka_hip_t a) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  return list_bound_now(a);
}

int nka_hip_set_stream(nka_hip_t a, void *stream) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  hipStream_t ns = (hipStream_t)stream;
  if (ns == a->stream) return 0;
  HIP_TRY(hipSetDevice(a->device));
  hipEvent_t ev;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e1 = hipEventRecord(ev, a->stream);
  hipError_t e2 = (e1 == hipSuccess) ? hipStreamWaitEvent(ns, ev, 0) : e1;
  hipEventDestroy(ev);
  if (e2 != hipSuccess) return fail(NKA_HIP_EHIP, std::string("set_stream: ") + hipGetErrorString(e2));
  a->stream = ns;
  return 0;
}

int nka_hip_destroy(nka_hip_t a) {
  if (!a) return 0;
  nka_detail::invalidate_span_cache();
  hipSetDevice(a->device);
  hipStreamSynchronize(a->stream);
  if (a->comm) rccl().CommDestroy(a->comm);
  nka_hip_p2p_detach(a);
  hipFree(a->vs.v);
  hipFree(a->vs.w);
  hipFree(a->ctl.ic);
  hipFree(a->ctl.dc);
  hipFree(a->partials);
  hipFree(a->chain_pred);
  hipFree(a->chain_summ);
  hipFree(a->tickets);
  hipFree(a->ctl.pc);
  for (void *p : a->extra_allocs) hipFree(p);
  if (a->list_word) hipHostFree(a->list_word);
  hipFree(a->f_stage);
  hipFree(a->hd_scratch);
  for (auto &e : a->ev)
    if (e) hipEventDestroy(e);
  a->ev.clear();
  delete a;
  return 0;
}

// The slot -> buffer tables as they stand on the device (synchronises).  t.wtab()[k], t.vtab()[k] for k = 1..mvec+1.
static int fetch_tables(nka_hip_t a, std::vector<long long> &pc, Ctl &t) {
  pc.assign((size_t)a->ctl.pc_count(), 0);
  HIP_TRY(hipMemcpyAsync(pc.data(), a->ctl.pc, sizeof(long long) * pc.size(), hipMemcpyDeviceToHost, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  t = a->ctl;
  t.pc = pc.data();
  return 0;
}
// w / v buffer of a slot for the host-side paths (queries, the user-dot-product path, deep copies)
static const double *slot_buffer(nka_hip_t a, const Ctl *tables, bool v, int slot) {
  if (tables) return buffer_at(a, v ? tables->vtab()[slot] : tables->wtab()[slot]);
  return (v ? a->vs.v : a->vs.w) + (size_t)(slot - 1) * a->vs.stride;
}
// After the control blocks of a copy have been filled from another object: the addresses of PA's plan through THIS
// object's tables (the combine plan is rewritten by every scalar step before PB reads it).
static __global__ void k_plan_pointers_from_slots(Ctl ctl) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int n = ctl.ic[IC_PLAN_NOLDER];
  for (int j = 0; j < n; j++) ctl.plan_w()[j] = ctl.wtab()[ctl.plan_slots()[j]];
  ctl.pc[PC_FIRST_W] = ctl.wtab()[ctl.ic[IC_PLAN_FIRST]];
}

int nka_hip_clone(nka_hip_t src, nka_hip_t *out) {
  if (!src || !out) return fail(NKA_HIP_EINVAL, "nka_hip_clone: null argument");
  *out = nullptr;
  nka_hip_t b = nullptr;
  if (int rc = nka_hip_create(&b, src->n, src->mvec, src->vtol, src->flavor, src->device, (void *)src->stream)) return rc;
  // same storage geometry by construction (n and mvec decide it)
  if (b->vs.stride != src->vs.stride || b->ctl.ic_count() != src->ctl.ic_count() || b->ctl.dc_count() != src->ctl.dc_count()) {
    nka_hip_destroy(b);
    return fail(NKA_HIP_ESTATE, "nka_hip_clone: storage geometry differs");
  }
  const size_t slot_bytes = (size_t)src->vs.stride * sizeof(double) * (size_t)(src->mvec + 1);
  hipStream_t s = src->stream;
  hipError_t e = hipSuccess;
  if (!src->swapped) {
    e = hipMemcpyAsync(b->vs.v, src->vs.v, slot_bytes, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(b->vs.w, src->vs.w, slot_bytes, hipMemcpyDeviceToDevice, s);
  } else {
    // out-of-place updates have moved src's vectors into other buffers (some of them the caller's): the copy gets
    // its own slot-major storage, filled slot by slot through src's tables
    std::vector<long long> pc;
    Ctl t{};
    if (int rc = fetch_tables(src, pc, t)) {
      nka_hip_destroy(b);
      return rc;
    }
    const size_t nb = sizeof(double) * (size_t)src->n;
    for (int k = 1; k <= src->mvec + 1 && e == hipSuccess && nb > 0; k++) {
      e = hipMemcpyAsync(b->vs.w + (size_t)(k - 1) * b->vs.stride, buffer_at(src, t.wtab()[k]), nb, hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess)
        e = hipMemcpyAsync(b->vs.v + (size_t)(k - 1) * b->vs.stride, buffer_at(src, t.vtab()[k]), nb, hipMemcpyDeviceToDevice, s);
    }
  }
  if (e == hipSuccess) e = hipMemcpyAsync(b->ctl.ic, src->ctl.ic, sizeof(int32_t) * src->ctl.ic_count(), hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(b->ctl.dc, src->ctl.dc, sizeof(double) * src->ctl.dc_count(), hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_plan_pointers_from_slots, dim3(1), dim3(1), 0, s, b->ctl);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    nka_hip_destroy(b);
    return fail(NKA_HIP_EHIP, std::string("nka_hip_clone: ") + hipGetErrorString(e));
  }
  // what the host knows without looking, and the caller's choices
  b->pending = src->pending;
  b->list_ub = src->list_ub;
  b->bpc[0] = src->bpc[0];
  b->bpc[1] = src->bpc[1];
  b->pa_pipe = src->pa_pipe;
  b->pb_pipe = src->pb_pipe;
  b->pb_tile = src->pb_tile;
  b->pb_tickets = src->pb_tickets;
  b->prime_pad = src->prime_pad;
  b->chain_walk = src->chain_walk;
  b->chain_many = src->chain_many;      // (the copy allocates its own block arrays when it needs them)
  b->chain_pred = nullptr; b->chain_summ = nullptr; b->chain_cap = 0;
  b->serial_solve = src->serial_solve;
  b->sum_order = src->sum_order;
  b->shard_rank = src->shard_rank;
  b->shard_n = src->shard_n;
  b->debug = src->debug;
  if (src->allreduce != rccl_allreduce && src->allreduce != p2p_allreduce) {      // a user hook travels with the object, the RCCL communicator (or the peers' mailboxes) does not
    b->allreduce = src->allreduce;
    b->allreduce_ctx = src->allreduce_ctx;
    b->needs_comm = src->needs_comm;
  } else {
    // ... and the copy must not run without one: its sums would be rank-local, the replicas would diverge and nothing
    // would say so (the reference's deep copy keeps its dp, F08:161).  accel_update fails with NKA_HIP_ECOMM until the
    // caller gives the copy a communicator (nka_hip_comm_init_rank, collectively) or a hook (nka_hip_set_allreduce;
    // NULL = "this copy really is single-rank").
    b->needs_comm = true;
  }
  b->host_dot = src->host_dot;
  b->host_dot_ctx = src->host_dot_ctx;
  *out = b;
  return 0;
}

int nka_hip_restart(nka_hip_t a) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (a->poisoned) return fail(NKA_HIP_ESTATE, "restart: an earlier update failed after its scalar step had been enqueued: destroy the handle");
  HIP_TRY(hipSetDevice(a->device));
  hipLaunchKernelGGL(k_restart, dim3(1), dim3(kSolveThreads), a->state_in_global ? 0 : lst_smem_bytes(a->mvec), a->stream,
                     a->ctl, a->state_in_global ? 1 : 0);
  HIP_TRY(hipGetLastError());
  a->pending = false;
  a->list_ub = 0;
  a->word_valid_after = a->seq;      // whatever the updates enqueued so far publish describes the flushed list
  a->relaxed_after.clear();
  return 0;
}

int nka_hip_relax(nka_hip_t a) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (a->poisoned) return fail(NKA_HIP_ESTATE, "relax: an earlier update failed after its scalar step had been enqueued: destroy the handle");
  HIP_TRY(hipSetDevice(a->device));
  hipLaunchKernelGGL(k_relax, dim3(1), dim3(kSolveThreads), a->state_in_global ? 0 : lst_smem_bytes(a->mvec), a->stream,
                     a->ctl, a->state_in_global ? 1 : 0);
  HIP_TRY(hipGetLastError());
  if (a->pending) {
    a->pending = false;
    a->list_ub = std::max(a->list_ub - 1, 0);
    if (!a->word_off && a->list_word) a->relaxed_after.push_back(a->seq);   // (only the list word's arithmetic reads these)
    if (a->relaxed_after.size() > 4096) a->relaxed_after.erase(a->relaxed_after.begin(), a->relaxed_after.begin() + 2048);
    // (a caller that relaxes thousands of times without ever letting a word through: dropping old entries can only
    //  LOOSEN the bound, list_bound_now subtracts one per entry)
  }
  return 0;
}

int nka_hip_set_vec_tol(nka_hip_t a, double vtol) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (!(vtol > 0.0)) return fail(NKA_HIP_EINVAL, "set_vec_tol: vtol must be > 0");  // F08:205
  HIP_TRY(hipSetDevice(a->device));
  a->vtol = vtol;
  HIP_TRY(hipMemcpyAsync(a->ctl.dc + DC_VTOL, &a->vtol, sizeof(double), hipMemcpyHostToDevice, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  return 0;
}

// ---- the three stages of an update, enqueued on the handle's stream -------------
static int enqueue_solve(nka_hip_t a, int mode, long long swap_w = kNoBuffer, long long swap_v = kNoBuffer, bool gather = false) {
  hipStream_t s = a->stream;
  const P2P x = gather ? a->p2p : P2P{};      // the scalar step starts by gathering the sums of all ranks (peer-to-peer exchange)
  if (a->mvec + 1 <= kSolveWaveMax && !a->serial_solve) {
    const size_t sm = solve_wave_smem_bytes(a->mvec);
    const int nl = a->mvec + 1;
    // (identity tables are COMPUTED from the stride only while no table entry can have changed AND no graph holds this
    //  launch: a replay re-issues it with the arguments frozen at capture, a later out-of-place update would be invisible)
    const long long id_stride = (a->swapped || a->captured) ? 0 : (long long)a->vs.stride, id_vbase = buffer_offset(a, a->vs.v);
#define ROWS(NL) \
  hipLaunchKernelGGL((k_solve_rows<NL>), dim3(1), dim3(kSolveThreads), sm, s, a->ctl, mode, swap_w, swap_v, id_stride, id_vbase, x)
    if (nl <= 6) ROWS(6);
    else if (nl <= 11) ROWS(11);
    else if (nl <= 21) ROWS(21);
    else if (nl <= 33) ROWS(33);
    else if (nl <= 48) ROWS(48);
    else {
      // 49..63 list entries (mvec 48..62): the working arrays exceed the 64 KiB a kernel gets without asking
      // (once per HANDLE, i.e. per device -- a process-wide static would serve only the first GPU of a process)
      if (!a->solve63_raised) {
        const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_rows<63>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      (int)solve_wave_smem_bytes(kSolveWaveMax - 1));
        if (raised != hipSuccess) return fail(NKA_HIP_EHIP, std::string("raising the dynamic LDS limit failed: ") + hipGetErrorString(raised));
        a->solve63_raised = true;
      }
      ROWS(63);
    }
#undef ROWS
  } else {
    hipLaunchKernelGGL(k_solve, dim3(1), dim3(kSolveThreads), a->state_in_global ? 0 : lst_smem_bytes(a->mvec), s, a->ctl, mode,
                       a->state_in_global ? 1 : 0, 0, swap_w, swap_v, x);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// PA: all inner products in one pure-read pass (F08:266-267, 286-290, 371).
// Unaligned f (not 16-B aligned) takes scalar loads with the narrow unroll.
static void enqueue_pa(nka_hip_t a, const double *f, int vec, int older_ub) {
  const int maxl = (vec == 1) ? 4 : (older_ub > kMaxPerPass ? kMaxPerPass : round_up4(older_ub));
  const int npass = std::max(1, (older_ub + maxl - 1) / maxl);
  // Automatic choice, from in-process A/B runs (tools/ab_inproc.py, profiles/r02/ab_inproc_*.txt):
  // the rolling-window kernel (small ring, one block per CU) wins by 4-11 % whenever the unroll width
  // carries no padding (m = 20: 2.58 vs 2.75 ms at n = 1e8, 0.374 vs 0.394 at 1.25e7, 44 vs 48 us
  // at 1e6; m = 10: 1.43 vs 1.60 ms at n = 1e8) -- padded entries are cache hits that occupy ring
  // slots and starve the window (m = 5 padded to 8 at n = 1e8: +20 %), which is why it is instantiated
  // for EVERY width 1..32; the fallback k_dots (multi-pass) serves mvec > 32 and unaligned f.
  int pa_pipe = a->pa_pipe;
  // the rolling-window kernel exists for every width 1..32, so the list needs no padding
  const bool exact = vec == 2 && older_ub >= 1 && older_ub <= kMaxPerPass;
  if (pa_pipe < 0) pa_pipe = exact ? 201 : 0;
  if (vec == 2 && npass == 1 && pa_pipe > 200 && pa_pipe < 210) {     // rolling window, 200 + blocks per CU
    int w = exact ? older_ub : maxl;
    if (a->prime_pad != 0 && heavy_prime(w)) w++;       // one dead ring slot instead of a ring as wide as the list (state: prime_pad)
    launch_dots_win(w, a, f, std::max(1, pa_pipe - 200));
  } else if (vec == 2 && older_ub > kMaxPerPass && a->pa_pipe != 0) {  // a long list: balanced passes of the window kernel
    const int np = balanced_passes(older_ub);
    std::vector<int> wd((size_t)np);
    balanced_widths(older_ub, np, wd.data());
    for (int p = 0, base = 0; p < np; p++) {
      launch_dots_win(wd[(size_t)p], a, f, a->pa_pipe > 200 ? a->pa_pipe - 200 : 1, base, p, older_ub);
      base += wd[(size_t)p];
    }
  } else {
    for (int p = 0; p < npass; p++) {
      if (vec == 2) launch_dots_w(maxl, a, f, p, npass);
      else launch_dots_1<4, 1>(a, f, p, npass);
    }
  }
}

// PB: normalise + combine + ring stores (F08:282-283, 361, 395-404).
// After the subspace update the list holds at most min(list_ub, mvec) vectors.
static int enqueue_pb(nka_hip_t a, double *f, int vec, int comb_ub) {
  const int maxk = (vec == 1) ? 4 : (comb_ub > kMaxPerPass ? kMaxPerPass : round_up4(comb_ub));
  const int npass = std::max(1, (comb_ub + maxk - 1) / maxk);
  // Automatic choice (in-process A/B, profiles/r02/ab_inproc_window_matrix.txt).  Compact flavour
  // (one vector per pair): the rolling-window kernel (ring of 4, one block per CU) is as fast or
  // faster than k_combine over n = 1e6..1e8, m = 5..20 (-2 % at n = 1e8 m = 20, -7...-11 % at m = 5/10).
  // Two-vector flavours (ring of 2 pairs): with the static tile mapping the window wins only once the
  // pass moves >~ 8 GB and loses below (n = 1e7: +3 % at m = 20, +12 % at m = 10), where k_combine's
  // deeper queue hides the ramp at both ends of the launch; with tile tickets (taken from 64 tiles per
  // block, launch_combine_win_1) it wins from there on: n = 1.25e7, m = 20: 0.782 vs 0.837 ms, n = 2.5e7:
  // 1.540 vs 1.726 ms (profiles/r02/ab_inproc_tile_tickets.txt).
  int pipe = a->pb_pipe;
  if (pipe < 0) {
    const bool tickets = pb_tickets_apply(a);
    pipe = (a->flavor == NKA_HIP_FLAVOR_C || tickets || (double)a->n * (2.0 * comb_ub + 6.0) >= 1.0e9) ? 201 : 0;
  }
  if (vec == 2 && pipe > 200 && pipe < 210 && comb_ub <= kMaxPerPass) {   // rolling window, 200 + blocks per CU
    int w = std::max(comb_ub, 1);                                             // exact width: no padding ...
    if (a->prime_pad != 0 && heavy_prime(w)) w++;                            // ... but for 23, 29, 31 (see prime_pad): one dead slot
    launch_combine_win(a->flavor, w, a, f, pipe - 200);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (vec == 2 && comb_ub > kMaxPerPass && a->pb_pipe != 0 && !(a->pb_flags & (kPbNoStoreW | kPbNoStoreF))) {
    // a long list, in place: balanced passes of the window kernel, f carrying the running value (k_combine_win, PASSES);
    // the out-of-place entry keeps the all-loads-in-flight passes below (its running value lives in v_new)
    const int np = balanced_passes(comb_ub), keep = a->pb_flags;
    std::vector<int> wd((size_t)np);
    balanced_widths(comb_ub, np, wd.data());
    for (int p = 0, base = 0; p < np; p++) {
      a->pb_flags = keep | (p > 0 ? kPbNotFirst : 0) | (p + 1 < np ? kPbNotLast : 0);
      launch_combine_win(a->flavor, wd[(size_t)p], a, f, a->pb_pipe > 200 ? a->pb_pipe - 200 : 1, base);
      base += wd[(size_t)p];
    }
    a->pb_flags = keep;
    HIP_TRY(hipGetLastError());
    return 0;
  }
  for (int p = 0; p < npass; p++) {
    const int last = (p == npass - 1);
    if (vec == 2) {
      switch (a->flavor) {
        case NKA_HIP_FLAVOR_F08_VECTOR: launch_combine_w<2, 1>(maxk, a, f, p, last); break;
        case NKA_HIP_FLAVOR_C: launch_combine_w<2, 2>(maxk, a, f, p, last); break;
        default: launch_combine_w<2, 0>(maxk, a, f, p, last);
      }
    } else {
      switch (a->flavor) {
        case NKA_HIP_FLAVOR_F08_VECTOR: launch_combine_1<4, 1, 1>(a, f, p, last); break;
        case NKA_HIP_FLAVOR_C: launch_combine_1<4, 1, 2>(a, f, p, last); break;
        default: launch_combine_1<4, 1, 0>(a, f, p, last);
      }
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// Operands of the user's dot product, formed ON THE DEVICE with the rounding of the
// stored vectors: out = w1 - f (F08:266), then out = out/s or (1/s)*out (F08:283 / F08V:256).
static __global__ __launch_bounds__(kBlock) void k_hostdot_diff(int64_t n, const double *__restrict__ w1,
                                                                const double *__restrict__ f, double *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    out[i] = w1[i] - f[i];
}
static __global__ __launch_bounds__(kBlock) void k_hostdot_normalise(int64_t n, double *__restrict__ d, double s, int rcp) {
  const double rs = 1.0 / s;
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    d[i] = rcp ? rs * d[i] : d[i] / s;
}

static int fetch_state(nka_hip_t a, std::vector<int32_t> &ic, std::vector<double> &dc);

// The inner products AND the scalar step of an update through the USER's host dot product
// (nka_hip_set_host_dot), as the reference's own sequence of dp calls on host copies of the operands, same
// operands, same order, no call the reference does not make:
//   dp(d,d) with d = w1 - f (F08:266-267); if s != 0, on the normalised w1' = d/s the Gram row dp(w1', w_k) for
//   every older list entry in list order (F08:286-290) -> device: norm, relax, Gram row, factorisation with its
//   drop decisions (k_solve phase 1) -> the list as it now stands is read back -> the projections dp(f, w_j) for
//   j = first ... last of THAT list (F08:371: after the drops) -> device: substitutions, plans, prepend (phase 2).
// d and w1' are formed by device kernels (the host only copies, calls dp and takes one square root to know
// whether s == 0).  Synchronous and PCIe-bound by construction: a compatibility path, never the measured one.
static int host_dot_update_scalars(nka_hip_t a, const double *f, int mode) {
  const int64_t n = a->n;
  const int mvec = a->mvec, m1 = mvec + 1;
  const int rcp = (a->flavor == NKA_HIP_FLAVOR_F08_VECTOR) ? 1 : 0;
  HIP_TRY(hipStreamSynchronize(a->stream));
  std::vector<int32_t> ic;
  std::vector<double> dc;
  if (int rc = fetch_state(a, ic, dc)) return rc;
  const int pending = ic[IC_PLAN_PENDING], first = ic[IC_PLAN_FIRST], nolder = ic[IC_PLAN_NOLDER];
  const int32_t *slots = ic.data() + (a->ctl.plan_slots() - a->ctl.ic);
  if (nolder < 0 || nolder > mvec + 1 || (pending && (first < 1 || first > mvec + 1)))
    return fail(NKA_HIP_ESTATE, "host dot path: corrupt dot plan on the device");
  const size_t nb = sizeof(double) * (size_t)n;
  std::vector<long long> pcs;      // (after out-of-place updates the vectors are where the tables say)
  Ctl tabs{};
  if (a->swapped)
    if (int rc = fetch_tables(a, pcs, tabs)) return rc;
  const Ctl *tp = a->swapped ? &tabs : nullptr;
  if (pending && !a->hd_scratch) HIP_TRY(hipMalloc((void **)&a->hd_scratch, sizeof(double) * (size_t)std::max<int64_t>(n, 1)));
  const int g = (int)std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock, (int64_t)a->num_cu * 8));
  std::vector<double> hf((size_t)n), hw1(pending ? (size_t)n : 0), hk((size_t)n);
  std::vector<double> red((size_t)a->ctl.red_count(), 0.0);
  if (n > 0) HIP_TRY(hipMemcpy(hf.data(), f, nb, hipMemcpyDeviceToHost));
  bool normed = false;
  if (pending) {
    hipLaunchKernelGGL(k_hostdot_diff, dim3(g), dim3(kBlock), 0, a->stream, n,
                       slot_buffer(a, tp, false, first), f, a->hd_scratch);
    HIP_TRY(hipGetLastError());
    if (n > 0) HIP_TRY(hipMemcpyAsync(hw1.data(), a->hd_scratch, nb, hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(hipStreamSynchronize(a->stream));
    red[0] = a->host_dot(a->host_dot_ctx, n, hw1.data(), hw1.data());           // F08:267 (the device takes the sqrt again)
    const double s = std::sqrt(red[0]);
    if (s != 0.0) {                                                               // F08:275, else relax
      normed = true;
      hipLaunchKernelGGL(k_hostdot_normalise, dim3(g), dim3(kBlock), 0, a->stream, n, a->hd_scratch, s, rcp);
      HIP_TRY(hipGetLastError());
      if (n > 0) HIP_TRY(hipMemcpyAsync(hw1.data(), a->hd_scratch, nb, hipMemcpyDeviceToHost, a->stream));
      HIP_TRY(hipStreamSynchronize(a->stream));
      for (int p = 0; p < nolder; p++) {                                          // F08:286-290, list order
        const int slot = slots[p];
        if (slot < 1 || slot > mvec + 1) return fail(NKA_HIP_ESTATE, "host dot path: slot out of range in the dot plan");
        if (n > 0) HIP_TRY(hipMemcpy(hk.data(), slot_buffer(a, tp, false, slot), nb, hipMemcpyDeviceToHost));
        red[2 + p] = a->host_dot(a->host_dot_ctx, n, hw1.data(), hk.data());
      }
    }
  }
  // (uploads on the handle's stream, from buffers that live until the synchronisation behind them: ordered with the
  //  kernels whatever kind of stream the caller bound the handle to)
  HIP_TRY(hipMemcpyAsync(a->ctl.red(), red.data(), sizeof(double) * red.size(), hipMemcpyHostToDevice, a->stream));
  // ---- device: norm, s == 0 -> relax, Gram row, Cholesky with drops (the reference's loops on one lane)
  const size_t smem = a->state_in_global ? 0 : lst_smem_bytes(a->mvec);
  hipLaunchKernelGGL(k_solve, dim3(1), dim3(kSolveThreads), smem, a->stream, a->ctl, mode | kSolvePrenorm,
                     a->state_in_global ? 1 : 0, 1, kNoBuffer, kNoBuffer, P2P{});
  HIP_TRY(hipGetLastError());
  // Phase 1 has changed the lists, the free list, h and the flags on the device.  Whatever fails between here and
  // phase 2 (a copy, a corrupt list, the user's dp) must not leave a half-applied update behind: the control blocks
  // as they stood at entry (ic0, dc0: read at the top) are written back, and accel_update's promise holds -- a failed
  // update is NOT done and the call can be repeated.
  const std::vector<int32_t> ic0 = ic;
  const std::vector<double> dc0 = dc;
  auto undo = [&](int rc) {
    const std::string msg = g_err;
    (void)hipStreamSynchronize(a->stream);
    (void)hipMemcpy(a->ctl.ic, ic0.data(), sizeof(int32_t) * ic0.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(a->ctl.dc, dc0.data(), sizeof(double) * dc0.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_plan_pointers_from_slots, dim3(1), dim3(1), 0, a->stream, a->ctl);    // (the plan's addresses too)
    (void)hipStreamSynchronize(a->stream);
    (void)hipGetLastError();
    g_err = msg;
    return rc;
  };
#define HIP_TRY_UNDO(expr)                                                                                  \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      (void)hipGetLastError();                                                                              \
      return undo(fail(NKA_HIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)));                   \
    }                                                                                                       \
  } while (0)
  // ---- the list after the drops; its projections in list order, first ... last (F08:369-371)
  if (int rc = fetch_state(a, ic, dc)) return undo(rc);
  std::vector<double> c((size_t)m1 + 1, 0.0);
  if (ic[IC_SUBSPACE]) {
    const int32_t *next = ic.data() + IC_HEADER;
    int steps = 0;
    for (int j = ic[IC_FIRST]; j != 0; j = next[j]) {
      if (j < 1 || j > m1 || ++steps > m1) return undo(fail(NKA_HIP_ESTATE, "host dot path: corrupt list on the device"));
      const double *wj = hk.data();
      if (normed && j == first) {
        wj = hw1.data();                          // the new w1' (not stored yet: the combine pass writes it)
      } else if (n > 0) {
        HIP_TRY_UNDO(hipMemcpy(hk.data(), slot_buffer(a, tp, false, j), nb, hipMemcpyDeviceToHost));
      }
      c[(size_t)j] = a->host_dot(a->host_dot_ctx, n, hf.data(), wj);              // F08:371
    }
  }
  HIP_TRY_UNDO(hipMemcpyAsync(a->ctl.c(), c.data(), sizeof(double) * c.size(), hipMemcpyHostToDevice, a->stream));
  // ---- device: new slot, substitutions, plans, prepend
  hipLaunchKernelGGL(k_solve, dim3(1), dim3(kSolveThreads), smem, a->stream, a->ctl, mode | kSolvePrenorm,
                     a->state_in_global ? 1 : 0, 2, kNoBuffer, kNoBuffer, P2P{});
  HIP_TRY_UNDO(hipGetLastError());
  HIP_TRY_UNDO(hipStreamSynchronize(a->stream));     // (c[] and red[] above are read by the stream until here)
#undef HIP_TRY_UNDO
  return 0;
}

// Reference-order sums: asked for, or free -- a vector of at most kOrdAutoMax elements on a single rank.
static bool ordered_sums(const nka_hip_state *a) {
  if (a->allreduce || a->host_dot || a->mvec > kOrdMaxMvec) return false;
  return a->sum_order == NKA_HIP_SUMS_REFERENCE_ORDER || (a->sum_order == NKA_HIP_SUMS_AUTO && a->n <= kOrdAutoMax);
}

// Does [p, p + n) touch memory the library holds -- the slot-major allocations or a buffer it has taken over -- that it
// has NOT lent to the caller?  (Ordered sets: a caller that hands over a fresh buffer every iteration makes them long.)
static bool held_by_library(const nka_hip_state *a, const double *p) {
  return a->book.held(p, a->n, a->vs.w, a->vs.v, a->vs.stride * (int64_t)(a->mvec + 1));
}

// Reference-order sums of a SHARDED accelerator (validation mode).  The reference's sum over the global
// vector is ONE chain of additions through the slices in rank order (its dp is a global dot product, F08:209-219; summed
// sequentially over the whole vector that is what a single-rank run of the reference computes).  So the ranks take turns:
// in round r rank r continues the running sums from the prefix over ranks 0..r-1 (k_dots_ordered, carry) while every other
// rank contributes zeros, and the installed all-reduce hands the new prefix to everybody -- x + 0 + ... + 0 is exact in any
// order, so ANY hook that sums serves as the chain's transport.  N rounds for the norm (the Gram row needs the GLOBAL s
// before w1' = d/s can be rounded), N rounds for the rows: 2N small exchanges and the serial walk of the whole global
// vector per update -- validation speed, bits of the single-rank compiled reference.
// The longest vectors: the block summaries of every sum by the whole device, then one wavefront per sum applies them
// (k_chain_blocks / k_chain_predict / k_chain_apply, nka_kernels.hpp).  false: not this way (short vectors, no memory for the
// block arrays, a capture in progress before they exist) -- the caller takes one compute unit per sum.
constexpr long long kChainManyMin = 1 << 19;
static bool chain_many_ready(nka_hip_t a, int nsum_max, long long n = -1) {
  if (n < 0) n = a->n;
  const long long nfull = n / kChainBlock;
  if (a->chain_many == 0 || nfull < 1 || (a->chain_many < 0 && n < kChainManyMin)) return false;
  const long long need = nfull * nsum_max;
  if (a->chain_cap >= need) return true;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(a->stream, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (cs != hipStreamCaptureStatusNone) return false;                 // (no allocation inside a capture)
  hipFree(a->chain_pred); hipFree(a->chain_summ);
  a->chain_pred = nullptr; a->chain_summ = nullptr; a->chain_cap = 0;
  void *p = nullptr, *q = nullptr;
  if (hipMalloc(&p, sizeof(double) * (size_t)need) != hipSuccess || hipMalloc(&q, sizeof(ChainSummary) * (size_t)need) != hipSuccess) {
    (void)hipGetLastError();
    hipFree(p); hipFree(q);
    return false;
  }
  a->chain_pred = static_cast<double *>(p); a->chain_summ = q; a->chain_cap = need;
  return true;
}
static int chain_many_stage(nka_hip_t a, const double *f, int rcp, int set, int with_f, int ub, int nsum, const double *probe = nullptr,
                            long long n_probe = -1) {
  hipStream_t s = a->stream;
  Vecs vs = a->vs;
  if (n_probe >= 0) vs.n = n_probe;                    // (the diagnostic entry sums arrays of its own length)
  const long long nfull = vs.n / kChainBlock;
  const long long items = nfull * nsum;
  const unsigned grid = (unsigned)((items + kChainWaves - 1) / kChainWaves);
  ChainSummary *summ = static_cast<ChainSummary *>(a->chain_summ);
  hipLaunchKernelGGL(k_chain_blocks, dim3(grid), dim3(kChainThreads), kChainLdsBytes, s, a->ctl, vs, f, rcp, set, with_f, ub, nsum, nfull,
                     a->chain_pred, summ, 0, probe);
  hipLaunchKernelGGL(k_chain_predict, dim3(nsum), dim3(kChainPredictThreads), 0, s, a->ctl, vs, f, rcp, set, with_f, ub, nfull, a->chain_pred,
                     probe);
  hipLaunchKernelGGL(k_chain_blocks, dim3(grid), dim3(kChainThreads), kChainLdsBytes, s, a->ctl, vs, f, rcp, set, with_f, ub, nsum, nfull,
                     a->chain_pred, summ, 1, probe);
  hipLaunchKernelGGL(k_chain_apply, dim3(nsum), dim3(64), 0, s, a->ctl, vs, f, rcp, set, with_f, ub, nfull,
                     static_cast<const ChainSummary *>(summ), a->chain_walk, probe);
  HIP_TRY(hipGetLastError());
  return 0;
}

// (beyond one chunk of k_dots_ordered: one workgroup per sum, k_chain_sums)
static bool chain_per_sum(const nka_hip_state *a, int rows) { return a->n > ord_chunk(rows); }

static int ordered_chain(nka_hip_t a, const double *f, int mode, int older_ub) {
  hipStream_t s = a->stream;
  const int N = a->shard_n, me = a->shard_rank;
  const int rows = 2 + older_ub, count = a->ctl.red_count();
  double *red = a->ctl.red();
  const bool per_sum = chain_per_sum(a, rows);
  auto exchange = [&](double *buf, int cnt) -> int {
    if (int rc = a->allreduce(a->allreduce_ctx, buf, cnt, s)) return rc < 0 ? rc : fail(NKA_HIP_ECOMM, "allreduce hook failed");
    return 0;
  };
  auto round = [&](int r, int phase, double *buf, int cnt) -> int {
    if (r == me && per_sum) {
      // every chain continues from what red[] holds: the prefix of the ranks before this one, or zeros on the first rank
      // (the norm rounds exchange red[0] alone; the rows rounds everything behind it)
      if (r == 0) HIP_TRY(hipMemsetAsync(buf, 0, sizeof(double) * (size_t)cnt, s));
      if (chain_many_ready(a, 1 + 2 * a->mvec)) {        // (the longest slices: the whole device summarises, one wavefront per sum applies)
        if (int rc = phase == kOrdNorm ? chain_many_stage(a, f, mode & kSolveRcp, (int)kChainNorm, 0, older_ub, 1)
                                       : chain_many_stage(a, f, mode & kSolveRcp, (int)kChainRows, 1, older_ub, 1 + 2 * older_ub))
          return rc;
      } else if (phase == kOrdNorm) {
        hipLaunchKernelGGL(k_chain_sums, dim3(1), dim3(kChainThreads), kChainLdsBytes, s, a->ctl, a->vs, f, mode & kSolveRcp, (int)kChainNorm, 0, older_ub, a->chain_walk, (const double *)nullptr);
      } else {
        hipLaunchKernelGGL(k_chain_sums, dim3(1 + 2 * older_ub), dim3(kChainThreads), kChainLdsBytes, s, a->ctl, a->vs, f, mode & kSolveRcp,
                           (int)kChainRows, 1, older_ub, a->chain_walk, (const double *)nullptr);
      }
      HIP_TRY(hipGetLastError());
    } else if (r == me) {
      hipLaunchKernelGGL(k_dots_ordered, dim3(1), dim3(kOrdThreads), ord_lds_bytes(rows), s, a->ctl, a->vs, f, mode & kSolveRcp,
                         ord_chunk(rows), phase, r > 0 ? 1 : 0);
      HIP_TRY(hipGetLastError());
    } else {
      HIP_TRY(hipMemsetAsync(buf, 0, sizeof(double) * (size_t)cnt, s));
    }
    return exchange(buf, cnt);
  };
  if (a->pending)
    for (int r = 0; r < N; r++)
      if (int rc = round(r, kOrdNorm, red, 1)) return rc;
  for (int r = 0; r < N; r++)
    if (int rc = round(r, kOrdRows, red + 1, count - 1)) return rc;
  return 0;
}

// ---- the sums of an update, one function per mode (include/nka_hip.h: nka_hip_set_sum_order; nka_hip_ext.h: table of supported
// combinations).  A failure up to and including the all-reduce leaves the update NOT done: only scratch (partials, red[]) has
// been written; f, the stored vectors, the lists and the host-side bookkeeping are untouched, so the same call may be repeated.
enum class SumsStage { HostDot, ReferenceChain, ReferenceOrder, Rounded, Blocked };
struct SumsResult {
  int rc = 0;              // != 0: the update is not done
  int mode_bits = 0;       // kSolvePrenorm: the sums were formed on the normalised pair, the scalar step takes them as they are
  bool solved = false;     // the scalar step has run already (user dot product: it is interleaved with the dp calls)
  bool gather = false;     // the scalar step starts by gathering the sums from the peer-to-peer mailboxes
};
static int hook_failed(int rc) { return rc < 0 ? rc : fail(NKA_HIP_ECOMM, "allreduce hook failed"); }

static SumsStage pick_sums_stage(const nka_hip_state *a) {
  if (a->host_dot) return SumsStage::HostDot;                                                   // the user's dp overrides everything
  if (a->sum_order == NKA_HIP_SUMS_REFERENCE_ORDER && a->allreduce) return SumsStage::ReferenceChain;
  if (ordered_sums(a)) return SumsStage::ReferenceOrder;               // asked for, or free (one rank, n <= 64)
  return a->sum_order == NKA_HIP_SUMS_BLOCKED ? SumsStage::Blocked : SumsStage::Rounded;        // AUTO resolves to Rounded (round 6)
}

// the user's dot product on host copies, in the reference's order, interleaved with the scalar step (nka_hip_set_host_dot)
static SumsResult sums_host_dot(nka_hip_t a, const double *f, int mode) {
  RoctxRange range("nka:host dot products + scalar step");
  SumsResult r;
  r.rc = host_dot_update_scalars(a, f, mode);
  r.solved = true;
  return r;
}

// reference order, sharded: the slices of the global vector walked rank after rank (ordered_chain): 2N exchanges
static SumsResult sums_reference_chain(nka_hip_t a, const double *f, int mode, int older_ub) {
  RoctxRange range("nka:PA dots in the reference's order, rank after rank");
  SumsResult r;
  if (a->pending || older_ub > 0) r.rc = ordered_chain(a, f, mode, older_ub);
  r.mode_bits = kSolvePrenorm;
  return r;
}

// reference order, one rank: every sum as the reference forms it -- the update returns the reference's bits
static SumsResult sums_reference_order(nka_hip_t a, const double *f, int mode, int older_ub) {
  RoctxRange range("nka:PA dots in the reference's order");
  SumsResult r;
  r.mode_bits = kSolvePrenorm;
  if (!(a->pending || older_ub > 0)) return r;
  hipStream_t s = a->stream;
  auto hip = [&](hipError_t e) { if (e != hipSuccess && !r.rc) r.rc = fail(NKA_HIP_EHIP, std::string("accel_update: ") + hipGetErrorString(e)); return e == hipSuccess; };
  const int rows = 2 + older_ub;                               // (older_ub bounds the device's count from above)
  if (chain_per_sum(a, rows)) {
    // long vectors: one workgroup per sum; the norm alone first (one chain: every other sum of the update would wait
    // for it on idle compute units if it shared a launch with sums twice as long), then everything else side by side
    if (!hip(hipMemsetAsync(a->ctl.red(), 0, sizeof(double) * (size_t)a->ctl.red_count(), s))) return r;
    if (chain_many_ready(a, 1 + 2 * a->mvec)) {
      // the longest vectors: every block of every sum summarised by the whole device, one wavefront per sum applies
      if (a->pending && (r.rc = chain_many_stage(a, f, mode & kSolveRcp, (int)kChainNorm, 0, older_ub, 1))) return r;
      if ((r.rc = chain_many_stage(a, f, mode & kSolveRcp, (int)kChainRows, 1, older_ub, 1 + 2 * older_ub))) return r;
    } else {
      if (a->pending)
        hipLaunchKernelGGL(k_chain_sums, dim3(1), dim3(kChainThreads), kChainLdsBytes, s, a->ctl, a->vs, f, mode & kSolveRcp,
                           (int)kChainNorm, 0, older_ub, a->chain_walk, (const double *)nullptr);
      hipLaunchKernelGGL(k_chain_sums, dim3(1 + 2 * older_ub), dim3(kChainThreads), kChainLdsBytes, s, a->ctl, a->vs, f, mode & kSolveRcp,
                         (int)kChainRows, 1, older_ub, a->chain_walk, (const double *)nullptr);
    }
  } else {
    hipLaunchKernelGGL(k_dots_ordered, dim3(1), dim3(kOrdThreads), ord_lds_bytes(rows), s, a->ctl, a->vs, f, mode & kSolveRcp,
                       ord_chunk(rows), (int)kOrdAll, 0);
  }
  hip(hipGetLastError());
  return r;
}

// THE DEFAULT since round 6 (NKA_HIP_SUMS_AUTO beyond 64 elements or sharded, and NKA_HIP_SUMS_BLOCKED_ROUNDED): on the same
// soak sequences the raw-sum Gram row ended beyond the rule's factor in 7 records of more than 512 elements, this form in 1
// (profiles/r06/soak_paired.txt).  The fast passes with the Gram row AS THE REFERENCE DEFINES IT: first the norm in a pass of
// its own (two streams: +2 of 49 words), then PA on the ROUNDED w1' = fl(d/s) -- the vector PB stores -- so that <w1',w_k> and
// <f,w1'> are inner products of stored vectors (F08:283-290, 371), summed in blocks with fma.  What is left of the device's
// deviations is the summation order and the fma, both CLOSER to the exact sums than the reference's sequential ones.  Sharded:
// two exchanges per update (the norm, then the rows) through the installed hook.
static SumsResult sums_rounded(nka_hip_t a, const double *f, int vec, int mode, int older_ub) {
  SumsResult r;
  if (!(a->pending || older_ub > 0)) return r;             // (the first update after init / restart: nothing to sum)
  RoctxRange range("nka:norm pass + PA on the rounded w1'");
  hipStream_t s = a->stream;
  auto hip = [&](hipError_t e) { if (e != hipSuccess && !r.rc) r.rc = fail(NKA_HIP_EHIP, std::string("accel_update: ") + hipGetErrorString(e)); return e == hipSuccess; };
  if (a->pending) {
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)a->num_cu, std::max<int64_t>(a->n / (kBlock * 2), 1)));
    hipLaunchKernelGGL(k_norm_diff, dim3(g), dim3(kBlock), 0, s, a->ctl, a->vs, f, a->partials);
    hipLaunchKernelGGL(k_norm_fin, dim3(1), dim3(64), 0, s, a->ctl, a->partials, g);
    if (!hip(hipGetLastError())) return r;
    if (a->allreduce)
      if (int rc = a->allreduce(a->allreduce_ctx, a->ctl.red(), 1, s)) { r.rc = hook_failed(rc); return r; }
  }
  a->pa_normed = (mode & kSolveRcp) ? 3 : 1;
  enqueue_pa(a, f, vec, older_ub);
  a->pa_normed = 0;
  if (!hip(hipGetLastError())) return r;
  if (a->allreduce)
    if (int rc = a->allreduce(a->allreduce_ctx, a->ctl.red() + 1, a->ctl.red_count() - 1, s)) { r.rc = hook_failed(rc); return r; }
  r.mode_bits = kSolvePrenorm;
  return r;
}

// NKA_HIP_SUMS_BLOCKED, the opt-in single-pass fast mode: ONE pure-read pass forms every sum (F08:266-267, 286-290, 371), the
// Gram row of the normalised difference is taken from raw sums in the scalar step; ONE exchange when sharded
static SumsResult sums_blocked(nka_hip_t a, const double *f, int vec, int older_ub) {
  SumsResult r;
  if (!(a->pending || older_ub > 0)) return r;
  RoctxRange range("nka:PA dots + all-reduce");
  // peer-to-peer exchange: the final sums go straight into every rank's mailbox and the scalar step gathers them -- no
  // kernel in between (nka_kernels.hpp: struct P2P)
  r.gather = a->allreduce == p2p_allreduce && a->p2p.base != nullptr;
  a->p2p_fused = r.gather;
  enqueue_pa(a, f, vec, older_ub);
  a->p2p_fused = false;
  if (hipError_t e = hipGetLastError(); e != hipSuccess) {
    r.rc = fail(NKA_HIP_EHIP, std::string("accel_update: ") + hipGetErrorString(e));
    return r;
  }
  // the ONE exchange of a sharded update: sum d^2, <f,d> and both Gram rows
  if (a->allreduce && !r.gather)
    if (int rc = a->allreduce(a->allreduce_ctx, a->ctl.red(), a->ctl.red_count(), a->stream)) r.rc = hook_failed(rc);
  return r;
}

static int update_impl(nka_hip_t a, double *f, long long swap_w, long long swap_v);
static int p2p_status_after_sync(nka_hip_t a);

int nka_hip_accel_update(nka_hip_t a, double *f) { return update_impl(a, f, kNoBuffer, kNoBuffer); }

// One update.  swap_w / swap_v != kNoBuffer (offsets from vs.w): out of place (nka_hip_accel_update_swap) -- f (== swap_w)
// becomes the w buffer of the new pair and swap_v its v buffer; PB then stores neither w_new nor f.
static int update_impl(nka_hip_t a, double *f, long long swap_w, long long swap_v) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (!f && a->n > 0) return fail(NKA_HIP_EINVAL, "accel_update: f is NULL");
  HIP_TRY(hipSetDevice(a->device));
  if (int rc = nka_detail::check_device_span(f, a->n, "accel_update: f")) return rc;   // F08:258 size(f) == vlen
  if (swap_w == kNoBuffer && f && f == a->last_acc)      // in place on the buffer the last out-of-place update lent for READING
    return fail(NKA_HIP_EINVAL, "accel_update: that buffer is the accelerated f lent by the previous out-of-place update (read only: "
                                "it is the stored v of the pending pair); copy it, or go on with nka_hip_accel_update_swap");
  if (a->poisoned)
    return fail(NKA_HIP_ESTATE, "accel_update: an earlier update of this handle failed after its scalar step had been enqueued "
                                "(a HIP error): the lists on the device are ahead of the stored vectors.  Destroy the handle");
  if (a->swapped && f && held_by_library(a, f))
    return fail(NKA_HIP_EINVAL, "accel_update: that buffer is held by the library (a stored vector, or a buffer handed over to "
                                "nka_hip_accel_update_swap earlier)");
  if (a->needs_comm)
    return fail(NKA_HIP_ECOMM, "accel_update: this accelerator is a copy of a sharded one and has no all-reduce yet: call "
                               "nka_hip_comm_init_rank or nka_hip_set_allreduce on it first (nka_hip_clone)");
  const bool chain = a->sum_order == NKA_HIP_SUMS_REFERENCE_ORDER && a->allreduce && !a->host_dot;
  if (chain && (a->shard_n < 1 || a->shard_rank < 0))
    return fail(NKA_HIP_ESTATE, "accel_update: reference-order sums on a sharded accelerator continue the running sums from rank "
                                "to rank: tell the handle where its slice lies in the global vector first (nka_hip_set_shard; "
                                "nka_hip_comm_init_rank does it for the built-in communicator)");
  if (chain && a->mvec > kOrdMaxMvec)
    return fail(NKA_HIP_EINVAL, "accel_update: reference-order sums are offered up to mvec = " + std::to_string(kOrdMaxMvec));
  if (a->debug && nka_hip_defined(a) != 1)                                                // F08:257 ASSERT(defined(this))
    return fail(NKA_HIP_ESTATE, "accel_update: the device state fails the defined() invariants");
  hipStream_t s = a->stream;
  const bool aligned = (reinterpret_cast<uintptr_t>(f) % 16) == 0;
  const int vec = aligned ? 2 : 1;
  int mode = (a->flavor == NKA_HIP_FLAVOR_F08_VECTOR) ? kSolveRcp : 0;
  {
    // Is this launch being captured into a graph?  (A handle that once was stops asking -- unless it carries a host callback.)
    const bool user_hook = a->allreduce && a->allreduce != rccl_allreduce && a->allreduce != p2p_allreduce;
    if (!a->captured || user_hook || a->host_dot) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();
      else if (cs != hipStreamCaptureStatusNone) {
        // NOT SUPPORTED (nka_hip_ext.h, table of combinations): a capture of an update whose sums pass through the HOST -- a
        // caller's all-reduce hook or dot product is a host callback that a replay would not call again: silently wrong
        // results.  Refused before anything of the handle has changed.
        if (user_hook || a->host_dot)
          return fail(NKA_HIP_EINVAL, "accel_update: cannot be captured into a graph with a caller's all-reduce hook or dot product "
                                      "installed (host callbacks: a replay would not call them); the built-in RCCL hook and the "
                                      "peer-to-peer exchange can be captured");
        // a captured update is replayed with the widths, the update number and the kernel ARGUMENTS of the capture: the list
        // word can neither describe the replays nor tighten them (capture_safe() asks for the full width anyway), and the
        // scalar step must find the slot -> buffer tables in memory (enqueue_solve)
        a->captured = a->word_off = true;
      }
    }
  }
  if (int rc = record(a, 0)) return rc;
  a->list_ub = list_bound_now(a);
  const int older_ub = a->pending ? std::max(a->list_ub - 1, 0) : a->list_ub;

  // ---- the inner products (F08:266-267, 286-290, 371): ONE function per way of forming them (round 6: update_impl used to
  // branch over all of them in line); each leaves the reduced sums in red[] -- or, with a user dot product, the whole scalar
  // step done -- and says what the scalar step must know.  Unaligned f (not 16-B aligned) takes scalar loads with the narrow
  // unroll.  A failure up to and including the all-reduce leaves the update NOT done (see above the stage functions).
  SumsResult sums{};
  switch (pick_sums_stage(a)) {
    case SumsStage::HostDot:        sums = sums_host_dot(a, f, mode); break;
    case SumsStage::ReferenceChain: sums = sums_reference_chain(a, f, mode, older_ub); break;
    case SumsStage::ReferenceOrder: sums = sums_reference_order(a, f, mode, older_ub); break;
    case SumsStage::Rounded:        sums = sums_rounded(a, f, vec, mode, older_ub); break;
    case SumsStage::Blocked:        sums = sums_blocked(a, f, vec, older_ub); break;
  }
  if (sums.rc) return sums.rc;
  mode |= sums.mode_bits;
  const bool solved = sums.solved, gather = sums.gather;
  if (int rc = record(a, 1)) return rc;

  // ---- scalar part on one wavefront (F08:267-275, 295-358, 366-392, 406-417) ----
  RoctxRange range_tail("nka:solve + PB combine");
  if (!solved)
    if (int rc = enqueue_solve(a, mode, swap_w, swap_v, gather)) return rc;
  // From here on the scalar step is in the stream: the lists, the factor and (out of place) the tables move on whatever
  // happens next.  What can still fail is a HIP call (an event record, a launch); the handle is then beyond repair.
  struct Poison {
    nka_hip_state *a;
    bool armed = true;
    ~Poison() { if (armed) a->poisoned = true; }
  } poison{a};
#ifdef NKA_DIAGNOSTIC
  if (a->fail_after_solve) {
    a->fail_after_solve = 0;
    return fail(NKA_HIP_EHIP, "accel_update: injected failure behind the scalar step (diagnostic switch fail_after_solve)");
  }
#endif
  if (int rc = record(a, 2)) return rc;

  const int comb_ub = a->pending ? std::min(a->list_ub, (int)a->mvec) : a->list_ub;
  a->ctl.seq = (unsigned long long)(a->seq + 1);       // PB publishes (this number, list length at exit)
  {
    Ctl &c = a->ctl;
    unsigned long long *const hw = c.hw;
    if (a->word_off) c.hw = nullptr;
    a->pb_flags = (swap_w != kNoBuffer ? (kPbNoStoreW | kPbNoStoreF) : 0) | (a->pb_reverse ? kPbReverse : 0);
    const int rc = enqueue_pb(a, f, vec, comb_ub);
    a->pb_flags = 0;
    c.hw = hw;
    if (rc) return rc;
  }
  a->seq++;
  if (int rc = record(a, 3)) return rc;

  if (a->timing_this) a->timing_count++;
  a->list_ub = comb_ub + 1;
  a->pending = true;
  if (swap_w == kNoBuffer) a->last_acc = nullptr;         // (the loan of the previous out-of-place update has ended)
  poison.armed = false;
  return 0;
}

int nka_hip_accel_update_host(nka_hip_t a, double *f_host) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (!f_host && a->n > 0) return fail(NKA_HIP_EINVAL, "accel_update_host: f is NULL");
  HIP_TRY(hipSetDevice(a->device));
  if (!a->f_stage) HIP_TRY(hipMalloc((void **)&a->f_stage, sizeof(double) * (size_t)std::max<int64_t>(a->n, 1)));
  HIP_TRY(hipMemcpyAsync(a->f_stage, f_host, sizeof(double) * (size_t)a->n, hipMemcpyHostToDevice, a->stream));
  if (int rc = nka_hip_accel_update(a, a->f_stage)) return rc;
  HIP_TRY(hipMemcpyAsync(f_host, a->f_stage, sizeof(double) * (size_t)a->n, hipMemcpyDeviceToHost, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  return p2p_status_after_sync(a);
}

// The buffers the last out-of-place update displaced become the spares of the next one.  They are known on the device
// (PC_OLD_W / PC_OLD_V, written by the scalar step) and reach the host with PB's record (words 1..3 of the list word's
// cache line) -- without a synchronisation if the caller has synchronised since (every solver does, once per
// iteration); otherwise this waits for the stream.
static int collect_spares(nka_hip_t a) {
  if (a->swap_pending) {
    auto fresh = [&]() {
      return a->list_word && !a->word_off &&
             (int64_t)__atomic_load_n(a->list_word + 3, __ATOMIC_ACQUIRE) == a->swap_seq;
    };
    if (!fresh()) {
      HIP_TRY(hipStreamSynchronize(a->stream));
      if (int rc = p2p_status_after_sync(a)) return rc;
    }
    long long ow = kNoBuffer, ov = kNoBuffer;
    if (fresh()) {
      ow = (long long)a->list_word[1];
      ov = (long long)a->list_word[2];
    } else {                                   // no record (the word is switched off): read the address block itself
      long long hdr[PC_HEADER] = {};
      HIP_TRY(hipMemcpy(hdr, a->ctl.pc, sizeof hdr, hipMemcpyDeviceToHost));
      ow = hdr[PC_OLD_W];
      ov = hdr[PC_OLD_V];
    }
    if (ow == kNoBuffer || ov == kNoBuffer)
      return fail(NKA_HIP_ESTATE, "accel_update_swap: the displaced buffers of the previous update are missing");
    a->spare_w = buffer_at(a, ow);
    a->spare_v = buffer_at(a, ov);
    a->swap_pending = false;
  }
  if (!a->spare_w || !a->spare_v) {            // first out-of-place update of this handle: two more buffers
    const size_t bytes = sizeof(double) * (size_t)a->vs.stride;
    for (double **p : {&a->spare_w, &a->spare_v}) {
      if (*p) continue;
      void *q = nullptr;
      hipError_t e = hipMalloc(&q, bytes);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(NKA_HIP_ENOMEM, std::string("accel_update_swap: hipMalloc of a spare buffer: ") + hipGetErrorString(e));
      }
      a->extra_allocs.push_back(q);
      a->book.taken.insert(static_cast<const double *>(q));
      *p = static_cast<double *>(q);
    }
  }
  return 0;
}

int nka_hip_accel_update_swap(nka_hip_t a, double **f_io, const double **f_acc) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (!f_io || !f_acc || (!*f_io && a->n > 0)) return fail(NKA_HIP_EINVAL, "accel_update_swap: null argument");
  if (a->host_dot) return fail(NKA_HIP_ESTATE, "accel_update_swap: not with a user dot product on host copies (nka_hip_set_host_dot)");
  if (reinterpret_cast<uintptr_t>(*f_io) % 16 != 0) return fail(NKA_HIP_EINVAL, "accel_update_swap: the buffer must be 16-byte aligned");
  HIP_TRY(hipSetDevice(a->device));
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(a->stream, &cs) != hipSuccess) (void)hipGetLastError();
  else if (cs != hipStreamCaptureStatusNone)
    return fail(NKA_HIP_ESTATE, "accel_update_swap: cannot be captured into a graph (the host chooses buffers per call)");
  if (int rc = collect_spares(a)) return rc;
  double *const in = *f_io, *const give_w = a->spare_w, *const vnew = a->spare_v;
  if (in && (in == give_w || in == vnew)) return fail(NKA_HIP_EINVAL, "accel_update_swap: that buffer is the library's own spare");
  if (in && in == a->last_acc)          // (an empty slice, vlen 0, hands over a null buffer every time)
    return fail(NKA_HIP_EINVAL, "accel_update_swap: that buffer is the accelerated f lent by the previous update (read only: it "
                                "is the stored v of the pending pair)");
  if (in && held_by_library(a, in))
    return fail(NKA_HIP_EINVAL, "accel_update_swap: that buffer is already held by the library (a stored vector, or a buffer "
                                "handed over earlier): two slots would share it");
  // (set BEFORE the update: it only makes the scalar step load the tables instead of computing them -- right from the
  //  moment its launch may have rewritten them, whatever fails behind it.  An update that fails before its scalar step
  //  is not done and the spares stay; one that fails behind it poisons the handle, update_impl)
  a->swapped = true;
  if (int rc = update_impl(a, in, buffer_offset(a, in), buffer_offset(a, vnew))) return rc;
  if (in) {
    a->book.lent.erase(in);
    const int64_t block = a->vs.stride * (int64_t)(a->mvec + 1);
    const bool inside = (in >= a->vs.w && in < a->vs.w + block) || (in >= a->vs.v && in < a->vs.v + block);
    if (!inside) a->book.taken.insert(in);
  }
  if (give_w) {
    a->book.lent.insert(give_w);
    // what is handed out is no longer "taken over" -- otherwise the set only grows (a caller that hands in a fresh
    // buffer per iteration), and a later foreign buffer overlapping a FREED one's old range would be refused.  The library's
    // own extra allocations stay (freed at destroy; they come back through `lent`).
    if (std::find(a->extra_allocs.begin(), a->extra_allocs.end(), (void *)give_w) == a->extra_allocs.end()) a->book.taken.erase(give_w);
  }
  a->swap_pending = true;
  a->swap_seq = a->seq;
  a->spare_w = a->spare_v = nullptr;
  *f_io = give_w;
  *f_acc = vnew;
  a->last_acc = vnew;
  return 0;
}

// ---- queries --------------------------------------------------------------------

// A gather of the peer-to-peer exchange that gave up stored NaNs and raised the status word; the scalar step and PB then ran
// on the NaNs.  Checked wherever the host has just synchronised the stream anyway (a caller that loops on
// accel_update_host, the out-of-place entry or get_timing alone must not go on with NaNs and NKA_HIP_OK); the handle is
// poisoned: every later update returns NKA_HIP_ESTATE.
static int p2p_status_after_sync(nka_hip_t a) {
  if (!a->p2p.base) return 0;
  int st = 0;
  HIP_TRY(hipMemcpy(&st, a->p2p.status, sizeof st, hipMemcpyDeviceToHost));
  if (st == 0) return 0;
  a->poisoned = true;
  return fail(NKA_HIP_ECOMM, "peer-to-peer exchange: a rank's sums did not arrive in time (nka_hip_p2p_attach); "
                             "the state of this handle is not usable any more");
}

static int fetch_state(nka_hip_t a, std::vector<int32_t> &ic, std::vector<double> &dc) {
  HIP_TRY(hipSetDevice(a->device));
  ic.resize(a->ctl.ic_count());
  dc.resize(a->ctl.dc_count());
  HIP_TRY(hipMemcpyAsync(ic.data(), a->ctl.ic, sizeof(int32_t) * ic.size(), hipMemcpyDeviceToHost, a->stream));
  HIP_TRY(hipMemcpyAsync(dc.data(), a->ctl.dc, sizeof(double) * dc.size(), hipMemcpyDeviceToHost, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  return p2p_status_after_sync(a);
}

int nka_hip_num_vec(nka_hip_t a) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  std::vector<int32_t> ic;
  std::vector<double> dc;
  if (int rc = fetch_state(a, ic, dc)) return rc;
  const int32_t *next = ic.data() + IC_HEADER;
  int n = 0;
  for (int k = ic[IC_FIRST]; k != 0 && n <= a->mvec + 1; k = next[k]) n++;  // F08:224-229
  return ic[IC_PENDING] ? n - 1 : n;                                          // F08:230
}

int nka_hip_max_vec(nka_hip_t a) { return a ? a->mvec : fail(NKA_HIP_EINVAL, "null handle"); }
int nka_hip_flavor(nka_hip_t a) { return a ? a->flavor : fail(NKA_HIP_EINVAL, "null handle"); }
int64_t nka_hip_vec_len(nka_hip_t a) { return a ? a->n : (int64_t)fail(NKA_HIP_EINVAL, "null handle"); }
double nka_hip_vec_tol(nka_hip_t a) {
  if (a) return a->vtol;
  fail(NKA_HIP_EINVAL, "null handle");
  return -1.0;   // never a valid tolerance (vtol > 0)
}

int nka_hip_get_state(nka_hip_t a, int32_t *subspace, int32_t *pending, int32_t *first, int32_t *last,
                      int32_t *free_, int32_t *next, int32_t *prev, double *h, double *c) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  std::vector<int32_t> ic;
  std::vector<double> dc;
  if (int rc = fetch_state(a, ic, dc)) return rc;
  const int m1 = a->mvec + 1;
  if (subspace) *subspace = ic[IC_SUBSPACE];
  if (pending) *pending = ic[IC_PENDING];
  if (first) *first = ic[IC_FIRST];
  if (last) *last = ic[IC_LAST];
  if (free_) *free_ = ic[IC_FREE];
  const int32_t *nx = ic.data() + IC_HEADER, *pv = nx + (m1 + 1);
  const double *hh = dc.data() + DC_HEADER, *cc = hh + (m1 + 1) * (m1 + 1);
  for (int k = 1; k <= m1; k++) {
    if (next) next[k - 1] = nx[k];
    if (prev) prev[k - 1] = pv[k];
    if (c) c[k - 1] = cc[k];
  }
  if (h)
    for (int j = 1; j <= m1; j++)
      for (int i = 1; i <= m1; i++) h[(i - 1) + (size_t)(j - 1) * m1] = hh[i * (m1 + 1) + j];
  return 0;
}

#ifdef NKA_DIAGNOSTIC
// Diagnostic builds (-DNKA_SOLVE_STAMPS): the s_memtime stamps of the last scalar step.
int nka_hip_get_stamps(nka_hip_t a, double *out16) {
  if (!a || !out16) return fail(NKA_HIP_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(a->device));
  HIP_TRY(hipMemcpyAsync(out16, a->ctl.stamps(), sizeof(double) * kStamps, hipMemcpyDeviceToHost, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  return 0;
}
#endif  // NKA_DIAGNOSTIC

int nka_hip_get_reductions(nka_hip_t a, double *red_out) {
  if (!a || !red_out) return fail(NKA_HIP_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(a->device));
  HIP_TRY(hipMemcpyAsync(red_out, a->ctl.red(), sizeof(double) * a->ctl.red_count(), hipMemcpyDeviceToHost, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  return 0;
}

// F08:460-524 on a snapshot of the device state
int nka_hip_defined(nka_hip_t a) {
  if (!a) return 0;
  std::vector<int32_t> ic;
  std::vector<double> dc;
  if (fetch_state(a, ic, dc)) return 0;
  const int n = a->mvec + 1;
  if (a->mvec < 1 || !a->vs.v || !a->vs.w) return 0;
  if (!(dc[DC_VTOL] > 0.0)) return 0;
  const int32_t *next = ic.data() + IC_HEADER, *prev = next + (n + 1);
  const int first = ic[IC_FIRST], last = ic[IC_LAST], fr = ic[IC_FREE];
  for (int k = 1; k <= n; k++)
    if (next[k] < 0 || next[k] > n) return 0;
  if (first < 0 || first > n || fr < 0 || fr > n) return 0;
  std::vector<char> tag(n + 1, 0);
  if (first == 0) {
    if (last != 0) return 0;
  } else {
    int k = first;
    if (prev[k] != 0) return 0;
    tag[k] = 1;
    while (next[k] != 0) {
      if (prev[next[k]] != k) return 0;
      k = next[k];
      if (tag[k]) return 0;
      tag[k] = 1;
    }
    if (last != k) return 0;
  }
  for (int k = fr; k != 0; k = next[k]) {
    if (tag[k]) return 0;
    tag[k] = 1;
  }
  for (int k = 1; k <= n; k++)
    if (!tag[k]) return 0;
  return 1;
}

static int get_slot(nka_hip_t a, bool v, int32_t slot, double *host_out) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (slot < 1 || slot > a->mvec + 1) return fail(NKA_HIP_EINVAL, "slot out of range");
  HIP_TRY(hipSetDevice(a->device));
  std::vector<long long> pc;
  Ctl t{};
  if (a->swapped)
    if (int rc = fetch_tables(a, pc, t)) return rc;
  HIP_TRY(hipMemcpyAsync(host_out, slot_buffer(a, a->swapped ? &t : nullptr, v, slot), sizeof(double) * (size_t)a->n,
                         hipMemcpyDeviceToHost, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  return 0;
}
int nka_hip_get_w(nka_hip_t a, int32_t slot, double *host_out) { return get_slot(a, false, slot, host_out); }
int nka_hip_get_v(nka_hip_t a, int32_t slot, double *host_out) { return get_slot(a, true, slot, host_out); }

// ---- distribution hook ---------------------------------------------------------

int nka_hip_set_allreduce(nka_hip_t a, nka_hip_allreduce_fn fn, void *ctx) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  a->allreduce = fn;
  a->allreduce_ctx = ctx;
  a->needs_comm = false;       // an explicit choice, NULL (single rank) included
  return 0;
}

// ---- peer-to-peer exchange: set-up (collective, like the RCCL communicator) -----------------------------------------
int nka_hip_p2p_export(nka_hip_t a, int32_t nranks, void *handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
  if (!a || !handle64) return fail(NKA_HIP_EINVAL, "null argument");
  if (nranks < 1 || nranks > 64) return fail(NKA_HIP_EINVAL, "p2p_export: 1..64 ranks (one node)");
  HIP_TRY(hipSetDevice(a->device));
  if (int rc = nka_hip_p2p_detach(a)) return rc;
  const int cap = std::max(a->ctl.red_count(), 64);
  const size_t bytes = p2p_mailbox_bytes(nranks, cap);
  void *m = nullptr;
  // fine-grained device memory: peers' stores land coherently (RCCL allocates its own flags the same way)
  hipError_t e = hipExtMallocWithFlags(&m, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(NKA_HIP_ECOMM, std::string("p2p_export: fine-grained allocation: ") + hipGetErrorString(e));
  }
  hipIpcMemHandle_t h;
  if ((e = hipMemset(m, 0, bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess ||
      (e = hipIpcGetMemHandle(&h, m)) != hipSuccess) {
    (void)hipGetLastError();
    hipFree(m);
    return fail(NKA_HIP_ECOMM, std::string("p2p_export: hipIpcGetMemHandle: ") + hipGetErrorString(e) +
                               " (this pool needs HSA_ENABLE_IPC_MODE_LEGACY=0)");
  }
  a->p2p_mail = m;
  a->p2p_ranks = nranks;
  memcpy(handle64, &h, sizeof h);
  return 0;
}

// `local` != nullptr: the peers live in THIS process (several handles, one per slice, driven by threads: hipIpc does not map a
// handle into the process that exported it) -- their mailboxes are given as plain device addresses in rank order.
static int p2p_attach_impl(nka_hip_t a, const void *handles, void *const *local, int32_t nranks, int32_t rank) {
  if (!a || (!handles && !local)) return fail(NKA_HIP_EINVAL, "null argument");
  if (!a->p2p_mail || nranks != a->p2p_ranks) return fail(NKA_HIP_ESTATE, "p2p_attach: call nka_hip_p2p_export(nranks) first");
  if (a->p2p.base || a->p2p_dev || !a->p2p_opened.empty())
    return fail(NKA_HIP_ESTATE, "p2p_attach: already attached (nka_hip_p2p_detach, then export and attach again, collectively)");
  if (rank < 0 || rank >= nranks) return fail(NKA_HIP_EINVAL, "p2p_attach: bad rank");
  if (local && local[rank] != a->p2p_mail) return fail(NKA_HIP_EINVAL, "p2p_attach_local: entry `rank` is not this handle's own mailbox");
  HIP_TRY(hipSetDevice(a->device));
  if (local) {
    // Slices that share a DEVICE wait for one another on that device: each of their streams must own a hardware queue, or a
    // wait sits in front of the kernel it waits for until the timeout (measured in round 6: profiles/r06/configs3_inproc.txt).
    // The runtime maps streams onto GPU_MAX_HW_QUEUES queues (default 4), read when HIP starts: refuse what cannot work.
    int here = 0;
    for (int q = 0; q < nranks; q++) {
      hipPointerAttribute_t at{};
      if (local[q] && hipPointerGetAttributes(&at, local[q]) == hipSuccess) here += at.device == a->device;
      else (void)hipGetLastError();
    }
    const int queues = env_int("GPU_MAX_HW_QUEUES", 4);
    if (here > queues)
      return fail(NKA_HIP_ESTATE, "p2p_attach_local: " + std::to_string(here) + " slices share device " + std::to_string(a->device) +
                                  " but the HIP runtime maps streams onto " + std::to_string(queues) + " hardware queues: set "
                                  "GPU_MAX_HW_QUEUES >= " + std::to_string(here) + " in the environment before HIP starts");
  }
  const int cap = std::max(a->ctl.red_count(), 64);
  std::vector<long long> off((size_t)nranks, 0);
  for (int q = 0; q < nranks; q++) {
    if (q == rank) continue;
    void *p = nullptr;
    if (local) {
      p = local[q];
      if (!p) return fail(NKA_HIP_EINVAL, "p2p_attach_local: null mailbox");
    } else {
      hipIpcMemHandle_t h;
      memcpy(&h, static_cast<const char *>(handles) + (size_t)q * sizeof h, sizeof h);
      hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        for (void *o : a->p2p_opened) hipIpcCloseMemHandle(o);
        a->p2p_opened.clear();
        return fail(NKA_HIP_ECOMM, std::string("p2p_attach: hipIpcOpenMemHandle(rank ") + std::to_string(q) + "): " + hipGetErrorString(e));
      }
      a->p2p_opened.push_back(p);
    }
    off[(size_t)q] = (long long)(reinterpret_cast<intptr_t>(p) - reinterpret_cast<intptr_t>(a->p2p_mail));
  }
  // offsets table | exchange counter | status word
  const size_t tbytes = sizeof(long long) * (size_t)nranks;
  hipError_t e = hipMalloc(&a->p2p_dev, tbytes + 16);
  unsigned long long one = 1;
  int zero = 0;
  if (e == hipSuccess) e = hipMemcpy(a->p2p_dev, off.data(), tbytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(static_cast<char *>(a->p2p_dev) + tbytes, &one, sizeof one, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(static_cast<char *>(a->p2p_dev) + tbytes + 8, &zero, sizeof zero, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    nka_hip_p2p_detach(a);
    return fail(NKA_HIP_EHIP, std::string("p2p_attach: ") + hipGetErrorString(e));
  }
  a->p2p.base = static_cast<char *>(a->p2p_mail);
  a->p2p.off = static_cast<const long long *>(a->p2p_dev);
  a->p2p.xseq = reinterpret_cast<unsigned long long *>(static_cast<char *>(a->p2p_dev) + tbytes);
  a->p2p.status = reinterpret_cast<int *>(static_cast<char *>(a->p2p_dev) + tbytes + 8);
  a->p2p.n = nranks;
  a->p2p.me = rank;
  a->p2p.cap = cap;
  a->p2p.timeout_ticks = (long long)env_int("NKA_HIP_P2P_TIMEOUT_MS", 10000) * 100000ll;      // wall_clock64: 100 MHz
  a->allreduce = p2p_allreduce;
  a->allreduce_ctx = a;
  a->needs_comm = false;
  a->shard_rank = rank;
  a->shard_n = nranks;
  return 0;
}

int nka_hip_p2p_attach(nka_hip_t a, const void *handles, int32_t nranks, int32_t rank) {
  return p2p_attach_impl(a, handles, nullptr, nranks, rank);
}
int nka_hip_p2p_attach_local(nka_hip_t a, void *const *mailboxes, int32_t nranks, int32_t rank) {
  return p2p_attach_impl(a, nullptr, mailboxes, nranks, rank);
}
int nka_hip_p2p_mailbox(nka_hip_t a, void **mailbox) {
  if (!a || !mailbox) return fail(NKA_HIP_EINVAL, "null argument");
  if (!a->p2p_mail) return fail(NKA_HIP_ESTATE, "p2p_mailbox: call nka_hip_p2p_export first");
  *mailbox = a->p2p_mail;
  return 0;
}

int nka_hip_p2p_detach(nka_hip_t a) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (!a->p2p_mail && !a->p2p_dev && a->p2p_opened.empty()) return 0;
  hipSetDevice(a->device);
  hipStreamSynchronize(a->stream);
  for (void *o : a->p2p_opened) hipIpcCloseMemHandle(o);
  a->p2p_opened.clear();
  hipFree(a->p2p_dev);
  hipFree(a->p2p_mail);
  a->p2p_dev = a->p2p_mail = nullptr;
  a->p2p = P2P{};
  a->p2p_ranks = 0;
  if (a->allreduce == p2p_allreduce) {
    a->allreduce = nullptr;
    a->allreduce_ctx = nullptr;
  }
  (void)hipGetLastError();
  return 0;
}

int nka_hip_set_shard(nka_hip_t a, int32_t rank, int32_t nranks) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(NKA_HIP_EINVAL, "set_shard: bad rank / nranks");
  a->shard_rank = rank;
  a->shard_n = nranks;
  return 0;
}

int nka_hip_set_host_dot(nka_hip_t a, nka_hip_host_dot_fn fn, void *ctx) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  a->host_dot = fn;
  a->host_dot_ctx = fn ? ctx : nullptr;
  return 0;
}

int nka_hip_comm_unique_id(void *id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  if (!id128) return fail(NKA_HIP_EINVAL, "id buffer is NULL");
  if (!rccl().ok()) return fail(NKA_HIP_ECOMM, rccl().err);
  ncclUniqueId id;
  ncclResult_t r = rccl().GetUniqueId(&id);
  if (r != ncclSuccess) return fail(NKA_HIP_ECOMM, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r));
  memcpy(id128, &id, sizeof id);
  return 0;
}

int nka_hip_comm_init_rank(nka_hip_t a, const void *id128, int32_t nranks, int32_t rank) {
  if (!a || !id128) return fail(NKA_HIP_EINVAL, "null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(NKA_HIP_EINVAL, "bad rank / nranks");
  if (!rccl().ok()) return fail(NKA_HIP_ECOMM, rccl().err);
  HIP_TRY(hipSetDevice(a->device));
  if (a->comm) {
    rccl().CommDestroy(a->comm);
    a->comm = nullptr;
    if (a->allreduce == rccl_allreduce) a->allreduce = nullptr;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclResult_t r = rccl().CommInitRank(&a->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    a->comm = nullptr;
    return fail(NKA_HIP_ECOMM, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
  }
  a->allreduce = rccl_allreduce;
  a->allreduce_ctx = a;
  a->needs_comm = false;
  a->shard_rank = rank;        // slices in rank order (nka_amd/dist.py: slice_bounds), as every sharded caller lays them out
  a->shard_n = nranks;
  return 0;
}

int nka_hip_comm_destroy(nka_hip_t a) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (a->comm) {
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(hipStreamSynchronize(a->stream));
    rccl().CommDestroy(a->comm);
    a->comm = nullptr;
  }
  if (a->allreduce == rccl_allreduce) {
    a->allreduce = nullptr;
    a->allreduce_ctx = nullptr;
  }
  return 0;
}

int nka_hip_comm_info(nka_hip_t a, int32_t *nranks, int32_t *rank) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (nranks) *nranks = 0;
  if (rank) *rank = -1;
  if (!a->comm) return 0;                       // no built-in communicator on this handle
  int n = 0, r = -1;
  ncclResult_t e = rccl().CommCount(a->comm, &n);
  if (e == ncclSuccess) e = rccl().CommUserRank(a->comm, &r);
  if (e != ncclSuccess) return fail(NKA_HIP_ECOMM, std::string("ncclCommCount/UserRank: ") + rccl().GetErrorString(e));
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  return 0;
}

int nka_hip_comm_library(char *path, int32_t len) {
  if (!path || len <= 0) return fail(NKA_HIP_EINVAL, "bad buffer");
  if (!rccl().ok()) return fail(NKA_HIP_ECOMM, rccl().err);
  snprintf(path, (size_t)len, "%s", rccl().path.c_str());
  return 0;
}

int nka_hip_allreduce_now(nka_hip_t a, double *buf_dev, int32_t count) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (count < 0) return fail(NKA_HIP_EINVAL, "negative count");
  HIP_TRY(hipSetDevice(a->device));
  if (int rc = nka_detail::check_device_span(buf_dev, count, "allreduce_now: buf")) return rc;
  if (a->allreduce && count > 0)
    if (int rc = a->allreduce(a->allreduce_ctx, buf_dev, count, a->stream))
      return rc < 0 ? rc : fail(NKA_HIP_ECOMM, "allreduce hook failed");
  return 0;
}

// FNV-1a over the two control blocks as they stand on the device.
int nka_hip_state_digest(nka_hip_t a, uint64_t *digest) {
  if (!a || !digest) return fail(NKA_HIP_EINVAL, "null argument");
  std::vector<int32_t> ic;
  std::vector<double> dc;
  if (int rc = fetch_state(a, ic, dc)) return rc;
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h](const void *p, size_t nbytes) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < nbytes; i++) {
      h ^= b[i];
      h *= 1099511628211ull;
    }
  };
  mix(ic.data(), ic.size() * sizeof(int32_t));
  mix(dc.data(), dc.size() * sizeof(double));
  *digest = h;
  return 0;
}

// ---- instrumentation -------------------------------------------------------------

int nka_hip_set_timing(nka_hip_t a, int32_t capacity) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (capacity < 0 || capacity > 4096) return fail(NKA_HIP_EINVAL, "timing capacity out of range");
  HIP_TRY(hipSetDevice(a->device));
  HIP_TRY(hipStreamSynchronize(a->stream));
  for (auto &e : a->ev)
    if (e) hipEventDestroy(e);
  a->ev.assign((size_t)capacity * kTimingEvents, nullptr);
  for (auto &e : a->ev) HIP_TRY(hipEventCreate(&e));
  a->timing_cap = capacity;
  a->timing_count = 0;
  a->update_seq = 0;
  return 0;
}

int nka_hip_get_timing(nka_hip_t a, int32_t back, float ms[4]) {
  if (!a || !ms) return fail(NKA_HIP_EINVAL, "null argument");
  for (int i = 0; i < 4; i++) ms[i] = 0.f;
  if (a->timing_cap <= 0 || back < 0 || back >= a->timing_cap || back >= a->timing_count)
    return fail(NKA_HIP_ESTATE, "no such timed update recorded");
  HIP_TRY(hipSetDevice(a->device));
  HIP_TRY(hipStreamSynchronize(a->stream));
  if (int rc = p2p_status_after_sync(a)) return rc;
  const int slot = (int)((a->timing_count - 1 - back) % a->timing_cap);
  hipEvent_t *e = a->ev.data() + (size_t)slot * kTimingEvents;
  HIP_TRY(hipEventElapsedTime(&ms[0], e[0], e[1]));
  HIP_TRY(hipEventElapsedTime(&ms[1], e[1], e[2]));
  HIP_TRY(hipEventElapsedTime(&ms[2], e[2], e[3]));
  HIP_TRY(hipEventElapsedTime(&ms[3], e[0], e[3]));
  return 0;
}

int nka_hip_set_sum_order(nka_hip_t a, int32_t order) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (order != NKA_HIP_SUMS_AUTO && order != NKA_HIP_SUMS_REFERENCE_ORDER && order != NKA_HIP_SUMS_BLOCKED &&
      order != NKA_HIP_SUMS_BLOCKED_ROUNDED)
    return fail(NKA_HIP_EINVAL, "set_sum_order: NKA_HIP_SUMS_AUTO, _REFERENCE_ORDER, _BLOCKED or _BLOCKED_ROUNDED");
  if (order == NKA_HIP_SUMS_REFERENCE_ORDER && a->mvec > kOrdMaxMvec)
    return fail(NKA_HIP_EINVAL, "set_sum_order: reference-order sums are offered up to mvec = " + std::to_string(kOrdMaxMvec));
  a->sum_order = order;
  return 0;
}

int nka_hip_set_timing_stride(nka_hip_t a, int32_t stride) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (stride < 1 || stride > 1024) return fail(NKA_HIP_EINVAL, "timing stride: 1..1024");
  a->timing_stride = stride;
  a->update_seq = 0;
  return 0;
}

#ifdef NKA_DIAGNOSTIC      // ---- the builder's lab: only in libnka_hip_diag.so (include/nka_hip_diag.h) ----
int nka_hip_set_grid(nka_hip_t a, int32_t pa, int32_t pb) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  const int32_t v[2] = {pa, pb};
  for (int i = 0; i < 2; i++) {
    if (v[i] < 0 || v[i] * a->num_cu > kMaxGrid) return fail(NKA_HIP_EINVAL, "blocks per CU out of range");
    a->bpc[i] = v[i];   // 0 = automatic
  }
  return 0;
}

// start + x[0]*y[0] + x[1]*y[1] + ... as the per-sum reference-order kernel forms it (k_chain_sums on one workgroup),
// over ANY two device arrays: the test bench of chain_block_summary / _apply (walk = 1: every block element after element).
int nka_hip_debug_chain_sum(nka_hip_t a, const double *x, const double *y, int64_t n, double start, int32_t walk, double *sum,
                            float *ms) {
  if (!a || !sum || n < 0 || (n > 0 && (!x || !y))) return fail(NKA_HIP_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(a->device));
  Vecs vs = a->vs;
  vs.n = n;
  double *slot = a->ctl.red() + 2 + a->mvec;
  struct Events {          // (destroyed on EVERY exit)
    hipEvent_t e[2] = {nullptr, nullptr};
    ~Events() { for (hipEvent_t x : e) if (x) hipEventDestroy(x); }
  } ev;
  hipEvent_t &e0 = ev.e[0], &e1 = ev.e[1];
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipMemcpyAsync(slot, &start, sizeof(double), hipMemcpyHostToDevice, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  const bool many = (walk & 2) != 0;                    // (bit 1: through k_chain_blocks / _predict / _apply)
  if (many) {
    const int keep = a->chain_many;
    a->chain_many = 1;
    const bool ready = chain_many_ready(a, 1, n);
    a->chain_many = keep;
    if (!ready) return fail(NKA_HIP_EINVAL, "debug_chain_sum: no full block, or no memory for the block arrays");
  }
  HIP_TRY(hipEventRecord(e0, a->stream));
  if (many) {
    const int keepw = a->chain_walk;
    a->chain_walk = walk & 1;
    const int rc = chain_many_stage(a, x, 0, (int)kChainProbe, 1, 0, 1, y, n);
    a->chain_walk = keepw;
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(k_chain_sums, dim3(1), dim3(kChainThreads), kChainLdsBytes, a->stream, a->ctl, vs, x, 0, (int)kChainProbe, 1, 0, (int)(walk & 1), y);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e1, a->stream));
  HIP_TRY(hipMemcpyAsync(sum, slot, sizeof(double), hipMemcpyDeviceToHost, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  if (ms) HIP_TRY(hipEventElapsedTime(ms, e0, e1));
  return 0;
}

// Measurement aid: the PA launches of the NEXT update, `reps` times back to back on the
// handle's stream, timed with HIP events (mean ms per repetition).  PA only writes
// scratch (partials, red[]), so the state is unchanged.
int nka_hip_debug_time_pa(nka_hip_t a, const double *f, int32_t reps, float *ms_mean) {
  if (!a || !f || !ms_mean || reps < 1) return fail(NKA_HIP_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(a->device));
  if (int rc = nka_detail::check_device_span(f, a->n, "debug_time_pa: f")) return rc;
  const int vec = (reinterpret_cast<uintptr_t>(f) % 16) == 0 ? 2 : 1;
  const int older_ub = a->pending ? std::max(a->list_ub - 1, 0) : a->list_ub;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  enqueue_pa(a, f, vec, older_ub);   // warm
  HIP_TRY(hipEventRecord(e0, a->stream));
  for (int r = 0; r < reps; r++) enqueue_pa(a, f, vec, older_ub);
  HIP_TRY(hipEventRecord(e1, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *ms_mean = ms / reps;
  return 0;
}

int nka_hip_set_tuning(nka_hip_t a, const char *key, int32_t value) {
  if (!a || !key) return fail(NKA_HIP_EINVAL, "null argument");
  const std::string k(key);
  if (k == "pb_pipe") {
    if (value != -1 && value != 0 && !(value > 200 && value <= 204))
      return fail(NKA_HIP_EINVAL, "pb_pipe: -1 (auto), 0 (every load of a tile in flight), 201..204 (rolling window, 1..4 blocks per CU)");
    a->pb_pipe = value;
  } else if (k == "pa_pipe") {
    if (value != -1 && value != 0 && !(value > 200 && value <= 204))
      return fail(NKA_HIP_EINVAL, "pa_pipe: -1 (auto), 0 (every load of a tile in flight), 201..204 (rolling window, 1..4 blocks per CU)");
    a->pa_pipe = value;
  } else if (k == "pb_tile") {
    if (value != -1 && value != 1 && value != 2) return fail(NKA_HIP_EINVAL, "pb_tile: -1 (auto), 1, 2");
    a->pb_tile = value;
  } else if (k == "pb_tickets") {
    if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
      return fail(NKA_HIP_EINVAL, "pb_tickets: -1 (auto), 0 (static tile mapping), 1, 2, 4, 8 (ticket counters)");
    a->pb_tickets = value;
  } else if (k == "fail_after_solve") {   // 1: the NEXT update returns NKA_HIP_EHIP right behind its enqueued scalar step
    a->fail_after_solve = value != 0;
  } else if (k == "prime_pad") {      // -1 automatic = 1: list lengths 23 / 29 / 31 run the next width (one dead ring slot); 0: exact widths
    if (value < -1 || value > 1) return fail(NKA_HIP_EINVAL, "prime_pad: -1, 0, 1");
    a->prime_pad = value;
  } else if (k == "chain_many") {     // -1 automatic, 0: one compute unit per reference-order sum at every length, 1: many wherever blocks exist
    if (value < -1 || value > 1) return fail(NKA_HIP_EINVAL, "chain_many: -1, 0, 1");
    a->chain_many = value;
  } else if (k == "chain_walk") {     // 1: the per-sum reference-order kernel walks every block (no summaries): same bits, for A/B
    a->chain_walk = value != 0;
  } else if (k == "pb_reverse") {     // 1: the rolling-window PB walks its tiles in the reverse of PA's order (round-5 A/B)
    a->pb_reverse = value != 0;
  } else if (k == "serial_solve") {
    a->serial_solve = value != 0;
  } else if (k == "list_word") {       // 0: the host's own bound only (the behaviour before round 4), for A/B runs
    a->word_off = value == 0;

  } else {
    return fail(NKA_HIP_EINVAL, "unknown tuning key: " + k);
  }
  return 0;
}
#endif  // NKA_DIAGNOSTIC

int nka_hip_device_info(nka_hip_t a, char *name64, int32_t *num_cu) {
  if (!a) return fail(NKA_HIP_EINVAL, "null handle");
  if (name64) snprintf(name64, 64, "%s", a->devname);
  if (num_cu) *num_cu = a->num_cu;
  return 0;
}

}  // extern "C"
