// vec_ops.hip -- device implementations of the abstract-vector hooks.
//
// The reference's `vector` base class (src-F08-vector/vector_class.F90:90-147)
// asks a concrete vector for eleven deferred procedures; its example
// grid_vector (grid_vector_type.F90:86-197) implements them as whole-array
// expressions.  These are the MI355X equivalents over raw device arrays: one
// HBM-streaming kernel per hook, 16-B/lane accesses, grid-stride persistent
// blocks, fixed-order two-stage reductions (bitwise reproducible).  Compiled
// with -ffp-contract=off so  a*x + b*y + z  rounds as the Fortran expression
// ((a*x) + (b*y)) + z  does.
#include "../../include/nka_hip.h"
#include "../../include/nka_hip_ext.h"
#include "../../include/nka_hip_vec.h"
#include "nka_kernels.hpp"
#include "host_logic.hpp"
#include "rccl_dl.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>

using namespace nka;

extern "C" const char *nka_hip_last_error(void);
namespace nka_detail {
int set_error(int code, const std::string &msg);
std::string last_error();
bool &span_check_failed();   // (thread-local) set by a failing check_device_span
int check_device_span(const void *p, int64_t n, const char *what);
void invalidate_span_cache();
void register_allocation(const void *p, size_t bytes, const void *owner);
void unregister_owner(const void *owner);
void unregister_allocation(const void *p);
}

struct nka_hip_vec_ws {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cu = 256;
  double *partials = nullptr;  // kMaxGrid
  unsigned *tickets = nullptr; // tile-ticket counters of k_update_many_keep_win (kTicketWords, zero between launches)
  int ticket_groups = -1;      // -1 automatic, 0 static tile mapping, 1/2/4/8 counters (nka_hip_vec_set_tuning "tickets")
  double *host_results = nullptr; // pinned, 2*kManyMax+2 doubles
  double *host_results_dev = nullptr;  // its device-side address: the final-sum kernel writes straight into host
                                       // memory (no copy kernel, no staging)
  // parallel-aware reductions ("the vector base class reduction methods will necessarily be
  // parallel-aware", src-F08-vector/README.md:16-22): every sum a reduction returns to the host is first
  // summed over the ranks -- on the device by `allreduce` (stream-ordered; built-in: RCCL), and/or on the
  // host by `host_allreduce` after the stream has been synchronised
  nka_hip_allreduce_fn allreduce = nullptr;
  void *allreduce_ctx = nullptr;
  nka_hip_host_allreduce_fn host_allreduce = nullptr;
  void *host_allreduce_ctx = nullptr;
  ncclComm_t comm = nullptr;
  double *red_dev = nullptr;      // 2*kManyMax+2 doubles: the sums of one reduction in canonical layout
  int sum_order = 0;              // nka_hip_vec_set_sum_order: 1 = dot() sums element after element, unfused (k_dot_ordered)
};

namespace {

#define HIP_TRYV(expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      (void)hipGetLastError(); /* reported here: not left for a later hipGetLastError() */ \
      return nka_detail::set_error(e_ == hipErrorOutOfMemory ? NKA_HIP_ENOMEM : NKA_HIP_EHIP, \
                                   std::string(#expr) + ": " + hipGetErrorString(e_));  \
    }                                                                                    \
  } while (0)

// OP: 0 setval  z = a
//     1 scale   z = a*z                         (grid_vector_type.F90:117)
//     2 update1 z = a*x + z                     (:127)
//     3 update2 z = a*x + b*z                   (:138)
//     4 update3 z = a*x + b*y + z               (:151)
//     5 update4 z = a*x + b*y + c*z             (:165)
template <int OP>
__device__ __forceinline__ double apply(double z, double x, double y, double a, double b, double c) {
  if (OP == 0) return a;
  if (OP == 1) return a * z;
  if (OP == 2) return a * x + z;
  if (OP == 3) return a * x + b * z;
  if (OP == 4) return (a * x + b * y) + z;
  return (a * x + b * y) + c * z;
}

template <int OP, int VEC>
__global__ __launch_bounds__(kBlock) void k_elementwise(int64_t n, double *z, const double *x, const double *y,
                                                        double a, double b, double c) {
  using V = typename VecT<VEC>::type;
  const int G = gridDim.x;
  const int64_t ntile = n / (kBlock * VEC);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    V zv, xv, yv;
    if (OP != 0) zv = ld<VEC>(z + e);
    if (OP >= 2) xv = ld<VEC>(x + e);
    if (OP >= 4) yv = ld<VEC>(y + e);
    V r;
#pragma unroll
    for (int q = 0; q < VEC; q++)
      setc(r, q, apply<OP>(OP != 0 ? ex(zv, q) : 0.0, OP >= 2 ? ex(xv, q) : 0.0, OP >= 4 ? ex(yv, q) : 0.0, a, b, c));
    st(z + e, r);
  }
  if (blockIdx.x == G - 1) {
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock)
      z[i] = apply<OP>(OP != 0 ? z[i] : 0.0, OP >= 2 ? x[i] : 0.0, OP >= 4 ? y[i] : 0.0, a, b, c);
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_dot(int64_t n, const double *__restrict__ x, const double *__restrict__ y,
                                                double *__restrict__ partials) {
  const int G = gridDim.x;
  double acc[1] = {0.0};
  const int64_t ntile = n / (kBlock * VEC);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    auto a = ld<VEC>(x + e);
    auto b = ld<VEC>(y + e);
#pragma unroll
    for (int q = 0; q < VEC; q++) acc[0] = fma(ex(a, q), ex(b, q), acc[0]);
  }
  if (blockIdx.x == G - 1)
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock) acc[0] = fma(x[i], y[i], acc[0]);
  block_reduce_store<1>(acc, partials, G);
}

// <x,y> in the REFERENCE'S ORDER (nka_hip_vec_set_sum_order): element after element, one rounding per product and per
// addition -- what `sum(x*y)` / a plain loop over the elements gives (grid_vector_type.F90:170-183; the oracle's default
// dot product).  One workgroup: chunks of x and y are staged in LDS by all threads, thread 0 walks them (nka::ord_sum: the
// reads of the next eight elements are in flight while the additions of this batch wait for one another).  The sum lands in
// partials[0]; the final-sum kernel then adds nothing to it.
constexpr int kDotOrdChunk = 1024;
static __global__ __launch_bounds__(kBlock) void k_dot_ordered(int64_t n, const double *__restrict__ x, const double *__restrict__ y,
                                                               double *__restrict__ partials) {
  __shared__ double sx[kDotOrdChunk + 16], sy[kDotOrdChunk + 16];
  double acc = 0.0;
  for (int64_t c0 = 0; c0 < n; c0 += kDotOrdChunk) {
    const int len = (int)(n - c0 < kDotOrdChunk ? n - c0 : kDotOrdChunk);
    for (int i = threadIdx.x; i < len; i += kBlock) {
      sx[i] = x[c0 + i];
      sy[i] = y[c0 + i];
    }
    __syncthreads();
    if (threadIdx.x == 0) acc = ord_sum(acc, sx, sy, len);
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[0] = acc;
}

// The same sum for LONG vectors (round 5): the per-sum machinery of the array flavour's reference-order pass (nka_kernels.hpp,
// chain_drive) -- the products of a group are rounded into LDS by eight wavefronts, whole blocks of 1024 go through the chain
// in exact integer arithmetic wherever that is provably the walk's result, the walk elsewhere.  Same bits as k_dot_ordered
// (tests/test_chain_sums_gpu.py holds the machinery to numpy's sequential sum), 10-20 x its speed from n ~ 1e5 on.
static __global__ __launch_bounds__(kChainThreads) void k_dot_chain(int64_t n, const double *__restrict__ x, const double *__restrict__ y,
                                                                    double *__restrict__ partials) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double chain_prod[];   // kChainLdsBytes
  __shared__ ChainSummary summ[kChainGroupBlocks];
  __shared__ double sh_a;
  using V2 = typename VecT<2>::type;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const bool vec16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  constexpr int kPairs = kChainLaneElems / 2;
  static_assert(kPairs == 8, "the pair mapping below assumes 16 elements per lane");
  V2 vx[kPairs], vy[kPairs];
  auto ldpair = [&](const double *p, int64_t i, bool full) -> V2 {
    V2 v;
    if (full && vec16) v = *reinterpret_cast<const V2 *>(p + i);
    else if (full) { v.x = p[i]; v.y = p[i + 1]; }
    else { v.x = i < n ? p[i] : 0.0; v.y = i + 1 < n ? p[i + 1] : 0.0; }
    return v;
  };
  auto load = [&](int64_t g0) {
    const bool full = g0 + kChainGroup <= n;
    const int64_t i0 = g0 + wave * kChainBlock + 2 * lane;
#pragma unroll
    for (int j = 0; j < kPairs; j++) { vx[j] = ldpair(x, i0 + j * 128, full); vy[j] = ldpair(y, i0 + j * 128, full); }
  };
  double *myblk = chain_prod + wave * kChainBlockLds;
  auto store = [&]() {
#pragma unroll
    for (int j = 0; j < kPairs; j++) {
      V2 p;
      p.x = vx[j].x * vy[j].x;
      p.y = vx[j].y * vy[j].y;
      *reinterpret_cast<V2 *>(myblk + (lane % kPairs) * kChainRow + 2 * (j * (64 / kPairs) + lane / kPairs)) = p;
    }
  };
  ChainStamps stamps;
  const double a = chain_drive(0.0, n, chain_prod, summ, &sh_a, 0, stamps, load, store);
  if (t == 0) partials[0] = a;
}
static int launch_dot_ordered(hipStream_t s, int64_t n, const double *x, const double *y, double *partials) {
  if (n <= 4 * kDotOrdChunk) {
    hipLaunchKernelGGL(k_dot_ordered, dim3(1), dim3(kBlock), 0, s, n, x, y, partials);
    return 0;
  }
  // (per DEVICE, not once per process -- a process may drive several GPUs through several workspaces)
  static std::mutex raised_lock;
  static std::set<int> raised_on;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  {
    std::lock_guard<std::mutex> g(raised_lock);
    if (!raised_on.count(dev)) {
      const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dot_chain),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLdsBytes);
      if (raised != hipSuccess)
        return nka_detail::set_error(NKA_HIP_EHIP, std::string("raising the dynamic LDS limit failed: ") + hipGetErrorString(raised));
      raised_on.insert(dev);
    }
  }
  hipLaunchKernelGGL(k_dot_chain, dim3(1), dim3(kChainThreads), kChainLdsBytes, s, n, x, y, partials);
  return 0;
}

// ---- batched hooks: one pass for many dots / many axpys -----------
constexpr int kManyMax = nka_host::kManyMax;   // = 24 vectors per launch (unroll widths 4, 8, ..., 24; window kernels: every width 1..24); longer lists run several launches
// ... of BALANCED widths (round 5; 25 = 13 + 12, not 24 + 1: a launch that is nearly all padding costs as much as a full one --
// profiles/r05/multipass.txt): group p of a list of `count` entries
using nka_host::many_groups;            // host_logic.hpp (pure arithmetic, checked under sanitizers on the CPU)
using nka_host::many_group_width;
struct ManyArgs {
  const double *x[kManyMax];
  const double *y[kManyMax];
  double a[kManyMax], b[kManyMax];
  int count;
};

// partials[j*G + block] = partial <x0, x_j>: x0 read ONCE while the others stream past.
template <int NV, int VEC>
__global__ __launch_bounds__(kBlock) void k_dot_many(int64_t n, const double *__restrict__ x0, ManyArgs m,
                                                     double *__restrict__ partials) {
  using V = typename VecT<VEC>::type;
  const int G = gridDim.x;
  double acc[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) acc[j] = 0.0;
  const int64_t ntile = n / (kBlock * VEC);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    const V xv = ld<VEC>(x0 + e);
    V yv[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) yv[j] = ld<VEC>((j < m.count ? m.x[j] : x0) + e);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NV; j++)
#pragma unroll
      for (int q = 0; q < VEC; q++) acc[j] = fma(ex(xv, q), ex(yv[j], q), acc[j]);
  }
  if (blockIdx.x == G - 1)
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock)
#pragma unroll
      for (int j = 0; j < NV; j++) acc[j] = fma(x0[i], (j < m.count ? m.x[j] : x0)[i], acc[j]);
  block_reduce_store<NV>(acc, partials, G);
}

// Two rows at once: partials[j] = <x0, x_j>, partials[NV+j] = <x1, x_j>, and
// partials[2NV] = <x0, x1>; x0, x1 and every x_j are read ONCE.
template <int NV, int VEC>
__global__ __launch_bounds__(kBlock) void k_dot_pair_many(int64_t n, const double *__restrict__ x0,
                                                          const double *__restrict__ x1, ManyArgs m,
                                                          double *__restrict__ partials) {
  using V = typename VecT<VEC>::type;
  const int G = gridDim.x;
  double acc[2 * NV + 1];
#pragma unroll
  for (int j = 0; j < 2 * NV + 1; j++) acc[j] = 0.0;
  const int64_t ntile = n / (kBlock * VEC);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    const V av = ld<VEC>(x0 + e), bv = ld<VEC>(x1 + e);
    V yv[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) yv[j] = ld<VEC>((j < m.count ? m.x[j] : x0) + e);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < VEC; q++) {
      acc[2 * NV] = fma(ex(av, q), ex(bv, q), acc[2 * NV]);
#pragma unroll
      for (int j = 0; j < NV; j++) {
        acc[j] = fma(ex(av, q), ex(yv[j], q), acc[j]);
        acc[NV + j] = fma(ex(bv, q), ex(yv[j], q), acc[NV + j]);
      }
    }
  }
  if (blockIdx.x == G - 1)
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock) {
      acc[2 * NV] = fma(x0[i], x1[i], acc[2 * NV]);
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const double y = (j < m.count ? m.x[j] : x0)[i];
        acc[j] = fma(x0[i], y, acc[j]);
        acc[NV + j] = fma(x1[i], y, acc[NV + j]);
      }
    }
  block_reduce_store<2 * NV + 1>(acc, partials, G);
}

// z <- (a_j*x_j + b_j*y_j) + z for j = 0..count-1 IN ORDER (the rounding of
// count successive update3_ calls, grid_vector_type.F90:151), z read and written once.
template <int NV, int VEC>
__global__ __launch_bounds__(kBlock) void k_update_many(int64_t n, double *z, ManyArgs m) {
  using V = typename VecT<VEC>::type;
  const int G = gridDim.x;
  const int64_t ntile = n / (kBlock * VEC);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    V zv = ld<VEC>(z + e);
    V xv[NV], yv[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
      xv[j] = ld<VEC>((j < m.count ? m.x[j] : z) + e);
      yv[j] = ld<VEC>((j < m.count ? m.y[j] : z) + e);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NV; j++)
      if (j < m.count)
#pragma unroll
        for (int q = 0; q < VEC; q++) setc(zv, q, (m.a[j] * ex(xv[j], q) + m.b[j] * ex(yv[j], q)) + ex(zv, q));
    st(z + e, zv);
  }
  if (blockIdx.x == G - 1)
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock) {
      double zi = z[i];
#pragma unroll
      for (int j = 0; j < NV; j++)
        if (j < m.count) zi = (m.a[j] * m.x[j][i] + m.b[j] * m.y[j][i]) + zi;
      z[i] = zi;
    }
}

// z <- a_j*x_j + z for j = 0..count-1 IN ORDER (the rounding of count successive
// update1_ calls, grid_vector_type.F90:127), z read and written once.
template <int NV, int VEC>
__global__ __launch_bounds__(kBlock) void k_axpy_many(int64_t n, double *z, ManyArgs m) {
  using V = typename VecT<VEC>::type;
  const int G = gridDim.x;
  const int64_t ntile = n / (kBlock * VEC);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    V zv = ld<VEC>(z + e);
    V xv[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) xv[j] = ld<VEC>((j < m.count ? m.x[j] : z) + e);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NV; j++)
      if (j < m.count)
#pragma unroll
        for (int q = 0; q < VEC; q++) setc(zv, q, m.a[j] * ex(xv[j], q) + ex(zv, q));
    st(z + e, zv);
  }
  if (blockIdx.x == G - 1)
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock) {
      double zi = z[i];
#pragma unroll
      for (int j = 0; j < NV; j++)
        if (j < m.count) zi = m.a[j] * m.x[j][i] + zi;
      z[i] = zi;
    }
}

// ---- fused stages of the vector-flavour update (vector_class: update_norm2,
// scale_dot_pair_many, update_many_keep, axpy_many_keep) ------------------------------

// r = a*x + z (update1_, grid_vector_type.F90:127) and the partial sum of r^2 (norm2,
// :185-197) in the same pass: F08V:237-238 as one kernel.  STORE: z <- r; otherwise z
// is left untouched and the next stage applies the same update before scaling.
template <int VEC, bool STORE>
__global__ __launch_bounds__(kBlock) void k_update_norm2(int64_t n, double *z, const double *__restrict__ x, double a,
                                                         double *__restrict__ partials) {
  using V = typename VecT<VEC>::type;
  const int G = gridDim.x;
  double acc[1] = {0.0};
  const int64_t ntile = n / (kBlock * VEC);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    V zv = ld<VEC>(z + e);
    const V xv = ld<VEC>(x + e);
#pragma unroll
    for (int q = 0; q < VEC; q++) {
      const double r = a * ex(xv, q) + ex(zv, q);
      setc(zv, q, r);
      acc[0] = fma(r, r, acc[0]);
    }
    if (STORE) st(z + e, zv);
  }
  if (blockIdx.x == G - 1)
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock) {
      const double r = a * x[i] + z[i];
      if (STORE) z[i] = r;
      acc[0] = fma(r, r, acc[0]);
    }
  block_reduce_store<1>(acc, partials, G);
}

// F08V:255-264 + :347 in one pass: [PRE: first w <- pre_a*f + w, the update1_ of F08V:237
// when the norm stage left it undone,] w <- a*w, v <- a*v (scale, grid_vector_type.F90:
// 117) [SUB: then v <- (-1)*w + v, the compact option's update1_], both stored, and
// with the NEW w: partials[j] = <w, y_j>, partials[NV+j] = <f, y_j>, partials[2NV] = <f, w>.
template <int NV, int VEC, bool SUB, bool PRE, bool DD = false>
__global__ __launch_bounds__(kBlock) void k_scale_dot_pair_many(int64_t n, double *w, double *v, double a, double pre_a,
                                                                const double *__restrict__ f, ManyArgs m,
                                                                double *__restrict__ partials, int store) {
  // store == 0: a PURE-READ pass -- the same sums, formed with the same a*(pre_a*f + w) in registers, but
  // neither w nor v is written (v is not even read): the combine stage then normalises the pair itself
  // (k_update_many_keep*, `Pend`).
  // DD: one more sum, partials[2NV+1] = <w', w'> of the new w' = a*(pre_a*f + w): with a = 1 the NORM stage
  // and this one are a single pure-read pass (nka_hip_vec_diff_norm_dot_pair_many).
  using V = typename VecT<VEC>::type;
  const int G = gridDim.x;
  constexpr int NA = 2 * NV + 1 + (DD ? 1 : 0);
  double acc[NA];
#pragma unroll
  for (int j = 0; j < NA; j++) acc[j] = 0.0;
  const int64_t ntile = n / (kBlock * VEC);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    V wv = ld<VEC>(w + e), vv = store ? ld<VEC>(v + e) : wv;
    const V fv = ld<VEC>(f + e);
    V yv[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) yv[j] = ld<VEC>((j < m.count ? m.x[j] : f) + e);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < VEC; q++) {
      const double fq = ex(fv, q);
      const double w0 = PRE ? pre_a * fq + ex(wv, q) : ex(wv, q);
      const double wn = a * w0;
      double vn = a * ex(vv, q);
      if (SUB) vn = (-1.0) * wn + vn;
      setc(wv, q, wn);
      setc(vv, q, vn);
      acc[2 * NV] = fma(fq, wn, acc[2 * NV]);
      if (DD) acc[NA - 1] = fma(wn, wn, acc[NA - 1]);
#pragma unroll
      for (int j = 0; j < NV; j++) {
        acc[j] = fma(wn, ex(yv[j], q), acc[j]);
        acc[NV + j] = fma(fq, ex(yv[j], q), acc[NV + j]);
      }
    }
    if (store) {
      st(w + e, wv);
      st(v + e, vv);
    }
  }
  if (blockIdx.x == G - 1)
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock) {
      const double fq = f[i];
      const double w0 = PRE ? pre_a * fq + w[i] : w[i];
      const double wn = a * w0;
      if (store) {
        double vn = a * v[i];
        if (SUB) vn = (-1.0) * wn + vn;
        w[i] = wn;
        v[i] = vn;
      }
      acc[2 * NV] = fma(fq, wn, acc[2 * NV]);
      if (DD) acc[NA - 1] = fma(wn, wn, acc[NA - 1]);
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const double y = (j < m.count ? m.x[j] : f)[i];
        acc[j] = fma(wn, y, acc[j]);
        acc[NV + j] = fma(fq, y, acc[NV + j]);
      }
    }
  block_reduce_store<NA>(acc, partials, G);
}

// F08V:336, 374, 382 in one pass: keep_in <- z (the raw f kept as w_new), then
// z <- (a_j*x_j + b_j*y_j) + z for j in order (PAIRS; update3_, grid_vector_type.F90:151)
// or z <- a_j*x_j + z (!PAIRS; update1_, :127), then keep_out <- z (v_new).
// keep_in / keep_out may be NULL (groups of a list longer than one launch).
// PENDING PAIR.  When the scale-and-dot stage ran as a pure read (k_scale_dot_pair_many*, store == 0),
// entry 0 of the lists is the RAW new pair: w = w1 (not yet differenced when PRE) and v = v1.  The
// combine then forms  wn = a*(pre_a*z_in + w)  [PRE; else a*w],  vn = a*v,  [SUB: vn = (-1)*wn + vn]
// -- the very expressions of the scale-and-dot stage, z_in being f --, stores both, and combines
// with them.  PAIRS: w = xs[0], v = ys[0].  !PAIRS (the v slots hold v - w): v = xs[0], w = pend.w.
struct Pend {
  double *w = nullptr;     // !PAIRS only: the w of the pending pair (PAIRS: xs[0])
  double a = 0.0, pre_a = 0.0;
  int flags = 0;           // 1 active, 2 PRE, 4 SUB
};
__device__ __forceinline__ void pend_normalise(const Pend &pd, double zin, double &w, double &v) {
  const double w0 = (pd.flags & 2) ? pd.pre_a * zin + w : w;
  const double wn = pd.a * w0;
  double vn = pd.a * v;
  if (pd.flags & 4) vn = (-1.0) * wn + vn;
  w = wn;
  v = vn;
}

template <int NV, int VEC, bool PAIRS>
__global__ __launch_bounds__(kBlock) void k_update_many_keep(int64_t n, double *z, ManyArgs m, double *keep_in,
                                                             double *keep_out, Pend pd) {
  using V = typename VecT<VEC>::type;
  const int G = gridDim.x;
  const int64_t ntile = n / (kBlock * VEC);
  const bool pend = (pd.flags & 1) != 0 && m.count > 0;
  double *const pw = PAIRS ? const_cast<double *>(m.x[0]) : pd.w;
  double *const pv = PAIRS ? const_cast<double *>(m.y[0]) : const_cast<double *>(m.x[0]);
  for (int64_t t = blockIdx.x; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    V zv = ld<VEC>(z + e);
    V xv[NV], yv[PAIRS ? NV : 1];
#pragma unroll
    for (int j = 0; j < NV; j++) {
      xv[j] = ld<VEC>((j < m.count ? m.x[j] : z) + e);
      if (PAIRS) yv[j] = ld<VEC>((j < m.count ? m.y[j] : z) + e);
    }
    V pwv = zv;
    if (!PAIRS && pend) pwv = ld<VEC>(pw + e);
    __builtin_amdgcn_sched_barrier(0);
    if (keep_in) st(keep_in + e, zv);
    if (pend) {
      V wq = PAIRS ? xv[0] : pwv, vq = PAIRS ? yv[0] : xv[0];
#pragma unroll
      for (int q = 0; q < VEC; q++) {
        double w = ex(wq, q), v = ex(vq, q);
        pend_normalise(pd, ex(zv, q), w, v);
        setc(wq, q, w);
        setc(vq, q, v);
      }
      st(pw + e, wq);
      st(pv + e, vq);
      if (PAIRS) { xv[0] = wq; yv[0] = vq; } else xv[0] = vq;
    }
#pragma unroll
    for (int j = 0; j < NV; j++)
      if (j < m.count)
#pragma unroll
        for (int q = 0; q < VEC; q++) {
          if (PAIRS) setc(zv, q, (m.a[j] * ex(xv[j], q) + m.b[j] * ex(yv[PAIRS ? j : 0], q)) + ex(zv, q));
          else setc(zv, q, m.a[j] * ex(xv[j], q) + ex(zv, q));
        }
    if (keep_out) st(keep_out + e, zv);
    st(z + e, zv);
  }
  if (blockIdx.x == G - 1)
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock) {
      double zi = z[i];
      if (keep_in) keep_in[i] = zi;
#pragma unroll
      for (int j = 0; j < NV; j++)
        if (j < m.count) {
          double x = m.x[j][i], y = PAIRS ? m.y[j][i] : 0.0;
          if (j == 0 && pend) {
            double w = PAIRS ? x : pw[i], v = PAIRS ? y : x;
            pend_normalise(pd, zi, w, v);      // j == 0: zi is still the input
            pw[i] = w;
            pv[i] = v;
            if (PAIRS) { x = w; y = v; } else x = v;
          }
          zi = PAIRS ? (m.a[j] * x + m.b[j] * y) + zi : m.a[j] * x + zi;
        }
      if (keep_out) keep_out[i] = zi;
      z[i] = zi;
    }
}

// ---- the two heavy stages with a ROLLING WINDOW of loads (16-byte path) -----------------
// Same idea as k_dots_win / k_combine_win of the array flavour (nka_kernels.hpp): the NV
// streamed vectors (pairs) of a tile go through a ring of WIN registers -- slot j mod WIN
// holds vector j, a consumed slot is re-loaded at once with vector j+WIN of this tile or
// j+WIN-NV of the block's next tile -- and the tile's own operands (w, v, f / z) are
// requested one tile ahead.  4-6 loads in flight per wave move more bytes per second
// than all of a tile's loads at once (profiles/r02/hbm_probe_rolling_window.txt).  Same
// arithmetic in the same order => same bits as the kernels above.
// The ring size: a small divisor of the width (4, 5, 6, 3 or 7), or the width itself when it is
// prime.  The window kernels are instantiated for EVERY width 1..kManyMax: a padded entry is a
// cache hit that occupies a ring slot and starves the window (nka_hip.hip:win_ring).
template <int NV>
constexpr int win_ring() {
  return NV % 4 == 0 ? 4 : NV % 5 == 0 ? 5 : NV % 6 == 0 ? 6 : NV % 3 == 0 ? 3 : NV % 7 == 0 ? 7 : NV;
}

template <int NV, bool SUB, bool PRE, bool DD = false, int kWin = win_ring<NV>()>
__global__ __launch_bounds__(kBlock) void k_scale_dot_pair_many_win(int64_t n, double *w, double *v, double a,
                                                                    double pre_a, const double *__restrict__ f,
                                                                    ManyArgs m, double *__restrict__ partials, int store) {
  // store == 0: pure read; DD: also <w', w'> -- see k_scale_dot_pair_many
  constexpr int VEC = 2;
  using V = typename VecT<VEC>::type;
  static_assert(NV % kWin == 0, "the ring must divide the unroll width");
  const int G = gridDim.x;
  constexpr int NA = 2 * NV + 1 + (DD ? 1 : 0);
  double acc[NA];
#pragma unroll
  for (int j = 0; j < NA; j++) acc[j] = 0.0;
  const double *ys[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) ys[j] = (j < m.count) ? m.x[j] : f;
  const int64_t ntile = n / (kBlock * VEC);
  V wv, vv, fv, ring[kWin];
  int64_t t = blockIdx.x;
  if (t < ntile) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    wv = ld<VEC>(w + e);
    vv = store ? ld<VEC>(v + e) : wv;
    fv = ld<VEC>(f + e);
#pragma unroll
    for (int j = 0; j < kWin; j++) ring[j] = ld<VEC>(ys[j] + e);
  }
  for (; t < ntile; t += G) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    const int64_t tn = (t + G < ntile) ? t + G : t;     // the last iteration prefetches its own tile again
    const int64_t en = tn * (kBlock * VEC) + threadIdx.x * VEC;
    double wn[VEC], fq[VEC];
    V wout = wv, vout = vv;
#pragma unroll
    for (int q = 0; q < VEC; q++) {
      fq[q] = ex(fv, q);
      const double w0 = PRE ? pre_a * fq[q] + ex(wv, q) : ex(wv, q);
      wn[q] = a * w0;
      double vn = a * ex(vv, q);
      if (SUB) vn = (-1.0) * wn[q] + vn;
      setc(wout, q, wn[q]);
      setc(vout, q, vn);
      acc[2 * NV] = fma(fq[q], wn[q], acc[2 * NV]);
      if (DD) acc[NA - 1] = fma(wn[q], wn[q], acc[NA - 1]);
    }
    if (store) {
      st(w + e, wout);
      st(v + e, vout);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tn != t) {                       // (on the last iteration w, v of this tile were just rewritten)
      wv = ld<VEC>(w + en);
      if (store) vv = ld<VEC>(v + en);
    }
    fv = ld<VEC>(f + en);
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const V y = ring[j % kWin];
      __builtin_amdgcn_sched_barrier(0);
      if (j + kWin < NV) ring[j % kWin] = ld<VEC>(ys[j + kWin] + e);
      else ring[j % kWin] = ld<VEC>(ys[j + kWin - NV] + en);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < VEC; q++) {
        acc[j] = fma(wn[q], ex(y, q), acc[j]);
        acc[NV + j] = fma(fq[q], ex(y, q), acc[NV + j]);
      }
    }
  }
  if (blockIdx.x == G - 1)
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock) {
      const double fq = f[i];
      const double w0 = PRE ? pre_a * fq + w[i] : w[i];
      const double wn = a * w0;
      if (store) {
        double vn = a * v[i];
        if (SUB) vn = (-1.0) * wn + vn;
        w[i] = wn;
        v[i] = vn;
      }
      acc[2 * NV] = fma(fq, wn, acc[2 * NV]);
      if (DD) acc[NA - 1] = fma(wn, wn, acc[NA - 1]);
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const double y = ys[j][i];
        acc[j] = fma(wn, y, acc[j]);
        acc[NV + j] = fma(fq, y, acc[NV + j]);
      }
    }
  block_reduce_store<NA>(acc, partials, G);
}

// two vectors per entry: half the ring keeps the same number of loads in flight (nka_hip.hip:win_ring_pairs)
template <int NV>
constexpr int win_ring_pairs() {
  return NV % 2 == 0 ? 2 : NV % 3 == 0 ? 3 : NV % 5 == 0 ? 5 : NV % 7 == 0 ? 7 : NV;
}

template <int NV, bool PAIRS, int kWin = (PAIRS ? win_ring_pairs<NV>() : win_ring<NV>())>
__global__ __launch_bounds__(kBlock) void k_update_many_keep_win(int64_t n, double *z, ManyArgs m, double *keep_in,
                                                                 double *keep_out, unsigned *tickets, int ng, Pend pd) {
  // `tickets` != nullptr: tiles from global ticket counters (compact front), as in k_combine_win
  // `pd`: entry 0 is the raw pending pair, normalised here (see Pend)
  constexpr int VEC = 2;
  using V = typename VecT<VEC>::type;
  static_assert(NV % kWin == 0, "the ring must divide the unroll width");
  __shared__ unsigned s_next[2];
  const int G = gridDim.x;
  const bool pend = (pd.flags & 1) != 0 && m.count > 0;
  double *const pw = PAIRS ? const_cast<double *>(m.x[0]) : pd.w;
  double *const pv = PAIRS ? const_cast<double *>(m.y[0]) : const_cast<double *>(m.x[0]);
  const double *const pwsrc = (!PAIRS && pend) ? pw : z;    // !PAIRS: the raw w of the pending pair, one more stream
  const double *xs[NV], *ys[PAIRS ? NV : 1];
#pragma unroll
  for (int j = 0; j < NV; j++) {
    xs[j] = (j < m.count) ? m.x[j] : z;
    if (PAIRS) ys[j] = (j < m.count) ? m.y[j] : z;
  }
  const int64_t ntile = n / (kBlock * VEC);
  V znext, pwnext, rx[kWin], ry[PAIRS ? kWin : 1];
  int64_t t = blockIdx.x;
  if (t < ntile) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    znext = ld<VEC>(z + e);
    if (!PAIRS) pwnext = ld<VEC>(pwsrc + e);
#pragma unroll
    for (int j = 0; j < kWin; j++) {
      rx[j] = ld<VEC>(xs[j] + e);
      if (PAIRS) ry[j] = ld<VEC>(ys[j] + e);
    }
  }
  const unsigned grp = tickets ? blockIdx.x % (unsigned)ng : 0u;
  unsigned *const my_ticket = tickets ? tickets + grp * kTicketStride : nullptr;
  const unsigned ticket_base = tickets ? 2u * (unsigned)G / (unsigned)ng : 0u;
  int64_t tnext = t + G;
  unsigned par = 0;
  while (t < ntile) {
    const int64_t e = t * (kBlock * VEC) + threadIdx.x * VEC;
    const bool more = tnext < ntile;
    unsigned claimed = kNoTicket;
    if (tickets && more && threadIdx.x == 0) claimed = ticket_request(my_ticket, ticket_base, (unsigned)ng, grp);
    const int64_t tn = more ? tnext : t;                 // the last iteration prefetches its own tile again
    const int64_t en = tn * (kBlock * VEC) + threadIdx.x * VEC;
    V zv = znext;
    const V zin = zv, pwv = PAIRS ? zv : pwnext;
    if (keep_in) st(keep_in + e, zv);
    __builtin_amdgcn_sched_barrier(0);
    if (tn != t) {
      znext = ld<VEC>(z + en);
      if (!PAIRS) pwnext = ld<VEC>(pwsrc + en);
    }
#pragma unroll
    for (int j = 0; j < NV; j++) {
      V xj = rx[j % kWin];
      V yj = ry[PAIRS ? j % kWin : 0];
      __builtin_amdgcn_sched_barrier(0);
      if (j + kWin < NV) {
        rx[j % kWin] = ld<VEC>(xs[j + kWin] + e);
        if (PAIRS) ry[j % kWin] = ld<VEC>(ys[j + kWin] + e);
      } else {
        rx[j % kWin] = ld<VEC>(xs[j + kWin - NV] + en);
        if (PAIRS) ry[j % kWin] = ld<VEC>(ys[j + kWin - NV] + en);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (j == 0 && pend) {
        V wq = PAIRS ? xj : pwv, vq = PAIRS ? yj : xj;
#pragma unroll
        for (int q = 0; q < VEC; q++) {
          double w = ex(wq, q), v = ex(vq, q);
          pend_normalise(pd, ex(zin, q), w, v);
          setc(wq, q, w);
          setc(vq, q, v);
        }
        st(pw + e, wq);
        st(pv + e, vq);
        if (PAIRS) { xj = wq; yj = vq; } else xj = vq;
      }
      if (j < m.count) {
#pragma unroll
        for (int q = 0; q < VEC; q++) {
          if (PAIRS) setc(zv, q, (m.a[j] * ex(xj, q) + m.b[j] * ex(yj, q)) + ex(zv, q));
          else setc(zv, q, m.a[j] * ex(xj, q) + ex(zv, q));
        }
      }
    }
    if (keep_out) st(keep_out + e, zv);
    st(z + e, zv);
    const int64_t t2 = tickets ? ticket_publish(s_next, par, claimed, ntile) : tnext + G;
    t = tnext;
    tnext = t2;
  }
  if (tickets) ticket_finish(tickets, ng, G);
  if (blockIdx.x == G - 1)
    for (int64_t i = ntile * (kBlock * VEC) + threadIdx.x; i < n; i += kBlock) {
      double zi = z[i];
      if (keep_in) keep_in[i] = zi;
#pragma unroll
      for (int j = 0; j < NV; j++)
        if (j < m.count) {
          double x = m.x[j][i], y = PAIRS ? m.y[j][i] : 0.0;
          if (j == 0 && pend) {
            double w = PAIRS ? x : pw[i], v = PAIRS ? y : x;
            pend_normalise(pd, zi, w, v);      // j == 0: zi is still the input
            pw[i] = w;
            pv[i] = v;
            if (PAIRS) { x = w; y = v; } else x = v;
          }
          zi = PAIRS ? (m.a[j] * x + m.b[j] * y) + zi : m.a[j] * x + zi;
        }
      if (keep_out) keep_out[i] = zi;
      z[i] = zi;
    }
}

// Persistent grid.  Light kernels (a few loads per thread) fill the chip with 8
// blocks per CU; the fused many-vector kernels keep `nloads` 16-byte loads per
// thread in flight and follow the rule measured for the array flavour (ONE block
// per CU once a block has >= 22 loads per thread in flight; nka_hip.hip:grid_for).
int grid_for(const nka_hip_vec_ws *ws, int64_t n, int vec, int nloads = 2) {
  const int per_cu = std::max(1, std::min(8, (22 + nloads - 1) / nloads));
  int64_t g = (int64_t)ws->num_cu * per_cu;
  g = std::min<int64_t>(g, std::max<int64_t>(n / (kBlock * vec), 1));
  return (int)std::min<int64_t>(g, kMaxGrid);
}

int width_for(int count) { return std::max(4, ((count + 3) / 4) * 4); }   // unroll width 4, 8, ..., kManyMax

#define NKA_DISPATCH_NV(nv, CALL) \
  switch (nv) {                  \
    case 4: CALL(4); break;      \
    case 8: CALL(8); break;      \
    case 12: CALL(12); break;    \
    case 16: CALL(16); break;    \
    case 20: CALL(20); break;    \
    default: CALL(24); break;    \
  }

#define NKA_DISPATCH_EXACT(nv, CALL)                                                              \
  switch (nv) {                                                                                   \
    case 1: CALL(1); break;   case 2: CALL(2); break;   case 3: CALL(3); break;   case 4: CALL(4); break;   \
    case 5: CALL(5); break;   case 6: CALL(6); break;   case 7: CALL(7); break;   case 8: CALL(8); break;   \
    case 9: CALL(9); break;   case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break; \
    case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; case 16: CALL(16); break; \
    case 17: CALL(17); break; case 18: CALL(18); break; case 19: CALL(19); break; case 20: CALL(20); break; \
    case 21: CALL(21); break; case 22: CALL(22); break; case 23: CALL(23); break; default: CALL(24); break; \
  }
static_assert(kManyMax == 24, "NKA_DISPATCH_EXACT covers widths 1..24");

bool al16(const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

template <int OP>
int run_elementwise(nka_hip_vec_ws *ws, int64_t n, double *z, const double *x, const double *y, double a, double b,
                    double c) {
  if (!ws) return nka_detail::set_error(NKA_HIP_EINVAL, "null workspace");
  if (n < 0) return nka_detail::set_error(NKA_HIP_EINVAL, "negative length");
  if (n == 0) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  if (int rc = nka_detail::check_device_span(z, n, "vector hook: z")) return rc;
  if (OP >= 2) if (int rc = nka_detail::check_device_span(x, n, "vector hook: x")) return rc;
  if (OP >= 4) if (int rc = nka_detail::check_device_span(y, n, "vector hook: y")) return rc;
  const bool v2 = al16(z) && (OP < 2 || al16(x)) && (OP < 4 || al16(y));
  const int g = grid_for(ws, n, v2 ? 2 : 1);
  if (v2)
    hipLaunchKernelGGL((k_elementwise<OP, 2>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, x, y, a, b, c);
  else
    hipLaunchKernelGGL((k_elementwise<OP, 1>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, x, y, a, b, c);
  HIP_TRYV(hipGetLastError());
  return 0;
}

}  // namespace

namespace {
template <bool PAIRS>
int update_many_keep(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a, const double *const *xs, const double *b,
                     const double *const *ys, int32_t count, double *keep_in, double *keep_out, const char *who,
                     Pend pend = Pend()) {
  if (!ws || n < 0 || count < 0 || (count > 0 && (!a || !xs || (PAIRS && (!b || !ys)))))
    return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  if ((pend.flags & 1) && (count < 1 || (!PAIRS && !pend.w)))
    return nka_detail::set_error(NKA_HIP_EINVAL, "pending pair: entry 0 of the lists (and its w) must be given");
  if (n == 0) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  if ((pend.flags & 1) && !PAIRS) if (int rc = nka_detail::check_device_span(pend.w, n, who)) return rc;
  if (int rc = nka_detail::check_device_span(z, n, who)) return rc;
  if (keep_in) if (int rc = nka_detail::check_device_span(keep_in, n, who)) return rc;
  if (keep_out) if (int rc = nka_detail::check_device_span(keep_out, n, who)) return rc;
  for (int j = 0; j < count; j++) {
    if (int rc = nka_detail::check_device_span(xs[j], n, who)) return rc;
    if (PAIRS) if (int rc = nka_detail::check_device_span(ys[j], n, who)) return rc;
  }
  int base = 0;
  const int ngroups = many_groups(count);
  for (int grp = 0; grp < ngroups; grp++) {   // at least one launch: with count == 0 the two keeps are still written
    ManyArgs m{};
    m.count = many_group_width(count, grp);
    double *kin = (grp == 0) ? keep_in : nullptr;
    double *kout = (grp == ngroups - 1) ? keep_out : nullptr;
    bool v2 = al16(z) && (!kin || al16(kin)) && (!kout || al16(kout));
    for (int j = 0; j < m.count; j++) {
      m.x[j] = xs[base + j];
      m.a[j] = a[base + j];
      v2 = v2 && al16(m.x[j]);
      if (PAIRS) {
        m.y[j] = ys[base + j];
        m.b[j] = b[base + j];
        v2 = v2 && al16(m.y[j]);
      }
    }
    Pend pd = (base == 0) ? pend : Pend();     // the pending pair is entry 0 of the first launch
    if ((pd.flags & 1) && !PAIRS) v2 = v2 && al16(pd.w);
    const bool win = v2;                                                    // 16-byte path = the rolling-window kernel
    const int nv = win ? std::max(m.count, 1) : width_for(m.count);      // window kernels: exact width, no padding
    const int g = grid_for(ws, n, v2 ? 2 : 1, win ? 22 : (PAIRS ? 2 : 1) * nv + 1);   // rolling-window kernels: one block per CU
    // tile tickets (k_combine_win): one counter while a tile carries >= 22 words per element, else two
    const int words = (PAIRS ? 2 : 1) * nv + 1 + 1 + (kin ? 1 : 0) + (kout ? 1 : 0) + ((pd.flags & 1) ? (PAIRS ? 2 : 3) : 0);
    int ng = ws->ticket_groups;
    if (ng < 0) ng = (win && n / (kBlock * 2) >= (int64_t)64 * g) ? (words >= 22 ? 1 : 2) : 0;
    if (!win || !ws->tickets || g % std::max(ng, 1) != 0 || n / (kBlock * 2) >= ((int64_t)1 << 31) - 2 * kMaxGrid) ng = 0;
    unsigned *const tix = ng > 0 ? ws->tickets : nullptr;
#define LAUNCHW(NV) hipLaunchKernelGGL((k_update_many_keep_win<NV, PAIRS>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, m, kin, kout, tix, std::max(ng, 1), pd)
#define LAUNCH1(NV) hipLaunchKernelGGL((k_update_many_keep<NV, 1, PAIRS>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, m, kin, kout, pd)
    if (win) { NKA_DISPATCH_EXACT(nv, LAUNCHW) } else { NKA_DISPATCH_NV(nv, LAUNCH1) }   // (unaligned operands: 8-byte path)
#undef LAUNCHW
#undef LAUNCH1
    HIP_TRYV(hipGetLastError());
    base += m.count;
  }
  return 0;
}

// ---- final sums of a reduction -> host, summed over the ranks when hooks are installed ------
// The reduction kernel just enqueued left per-block partials of `rows` rows of `nv` columns (plus
// one `cross` column); the first `count` columns of each row are wanted.  One block per column sums
// its partials in a fixed order (thread t adds the blocks t, t+256, ... in that order, then the butterfly of its wave,
// then the four wave sums in order: never depends on timing) and writes the CANONICAL layout
//   out[r*count + j] = row r, column j   ;   out[rows*count + e] = extra scalar column e (cross, <w',w'>)
// which depends only on (rows, count) -- never on the padded width nv of the kernel variant a rank
// happened to take (alignment of its pointers) --, so the ranks of a sharded run always reduce
// the same number of values in the same places.
static __global__ __launch_bounds__(kBlock) void k_finalize_rows(const double *__restrict__ partials, int G, int rows,
                                                                 int nv, int count, double *__restrict__ out) {
  __shared__ double sm[kWavesPerBlock];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x;
  int dst;
  if (c < rows * nv) {
    const int r = c / nv, j = c - r * nv;
    if (j >= count) return;                     // padding column of the unrolled kernel (whole block leaves)
    dst = r * count + j;
  } else {
    dst = rows * count + (c - rows * nv);       // the scalar columns behind the rows
  }
  double r = 0.0;
  for (int b = threadIdx.x; b < G; b += kBlock) r += partials[(size_t)c * G + b];
  r = wave_sum(r);
  if (lane == 0) sm[wv] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = sm[0];
#pragma unroll
    for (int q = 1; q < kWavesPerBlock; q++) t += sm[q];
    out[dst] = t;
  }
}

bool ws_parallel(const nka_hip_vec_ws *ws) { return ws->allreduce || ws->host_allreduce; }

// A rank whose OWN arguments fail the pointer check must not simply return from a parallel reduction: its peers are (or
// will be) waiting in the collective.  Every reduction entry checks its pointers before its first launch and handles an
// empty slice (n == 0) by joining each of its collectives with zeros, so: run the entry; if it came back because of a
// pointer check, run it once more as an empty slice -- nobody hangs -- and report the error.  The caller's error stop
// then ends the job the usual way.
template <class F>
int join_on_bad_pointers(nka_hip_vec_ws_t ws, int64_t n, F &&entry) {
  nka_detail::span_check_failed() = false;
  const int rc = entry(n);
  if (rc && ws && n > 0 && nka_detail::span_check_failed() && ws_parallel(ws)) {
    const std::string msg = nka_detail::last_error();
    (void)entry(0);
    nka_detail::set_error(rc, msg);
  }
  return rc;
}

int run_host_hook(nka_hip_vec_ws_t ws, double *vals, int total) {
  if (!ws->host_allreduce || total <= 0) return 0;
  if (int rc = ws->host_allreduce(ws->host_allreduce_ctx, vals, total))
    return rc < 0 ? rc : nka_detail::set_error(NKA_HIP_ECOMM, "vector reduction: the host all-reduce hook failed");
  return 0;
}

// Results land in ws->host_results (canonical layout).  have == false: this rank's slice is
// empty, no kernel ran; its contribution is zero but it still takes part in every collective.
int fetch_sums(nka_hip_vec_ws_t ws, int g, int rows, int nv, int count, int extra, bool have) {
  const int total = rows * count + extra;     // `extra` scalar columns behind the rows (0, 1: cross, 2: cross and <w',w'>)
  if (total <= 0) return 0;
  const int ncols = rows * nv + extra;
  if (!ws->allreduce) {
    if (have) {
      // single rank (or host-side hook only): straight into pinned host memory, no copy in between
      hipLaunchKernelGGL(k_finalize_rows, dim3(ncols), dim3(kBlock), 0, ws->stream, ws->partials, g, rows, nv, count,
                         ws->host_results_dev);
      HIP_TRYV(hipGetLastError());
      HIP_TRYV(hipStreamSynchronize(ws->stream));
    } else {
      for (int i = 0; i < total; i++) ws->host_results[i] = 0.0;
    }
    return run_host_hook(ws, ws->host_results, total);
  }
  if (have) {
    hipLaunchKernelGGL(k_finalize_rows, dim3(ncols), dim3(kBlock), 0, ws->stream, ws->partials, g, rows, nv, count, ws->red_dev);
    HIP_TRYV(hipGetLastError());
  } else {
    HIP_TRYV(hipMemsetAsync(ws->red_dev, 0, sizeof(double) * (size_t)total, ws->stream));
  }
  if (int rc = ws->allreduce(ws->allreduce_ctx, ws->red_dev, total, ws->stream))
    return rc < 0 ? rc : nka_detail::set_error(NKA_HIP_ECOMM, "vector reduction: the all-reduce hook failed");
  HIP_TRYV(hipMemcpyAsync(ws->host_results, ws->red_dev, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, ws->stream));
  HIP_TRYV(hipStreamSynchronize(ws->stream));
  return run_host_hook(ws, ws->host_results, total);
}

int rccl_vec_allreduce(void *ctx, double *buf, int32_t count, void *stream) {
  auto *ws = static_cast<nka_hip_vec_ws *>(ctx);
  const auto &R = nka_detail::rccl();
  ncclResult_t r = R.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, ws->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return nka_detail::set_error(NKA_HIP_ECOMM, std::string("ncclAllReduce: ") + R.GetErrorString(r));
  return 0;
}
}  // namespace

extern "C" {

int nka_hip_vec_workspace_create(nka_hip_vec_ws_t *out, int32_t device, void *stream) {
  if (!out) return nka_detail::set_error(NKA_HIP_EINVAL, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  HIP_TRYV(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return nka_detail::set_error(NKA_HIP_EINVAL, "no such HIP device");
  HIP_TRYV(hipSetDevice(device));
  auto *ws = new nka_hip_vec_ws();
  ws->device = device;
  ws->stream = (hipStream_t)stream;  // NULL = HIP's default stream
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e == hipSuccess) ws->num_cu = prop.multiProcessorCount;
  if (e == hipSuccess) e = hipMalloc((void **)&ws->partials, sizeof(double) * kMaxGrid * (2 * kManyMax + 2));
  if (e == hipSuccess) e = hipMalloc((void **)&ws->tickets, sizeof(unsigned) * kTicketWords);
  if (e == hipSuccess) e = hipMemset(ws->tickets, 0, sizeof(unsigned) * kTicketWords);
  if (e == hipSuccess) e = hipHostMalloc((void **)&ws->host_results, sizeof(double) * (2 * kManyMax + 2), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ws->host_results_dev, ws->host_results, 0);
  if (e == hipSuccess) e = hipMalloc((void **)&ws->red_dev, sizeof(double) * (2 * kManyMax + 2));
  if (e != hipSuccess) {   // free whatever was obtained (hipFree / hipHostFree accept NULL)
    (void)hipGetLastError();   // reported here: the sticky last error must not fail the next, unrelated call
    nka_hip_vec_workspace_destroy(ws);
    return nka_detail::set_error(e == hipErrorOutOfMemory ? NKA_HIP_ENOMEM : NKA_HIP_EHIP,
                                 std::string("vec_workspace_create: ") + hipGetErrorString(e));
  }
  *out = ws;
  return 0;
}

int nka_hip_vec_workspace_destroy(nka_hip_vec_ws_t ws) {
  if (!ws) return 0;
  nka_detail::invalidate_span_cache();
  nka_detail::unregister_owner(ws);      // its vectors can no longer be freed through the library: no entry may outlive it
  hipSetDevice(ws->device);
  hipStreamSynchronize(ws->stream);
  hipFree(ws->partials);
  hipFree(ws->tickets);
  if (ws->comm) nka_detail::rccl().CommDestroy(ws->comm);
  hipFree(ws->red_dev);
  hipHostFree(ws->host_results);
  delete ws;
  return 0;
}

#ifdef NKA_DIAGNOSTIC      // only in libnka_hip_diag.so (include/nka_hip_diag.h)
int nka_hip_vec_set_tuning(nka_hip_vec_ws_t ws, const char *key, int32_t value) {
  if (!ws || !key) return nka_detail::set_error(NKA_HIP_EINVAL, "null argument");
  if (std::string(key) == "tickets") {
    if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
      return nka_detail::set_error(NKA_HIP_EINVAL, "tickets: -1 (auto), 0 (static tile mapping), 1, 2, 4, 8 (ticket counters)");
    ws->ticket_groups = value;
    return 0;
  }
  return nka_detail::set_error(NKA_HIP_EINVAL, std::string("unknown tuning key: ") + key);
}
#endif  // NKA_DIAGNOSTIC

int nka_hip_vec_alloc(nka_hip_vec_ws_t ws, int64_t n, double **out_dev) {
  if (!ws || !out_dev || n < 0) return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  HIP_TRYV(hipSetDevice(ws->device));
  HIP_TRYV(hipMalloc((void **)out_dev, sizeof(double) * (size_t)std::max<int64_t>(n, 1)));
  nka_detail::register_allocation(*out_dev, sizeof(double) * (size_t)std::max<int64_t>(n, 1), ws);   // exact pointer checks, no HIP call
  return 0;
}

int nka_hip_vec_free(nka_hip_vec_ws_t ws, double *dev) {
  if (!ws) return nka_detail::set_error(NKA_HIP_EINVAL, "null workspace");
  HIP_TRYV(hipSetDevice(ws->device));
  HIP_TRYV(hipStreamSynchronize(ws->stream));
  nka_detail::invalidate_span_cache();
  nka_detail::unregister_allocation(dev);
  HIP_TRYV(hipFree(dev));
  return 0;
}

int nka_hip_vec_copy(nka_hip_vec_ws_t ws, int64_t n, double *dst, const double *src) {
  if (!ws || n < 0) return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  if (n == 0 || dst == src) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  if (int rc = nka_detail::check_device_span(dst, n, "vec_copy: dst")) return rc;
  if (int rc = nka_detail::check_device_span(src, n, "vec_copy: src")) return rc;
  HIP_TRYV(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ws->stream));
  return 0;
}

int nka_hip_vec_setval(nka_hip_vec_ws_t ws, int64_t n, double *x, double val) {
  return run_elementwise<0>(ws, n, x, nullptr, nullptr, val, 0, 0);
}
int nka_hip_vec_scale(nka_hip_vec_ws_t ws, int64_t n, double *x, double a) {
  return run_elementwise<1>(ws, n, x, nullptr, nullptr, a, 0, 0);
}
int nka_hip_vec_update1(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x) {
  return run_elementwise<2>(ws, n, z, x, nullptr, a, 0, 0);
}
int nka_hip_vec_update2(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x, double b) {
  return run_elementwise<3>(ws, n, z, x, nullptr, a, b, 0);
}
int nka_hip_vec_update3(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x, double b,
                        const double *y) {
  return run_elementwise<4>(ws, n, z, x, y, a, b, 0);
}
int nka_hip_vec_update4(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x, double b,
                        const double *y, double c) {
  return run_elementwise<5>(ws, n, z, x, y, a, b, c);
}

static int nka_hip_vec_dot_entry(nka_hip_vec_ws_t ws, int64_t n, const double *x, const double *y, double *host_result) {
  if (!ws || !host_result || n < 0) return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  *host_result = 0.0;
  if (n == 0 && !ws_parallel(ws)) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  int g = 1;
  if (n > 0) {
    if (int rc = nka_detail::check_device_span(x, n, "vec_dot: x")) return rc;
    if (int rc = nka_detail::check_device_span(y, n, "vec_dot: y")) return rc;
    const bool v2 = al16(x) && al16(y);
    g = ws->sum_order == 1 ? 1 : grid_for(ws, n, v2 ? 2 : 1);
    if (ws->sum_order == 1)
      { if (int rc = launch_dot_ordered(ws->stream, n, x, y, ws->partials)) return rc; }
    else if (v2)
      hipLaunchKernelGGL((k_dot<2>), dim3(g), dim3(kBlock), 0, ws->stream, n, x, y, ws->partials);
    else
      hipLaunchKernelGGL((k_dot<1>), dim3(g), dim3(kBlock), 0, ws->stream, n, x, y, ws->partials);
  }
  if (int rc = fetch_sums(ws, g, 1, 1, 1, 0, n > 0)) return rc;
  *host_result = ws->host_results[0];
  return 0;
}
int nka_hip_vec_dot(nka_hip_vec_ws_t ws, int64_t n, const double *x, const double *y, double *host_result) {
  return join_on_bad_pointers(ws, n, [&](int64_t n_) { return nka_hip_vec_dot_entry(ws, n_, x, y, host_result); });
}

int nka_hip_vec_norm2(nka_hip_vec_ws_t ws, int64_t n, const double *x, double *host_result) {
  double d = 0.0;
  if (int rc = nka_hip_vec_dot(ws, n, x, x, &d)) return rc;
  *host_result = std::sqrt(d);  // grid_vector_type.F90:185-197
  return 0;
}

// vals[j] = <x, ys[j]>, j < count: x is read once per group of kManyMax vectors.
static int nka_hip_vec_dot_many_entry(nka_hip_vec_ws_t ws, int64_t n, const double *x, const double *const *ys, int32_t count,
                         double *host_vals) {
  if (ws && ws->sum_order == 1)     // (the vector types fall back to the deferred hooks then: dot(), whose sum IS ordered)
    return nka_detail::set_error(NKA_HIP_ESTATE, "this batched reduction sums in blocks; with reference-order sums "
                                 "(nka_hip_vec_set_sum_order) the vector type must use dot() / norm2()");
  if (!ws || n < 0 || count < 0 || (count > 0 && (!ys || !host_vals))) return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  for (int j = 0; j < count; j++) host_vals[j] = 0.0;
  if (count == 0 || (n == 0 && !ws_parallel(ws))) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  if (n > 0) {
    if (int rc = nka_detail::check_device_span(x, n, "vec_dot_many: x")) return rc;
    for (int j = 0; j < count; j++)
      if (int rc = nka_detail::check_device_span(ys[j], n, "vec_dot_many: ys[j]")) return rc;
  }
  for (int base = 0; base < count; base += kManyMax) {
    ManyArgs m{};
    m.count = std::min(kManyMax, count - base);
    if (n == 0) {                      // empty slice: zeros, but the collective is still joined
      if (int rc = fetch_sums(ws, 1, 1, m.count, m.count, 0, false)) return rc;
      for (int j = 0; j < m.count; j++) host_vals[base + j] = ws->host_results[j];
      continue;
    }
    bool v2 = al16(x);
    for (int j = 0; j < m.count; j++) {
      m.x[j] = ys[base + j];
      v2 = v2 && al16(m.x[j]);
    }
    const int nv = width_for(m.count);
    const int g = grid_for(ws, n, v2 ? 2 : 1, nv + 1);
#define LAUNCH2(NV) hipLaunchKernelGGL((k_dot_many<NV, 2>), dim3(g), dim3(kBlock), 0, ws->stream, n, x, m, ws->partials)
#define LAUNCH1(NV) hipLaunchKernelGGL((k_dot_many<NV, 1>), dim3(g), dim3(kBlock), 0, ws->stream, n, x, m, ws->partials)
    if (v2) { NKA_DISPATCH_NV(nv, LAUNCH2) } else { NKA_DISPATCH_NV(nv, LAUNCH1) }
#undef LAUNCH2
#undef LAUNCH1
    HIP_TRYV(hipGetLastError());
    if (int rc = fetch_sums(ws, g, 1, nv, m.count, 0, true)) return rc;
    for (int j = 0; j < m.count; j++) host_vals[base + j] = ws->host_results[j];
  }
  return 0;
}
int nka_hip_vec_dot_many(nka_hip_vec_ws_t ws, int64_t n, const double *x, const double *const *ys, int32_t count,
                         double *host_vals) {
  return join_on_bad_pointers(ws, n, [&](int64_t n_) { return nka_hip_vec_dot_many_entry(ws, n_, x, ys, count, host_vals); });
}

// vals0[j] = <x0, ys[j]>, vals1[j] = <x1, ys[j]>, *cross = <x0, x1>: both rows of
// the Gram update in ONE pass over the stored vectors.
static int nka_hip_vec_dot_pair_many_entry(nka_hip_vec_ws_t ws, int64_t n, const double *x0, const double *x1,
                              const double *const *ys, int32_t count, double *host_vals0, double *host_vals1,
                              double *host_cross) {
  if (ws && ws->sum_order == 1)     // (the vector types fall back to the deferred hooks then: dot(), whose sum IS ordered)
    return nka_detail::set_error(NKA_HIP_ESTATE, "this batched reduction sums in blocks; with reference-order sums "
                                 "(nka_hip_vec_set_sum_order) the vector type must use dot() / norm2()");
  if (!ws || n < 0 || count < 0 || !host_cross || (count > 0 && (!ys || !host_vals0 || !host_vals1)))
    return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  *host_cross = 0.0;
  for (int j = 0; j < count; j++) host_vals0[j] = host_vals1[j] = 0.0;
  if (n == 0 && !ws_parallel(ws)) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  if (n > 0) {
    if (int rc = nka_detail::check_device_span(x0, n, "vec_dot_pair_many: x0")) return rc;
    if (int rc = nka_detail::check_device_span(x1, n, "vec_dot_pair_many: x1")) return rc;
    for (int j = 0; j < count; j++)
      if (int rc = nka_detail::check_device_span(ys[j], n, "vec_dot_pair_many: ys[j]")) return rc;
  }
  int base = 0;
  do {  // at least one launch so that <x0,x1> is computed even when count == 0
    ManyArgs m{};
    m.count = std::max(0, std::min(kManyMax, count - base));
    if (n == 0) {                      // empty slice: zeros, but the collective is still joined
      if (int rc = fetch_sums(ws, 1, 2, std::max(m.count, 1), m.count, 1, false)) return rc;
      for (int j = 0; j < m.count; j++) {
        host_vals0[base + j] = ws->host_results[j];
        host_vals1[base + j] = ws->host_results[m.count + j];
      }
      if (base == 0) *host_cross = ws->host_results[2 * m.count];
      base += kManyMax;
      continue;
    }
    bool v2 = al16(x0) && al16(x1);
    for (int j = 0; j < m.count; j++) {
      m.x[j] = ys[base + j];
      v2 = v2 && al16(m.x[j]);
    }
    const int nv = width_for(m.count);
    const int g = grid_for(ws, n, v2 ? 2 : 1, nv + 2);
#define LAUNCH2(NV) hipLaunchKernelGGL((k_dot_pair_many<NV, 2>), dim3(g), dim3(kBlock), 0, ws->stream, n, x0, x1, m, ws->partials)
#define LAUNCH1(NV) hipLaunchKernelGGL((k_dot_pair_many<NV, 1>), dim3(g), dim3(kBlock), 0, ws->stream, n, x0, x1, m, ws->partials)
    if (v2) { NKA_DISPATCH_NV(nv, LAUNCH2) } else { NKA_DISPATCH_NV(nv, LAUNCH1) }
#undef LAUNCH2
#undef LAUNCH1
    HIP_TRYV(hipGetLastError());
    if (int rc = fetch_sums(ws, g, 2, nv, m.count, 1, true)) return rc;
    for (int j = 0; j < m.count; j++) {
      host_vals0[base + j] = ws->host_results[j];
      host_vals1[base + j] = ws->host_results[m.count + j];
    }
    if (base == 0) *host_cross = ws->host_results[2 * m.count];
    base += kManyMax;
  } while (base < count);
  return 0;
}
int nka_hip_vec_dot_pair_many(nka_hip_vec_ws_t ws, int64_t n, const double *x0, const double *x1,
                              const double *const *ys, int32_t count, double *host_vals0, double *host_vals1,
                              double *host_cross) {
  return join_on_bad_pointers(ws, n, [&](int64_t n_) { return nka_hip_vec_dot_pair_many_entry(ws, n_, x0, x1, ys, count, host_vals0, host_vals1, host_cross); });
}

// z <- (a[j]*xs[j] + b[j]*ys[j]) + z for j = 0..count-1 in order; z is read and
// written once per group of kManyMax pairs.
int nka_hip_vec_update_many(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a, const double *const *xs,
                            const double *b, const double *const *ys, int32_t count) {
  if (!ws || n < 0 || count < 0 || (count > 0 && (!a || !b || !xs || !ys))) return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  if (n == 0 || count == 0) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  if (int rc = nka_detail::check_device_span(z, n, "vec_update_many: z")) return rc;
  for (int j = 0; j < count; j++) {
    if (int rc = nka_detail::check_device_span(xs[j], n, "vec_update_many: xs[j]")) return rc;
    if (int rc = nka_detail::check_device_span(ys[j], n, "vec_update_many: ys[j]")) return rc;
  }
  for (int base = 0; base < count; base += kManyMax) {
    ManyArgs m{};
    m.count = std::min(kManyMax, count - base);
    bool v2 = al16(z);
    for (int j = 0; j < m.count; j++) {
      m.x[j] = xs[base + j];
      m.y[j] = ys[base + j];
      m.a[j] = a[base + j];
      m.b[j] = b[base + j];
      v2 = v2 && al16(m.x[j]) && al16(m.y[j]);
    }
    const int nv = width_for(m.count);
    const int g = grid_for(ws, n, v2 ? 2 : 1, 2 * nv + 1);
#define LAUNCH2(NV) hipLaunchKernelGGL((k_update_many<NV, 2>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, m)
#define LAUNCH1(NV) hipLaunchKernelGGL((k_update_many<NV, 1>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, m)
    if (v2) { NKA_DISPATCH_NV(nv, LAUNCH2) } else { NKA_DISPATCH_NV(nv, LAUNCH1) }
#undef LAUNCH2
#undef LAUNCH1
    HIP_TRYV(hipGetLastError());
  }
  return 0;
}

// z <- a[j]*xs[j] + z for j = 0..count-1 in order; z read and written once per
// group of kManyMax vectors.
int nka_hip_vec_axpy_many(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a, const double *const *xs,
                          int32_t count) {
  if (!ws || n < 0 || count < 0 || (count > 0 && (!a || !xs))) return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  if (n == 0 || count == 0) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  if (int rc = nka_detail::check_device_span(z, n, "vec_axpy_many: z")) return rc;
  for (int j = 0; j < count; j++)
    if (int rc = nka_detail::check_device_span(xs[j], n, "vec_axpy_many: xs[j]")) return rc;
  for (int base = 0; base < count; base += kManyMax) {
    ManyArgs m{};
    m.count = std::min(kManyMax, count - base);
    bool v2 = al16(z);
    for (int j = 0; j < m.count; j++) {
      m.x[j] = xs[base + j];
      m.a[j] = a[base + j];
      v2 = v2 && al16(m.x[j]);
    }
    const int nv = width_for(m.count);
    const int g = grid_for(ws, n, v2 ? 2 : 1, nv + 1);
#define LAUNCH2(NV) hipLaunchKernelGGL((k_axpy_many<NV, 2>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, m)
#define LAUNCH1(NV) hipLaunchKernelGGL((k_axpy_many<NV, 1>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, m)
    if (v2) { NKA_DISPATCH_NV(nv, LAUNCH2) } else { NKA_DISPATCH_NV(nv, LAUNCH1) }
#undef LAUNCH2
#undef LAUNCH1
    HIP_TRYV(hipGetLastError());
  }
  return 0;
}

// ---- fused stages (overrides of vector%update_norm2 / scale_dot_pair_many /
// update_many_keep / axpy_many_keep, nka_amd/fortran/vector/vector_class.F90) ---------

// *host_norm = ||a*x + z||_2 ; store != 0: z <- a*x + z, else z is left as it is (the
// caller applies the update in the next stage, nka_hip_vec_scale_dot_pair_many with pre).
static int nka_hip_vec_update_norm2_entry(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x, int32_t store,
                             double *host_norm) {
  if (ws && ws->sum_order == 1)     // (the vector types fall back to the deferred hooks then: dot(), whose sum IS ordered)
    return nka_detail::set_error(NKA_HIP_ESTATE, "this batched reduction sums in blocks; with reference-order sums "
                                 "(nka_hip_vec_set_sum_order) the vector type must use dot() / norm2()");
  if (!ws || !host_norm || n < 0) return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  *host_norm = 0.0;
  if (n == 0 && !ws_parallel(ws)) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  if (n == 0) {                        // empty slice: zero, but the collective is still joined
    if (int rc = fetch_sums(ws, 1, 1, 1, 1, 0, false)) return rc;
    *host_norm = std::sqrt(ws->host_results[0]);
    return 0;
  }
  if (int rc = nka_detail::check_device_span(z, n, "vec_update_norm2: z")) return rc;
  if (int rc = nka_detail::check_device_span(x, n, "vec_update_norm2: x")) return rc;
  const bool v2 = al16(z) && al16(x);
  const int g = grid_for(ws, n, v2 ? 2 : 1);
  if (v2 && store)
    hipLaunchKernelGGL((k_update_norm2<2, true>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, x, a, ws->partials);
  else if (v2)
    hipLaunchKernelGGL((k_update_norm2<2, false>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, x, a, ws->partials);
  else if (store)
    hipLaunchKernelGGL((k_update_norm2<1, true>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, x, a, ws->partials);
  else
    hipLaunchKernelGGL((k_update_norm2<1, false>), dim3(g), dim3(kBlock), 0, ws->stream, n, z, x, a, ws->partials);
  HIP_TRYV(hipGetLastError());
  if (int rc = fetch_sums(ws, g, 1, 1, 1, 0, true)) return rc;
  *host_norm = std::sqrt(ws->host_results[0]);     // the square root of the GLOBAL sum
  return 0;
}
int nka_hip_vec_update_norm2(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x, int32_t store,
                             double *host_norm) {
  return join_on_bad_pointers(ws, n, [&](int64_t n_) { return nka_hip_vec_update_norm2_entry(ws, n_, z, a, x, store, host_norm); });
}

// [pre != 0: w <- pre_a*f + w ;] w <- a*w ; v <- a*v (subtract != 0: then v <- (-1)*w + v) ;
// with the new w: vals_w[j] = <w, ys[j]>, vals_f[j] = <f, ys[j]>, *cross = <f, w>.  One pass
// when count <= 24; longer lists scale in the first launch and only add dots after it.
static int scale_dot_pair_many_impl(nka_hip_vec_ws_t ws, int64_t n, double *w, double *v, double a, int32_t subtract,
                                    int32_t pre, double pre_a, const double *f, const double *const *ys, int32_t count,
                                    double *host_vals_w, double *host_vals_f, double *host_cross, int store,
                                    double *host_dd = nullptr);

static int nka_hip_vec_scale_dot_pair_many_entry(nka_hip_vec_ws_t ws, int64_t n, double *w, double *v, double a, int32_t subtract,
                                    int32_t pre, double pre_a, const double *f, const double *const *ys, int32_t count,
                                    double *host_vals_w, double *host_vals_f, double *host_cross) {
  if (ws && ws->sum_order == 1)     // (the vector types fall back to the deferred hooks then: dot(), whose sum IS ordered)
    return nka_detail::set_error(NKA_HIP_ESTATE, "this batched reduction sums in blocks; with reference-order sums "
                                 "(nka_hip_vec_set_sum_order) the vector type must use dot() / norm2()");
  return scale_dot_pair_many_impl(ws, n, w, v, a, subtract, pre, pre_a, f, ys, count, host_vals_w, host_vals_f, host_cross, 1);
}
int nka_hip_vec_scale_dot_pair_many(nka_hip_vec_ws_t ws, int64_t n, double *w, double *v, double a, int32_t subtract,
                                    int32_t pre, double pre_a, const double *f, const double *const *ys, int32_t count,
                                    double *host_vals_w, double *host_vals_f, double *host_cross) {
  return join_on_bad_pointers(ws, n, [&](int64_t n_) { return nka_hip_vec_scale_dot_pair_many_entry(ws, n_, w, v, a, subtract, pre, pre_a, f, ys, count, host_vals_w, host_vals_f, host_cross); });
}

// The same sums with NOTHING stored: w stays as it is (raw, or already differenced when pre == 0),
// v is not touched; the caller hands (w, v, a, pre, pre_a) to nka_hip_vec_update_many_keep_pend /
// nka_hip_vec_axpy_many_keep_pend, which normalise the pair while they combine.  count <= 24.
static int nka_hip_vec_dot_pair_many_scaled_entry(nka_hip_vec_ws_t ws, int64_t n, const double *w, double a, int32_t pre, double pre_a,
                                     const double *f, const double *const *ys, int32_t count, double *host_vals_w,
                                     double *host_vals_f, double *host_cross) {
  if (ws && ws->sum_order == 1)     // (the vector types fall back to the deferred hooks then: dot(), whose sum IS ordered)
    return nka_detail::set_error(NKA_HIP_ESTATE, "this batched reduction sums in blocks; with reference-order sums "
                                 "(nka_hip_vec_set_sum_order) the vector type must use dot() / norm2()");
  // (a list longer than one launch: balanced groups, each forming w' in registers again -- a pure-read stage, nothing to undo;
  //  <f,w'> comes from the first group)
  const int ngroups = many_groups(count);
  for (int grp = 0, base = 0; grp < ngroups; grp++) {
    const int wdt = many_group_width(count, grp);
    double cross = 0.0;
    if (int rc = scale_dot_pair_many_impl(ws, n, const_cast<double *>(w), const_cast<double *>(w), a, 0, pre, pre_a, f, ys + base, wdt,
                                          host_vals_w + base, host_vals_f + base, &cross, 0)) return rc;
    if (grp == 0 && host_cross) *host_cross = cross;
    base += wdt;
  }
  return 0;
}
int nka_hip_vec_dot_pair_many_scaled(nka_hip_vec_ws_t ws, int64_t n, const double *w, double a, int32_t pre, double pre_a,
                                     const double *f, const double *const *ys, int32_t count, double *host_vals_w,
                                     double *host_vals_f, double *host_cross) {
  return join_on_bad_pointers(ws, n, [&](int64_t n_) { return nka_hip_vec_dot_pair_many_scaled_entry(ws, n_, w, a, pre, pre_a, f, ys, count, host_vals_w, host_vals_f, host_cross); });
}

// The norm stage and the scale-and-dot stage as ONE pure-read pass: with d = a*x + z (nothing stored),
//   *host_dd = <d,d>, vals_z[j] = <d, ys[j]>, vals_x[j] = <x, ys[j]>, *cross = <x, d>   -- the RAW sums; the caller
// takes s = sqrt(<d,d>) and scales by 1/s itself (what the array flavour's pass PA does: the Gram row of the
// normalised pair as fl(<d,w_k>/s) instead of the sum of fl(d_i/s)*w_k,i -- last-bit differences).
// count <= 24.
static int nka_hip_vec_diff_norm_dot_pair_many_entry(nka_hip_vec_ws_t ws, int64_t n, const double *z, double a, const double *x,
                                        const double *const *ys, int32_t count, double *host_dd, double *host_vals_z,
                                        double *host_vals_x, double *host_cross) {
  if (ws && ws->sum_order == 1)     // (the vector types fall back to the deferred hooks then: dot(), whose sum IS ordered)
    return nka_detail::set_error(NKA_HIP_ESTATE, "this batched reduction sums in blocks; with reference-order sums "
                                 "(nka_hip_vec_set_sum_order) the vector type must use dot() / norm2()");
  if (!host_dd) return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  // (a list longer than one launch: balanced groups, d = a*x + z formed in registers again by each -- pure read; <d,d> and
  //  <x,d> come from the first group)
  *host_dd = 0.0;
  const int ngroups = many_groups(count);
  for (int grp = 0, base = 0; grp < ngroups; grp++) {
    const int wdt = many_group_width(count, grp);
    double cross = 0.0, dd = 0.0;
    if (int rc = scale_dot_pair_many_impl(ws, n, const_cast<double *>(z), const_cast<double *>(z), 1.0, 0, 1, a, x, ys + base, wdt,
                                          host_vals_z + base, host_vals_x + base, &cross, 0, &dd)) return rc;
    if (grp == 0) {
      *host_dd = dd;
      if (host_cross) *host_cross = cross;
    }
    base += wdt;
  }
  return 0;
}
int nka_hip_vec_diff_norm_dot_pair_many(nka_hip_vec_ws_t ws, int64_t n, const double *z, double a, const double *x,
                                        const double *const *ys, int32_t count, double *host_dd, double *host_vals_z,
                                        double *host_vals_x, double *host_cross) {
  return join_on_bad_pointers(ws, n, [&](int64_t n_) { return nka_hip_vec_diff_norm_dot_pair_many_entry(ws, n_, z, a, x, ys, count, host_dd, host_vals_z, host_vals_x, host_cross); });
}

static int scale_dot_pair_many_impl(nka_hip_vec_ws_t ws, int64_t n, double *w, double *v, double a, int32_t subtract,
                                    int32_t pre, double pre_a, const double *f, const double *const *ys, int32_t count,
                                    double *host_vals_w, double *host_vals_f, double *host_cross, int store,
                                    double *host_dd) {
  if (!ws || n < 0 || count < 0 || !host_cross || (count > 0 && (!ys || !host_vals_w || !host_vals_f)))
    return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  const int extra = host_dd ? 2 : 1;
  *host_cross = 0.0;
  for (int j = 0; j < count; j++) host_vals_w[j] = host_vals_f[j] = 0.0;
  if (n == 0 && !ws_parallel(ws)) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  if (n == 0) {                        // empty slice: zeros, but every collective is still joined
    const int c0 = std::min(kManyMax, count);
    if (int rc = fetch_sums(ws, 1, 2, std::max(c0, 1), c0, extra, false)) return rc;
    for (int j = 0; j < c0; j++) {
      host_vals_w[j] = ws->host_results[j];
      host_vals_f[j] = ws->host_results[c0 + j];
    }
    *host_cross = ws->host_results[2 * c0];
    if (host_dd) *host_dd = ws->host_results[2 * c0 + 1];
    if (count > kManyMax) {
      double cross_again = 0.0;
      return nka_hip_vec_dot_pair_many(ws, n, w, f, ys + kManyMax, count - kManyMax, host_vals_w + kManyMax,
                                       host_vals_f + kManyMax, &cross_again);
    }
    return 0;
  }
  if (int rc = nka_detail::check_device_span(w, n, "vec_scale_dot_pair_many: w")) return rc;
  if (int rc = nka_detail::check_device_span(v, n, "vec_scale_dot_pair_many: v")) return rc;
  if (int rc = nka_detail::check_device_span(f, n, "vec_scale_dot_pair_many: f")) return rc;
  for (int j = 0; j < count; j++)
    if (int rc = nka_detail::check_device_span(ys[j], n, "vec_scale_dot_pair_many: ys[j]")) return rc;
  {
    ManyArgs m{};
    m.count = std::min(kManyMax, count);
    bool v2 = al16(w) && al16(v) && al16(f);
    for (int j = 0; j < m.count; j++) {
      m.x[j] = ys[j];
      v2 = v2 && al16(m.x[j]);
    }
    const bool win = v2;                                                    // 16-byte path = the rolling-window kernel
    const int nv = win ? std::max(m.count, 1) : width_for(m.count);      // window kernel: exact width, no padding
    const int g = grid_for(ws, n, v2 ? 2 : 1, win ? 22 : nv + 3);   // rolling-window kernel: one block per CU
#define NKA_SDPM(NV, VEC, SUB, PRE)                                                                              \
  hipLaunchKernelGGL((k_scale_dot_pair_many<NV, VEC, SUB, PRE>), dim3(g), dim3(kBlock), 0, ws->stream, n, w, v, a, pre_a, f, m, \
                     ws->partials, store)
#define NKA_SDPMW(NV, SUB, PRE)                                                                                  \
  hipLaunchKernelGGL((k_scale_dot_pair_many_win<NV, SUB, PRE>), dim3(g), dim3(kBlock), 0, ws->stream, n, w, v, a, pre_a, f, m, \
                     ws->partials, store)
#define LWSP(NV) NKA_SDPMW(NV, true, true)
#define LWSN(NV) NKA_SDPMW(NV, true, false)
#define LWNP(NV) NKA_SDPMW(NV, false, true)
#define LWNN(NV) NKA_SDPMW(NV, false, false)
#define L1SP(NV) NKA_SDPM(NV, 1, true, true)
#define L1SN(NV) NKA_SDPM(NV, 1, true, false)
#define L1NP(NV) NKA_SDPM(NV, 1, false, true)
#define L1NN(NV) NKA_SDPM(NV, 1, false, false)
#define LWDD(NV) hipLaunchKernelGGL((k_scale_dot_pair_many_win<NV, false, true, true>), dim3(g), dim3(kBlock), 0, ws->stream, n, w, v, a, pre_a, f, m, ws->partials, store)
#define L1DD(NV) hipLaunchKernelGGL((k_scale_dot_pair_many<NV, 1, false, true, true>), dim3(g), dim3(kBlock), 0, ws->stream, n, w, v, a, pre_a, f, m, ws->partials, store)
    if (host_dd) {                       // norm + both rows in one pure-read pass (SUB = false, PRE = true, nothing stored)
      if (win) { NKA_DISPATCH_EXACT(nv, LWDD) } else { NKA_DISPATCH_NV(nv, L1DD) }
    } else if (win) {
      if (subtract) { if (pre) { NKA_DISPATCH_EXACT(nv, LWSP) } else { NKA_DISPATCH_EXACT(nv, LWSN) } }
      else          { if (pre) { NKA_DISPATCH_EXACT(nv, LWNP) } else { NKA_DISPATCH_EXACT(nv, LWNN) } }
    } else {
      if (subtract) { if (pre) { NKA_DISPATCH_NV(nv, L1SP) } else { NKA_DISPATCH_NV(nv, L1SN) } }
      else          { if (pre) { NKA_DISPATCH_NV(nv, L1NP) } else { NKA_DISPATCH_NV(nv, L1NN) } }
    }
#undef LWSP
#undef LWSN
#undef LWNP
#undef LWNN
#undef L1SP
#undef L1SN
#undef L1NP
#undef L1NN
#undef NKA_SDPM
#undef NKA_SDPMW
#undef LWDD
#undef L1DD
    HIP_TRYV(hipGetLastError());
    if (int rc = fetch_sums(ws, g, 2, nv, m.count, extra, true)) return rc;
    for (int j = 0; j < m.count; j++) {
      host_vals_w[j] = ws->host_results[j];
      host_vals_f[j] = ws->host_results[m.count + j];
    }
    *host_cross = ws->host_results[2 * m.count];
    if (host_dd) *host_dd = ws->host_results[2 * m.count + 1];
  }
  if (count > kManyMax) {   // the rest of a long list: plain two-row dots against the already scaled w
    double cross_again = 0.0;
    return nka_hip_vec_dot_pair_many(ws, n, w, f, ys + kManyMax, count - kManyMax, host_vals_w + kManyMax,
                                     host_vals_f + kManyMax, &cross_again);
  }
  return 0;
}

// keep_in <- z ; z <- (a[j]*xs[j] + b[j]*ys[j]) + z, j in order ; keep_out <- z.
int nka_hip_vec_update_many_keep(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a, const double *const *xs,
                                 const double *b, const double *const *ys, int32_t count, double *keep_in,
                                 double *keep_out) {
  return update_many_keep<true>(ws, n, z, a, xs, b, ys, count, keep_in, keep_out, "vec_update_many_keep");
}

// keep_in <- z ; z <- a[j]*xs[j] + z, j in order ; keep_out <- z.
int nka_hip_vec_axpy_many_keep(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a, const double *const *xs,
                               int32_t count, double *keep_in, double *keep_out) {
  return update_many_keep<false>(ws, n, z, a, xs, nullptr, nullptr, count, keep_in, keep_out, "vec_axpy_many_keep");
}

// The same with entry 0 of the lists being the RAW new pair, normalised on the way (see Pend): after
// nka_hip_vec_dot_pair_many_scaled, which left it untouched.  xs[0] = w1, ys[0] = v1 are rewritten with
//   w1 <- a*(pre_a*z_in + w1)  [pre != 0; else a*w1] ,  v1 <- a*v1 ,  [subtract != 0: v1 <- (-1)*w1 + v1].
int nka_hip_vec_update_many_keep_pend(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a, const double *const *xs,
                                      const double *b, const double *const *ys, int32_t count, double *keep_in,
                                      double *keep_out, double pend_a, int32_t pend_pre, double pend_pre_a,
                                      int32_t pend_subtract) {
  Pend pd;
  pd.a = pend_a;
  pd.pre_a = pend_pre_a;
  pd.flags = 1 | (pend_pre ? 2 : 0) | (pend_subtract ? 4 : 0);
  return update_many_keep<true>(ws, n, z, a, xs, b, ys, count, keep_in, keep_out, "vec_update_many_keep_pend", pd);
}

// Compact storage (the v slots hold v - w): xs[0] = v1 raw, pend_w = w1 raw; rewritten with
//   w1 <- a*(pre_a*z_in + w1)  [pre != 0; else a*w1] ,  v1 <- (-1)*w1 + a*v1 .
int nka_hip_vec_axpy_many_keep_pend(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a, const double *const *xs,
                                    int32_t count, double *keep_in, double *keep_out, double *pend_w, double pend_a,
                                    int32_t pend_pre, double pend_pre_a) {
  Pend pd;
  pd.w = pend_w;
  pd.a = pend_a;
  pd.pre_a = pend_pre_a;
  pd.flags = 1 | (pend_pre ? 2 : 0) | 4;
  return update_many_keep<false>(ws, n, z, a, xs, nullptr, nullptr, count, keep_in, keep_out, "vec_axpy_many_keep_pend", pd);
}

// ---- parallel-aware reductions: hooks on the workspace (include/nka_hip.h) ------------------
int nka_hip_vec_set_sum_order(nka_hip_vec_ws_t ws, int32_t order) {
  if (!ws) return nka_detail::set_error(NKA_HIP_EINVAL, "null workspace");
  if (order != NKA_HIP_SUMS_REFERENCE_ORDER && order != NKA_HIP_SUMS_BLOCKED && order != NKA_HIP_SUMS_AUTO &&
      order != NKA_HIP_SUMS_BLOCKED_ROUNDED)
    return nka_detail::set_error(NKA_HIP_EINVAL, "vec_set_sum_order: NKA_HIP_SUMS_REFERENCE_ORDER, _BLOCKED (or _AUTO = blocked), _BLOCKED_ROUNDED");
  // (3 = blocked sums, but the vector types keep the norm stage a pass of its own, so that the Gram row is summed on the ROUNDED
  //  pair -- the reductions of this library sum in blocks either way: only the vector types read the difference)
  ws->sum_order = order == NKA_HIP_SUMS_REFERENCE_ORDER ? 1 : order == NKA_HIP_SUMS_BLOCKED_ROUNDED ? 3 : 0;
  return 0;
}
int nka_hip_vec_get_sum_order(nka_hip_vec_ws_t ws) {
  if (!ws) return nka_detail::set_error(NKA_HIP_EINVAL, "null workspace");
  return ws->sum_order == 1 ? NKA_HIP_SUMS_REFERENCE_ORDER : ws->sum_order == 3 ? NKA_HIP_SUMS_BLOCKED_ROUNDED : NKA_HIP_SUMS_BLOCKED;
}

int nka_hip_vec_set_allreduce(nka_hip_vec_ws_t ws, nka_hip_allreduce_fn fn, void *ctx) {
  if (!ws) return nka_detail::set_error(NKA_HIP_EINVAL, "null workspace");
  ws->allreduce = fn;
  ws->allreduce_ctx = fn ? ctx : nullptr;
  return 0;
}

int nka_hip_vec_set_host_allreduce(nka_hip_vec_ws_t ws, nka_hip_host_allreduce_fn fn, void *ctx) {
  if (!ws) return nka_detail::set_error(NKA_HIP_EINVAL, "null workspace");
  ws->host_allreduce = fn;
  ws->host_allreduce_ctx = fn ? ctx : nullptr;
  return 0;
}

int nka_hip_vec_comm_init_rank(nka_hip_vec_ws_t ws, const void *id128, int32_t nranks, int32_t rank) {
  if (!ws || !id128) return nka_detail::set_error(NKA_HIP_EINVAL, "null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return nka_detail::set_error(NKA_HIP_EINVAL, "bad rank / nranks");
  const auto &R = nka_detail::rccl();
  if (!R.ok()) return nka_detail::set_error(NKA_HIP_ECOMM, R.err);
  HIP_TRYV(hipSetDevice(ws->device));
  if (int rc = nka_hip_vec_comm_destroy(ws)) return rc;
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  memcpy(&id, id128, sizeof id);
  ncclResult_t r = R.CommInitRank(&ws->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    ws->comm = nullptr;
    return nka_detail::set_error(NKA_HIP_ECOMM, std::string("ncclCommInitRank: ") + R.GetErrorString(r));
  }
  ws->allreduce = rccl_vec_allreduce;
  ws->allreduce_ctx = ws;
  return 0;
}

int nka_hip_vec_comm_destroy(nka_hip_vec_ws_t ws) {
  if (!ws) return nka_detail::set_error(NKA_HIP_EINVAL, "null workspace");
  if (ws->comm) {
    HIP_TRYV(hipSetDevice(ws->device));
    HIP_TRYV(hipStreamSynchronize(ws->stream));
    nka_detail::rccl().CommDestroy(ws->comm);
    ws->comm = nullptr;
  }
  if (ws->allreduce == rccl_vec_allreduce) {
    ws->allreduce = nullptr;
    ws->allreduce_ctx = nullptr;
  }
  return 0;
}

int nka_hip_vec_allreduce_now(nka_hip_vec_ws_t ws, double *host_vals, int32_t count) {
  if (!ws || count < 0 || count > 2 * kManyMax + 2 || (count > 0 && !host_vals))
    return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  if (count == 0) return 0;
  HIP_TRYV(hipSetDevice(ws->device));
  if (ws->allreduce) {
    HIP_TRYV(hipMemcpyAsync(ws->red_dev, host_vals, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ws->stream));
    if (int rc = ws->allreduce(ws->allreduce_ctx, ws->red_dev, count, ws->stream))
      return rc < 0 ? rc : nka_detail::set_error(NKA_HIP_ECOMM, "vector reduction: the all-reduce hook failed");
    HIP_TRYV(hipMemcpyAsync(ws->host_results, ws->red_dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ws->stream));
    HIP_TRYV(hipStreamSynchronize(ws->stream));
    for (int i = 0; i < count; i++) host_vals[i] = ws->host_results[i];
  }
  return run_host_hook(ws, host_vals, count);
}

int nka_hip_vec_h2d(nka_hip_vec_ws_t ws, int64_t n, double *dst_dev, const double *src_host) {
  if (!ws || n < 0) return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  HIP_TRYV(hipSetDevice(ws->device));
  HIP_TRYV(hipMemcpyAsync(dst_dev, src_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ws->stream));
  HIP_TRYV(hipStreamSynchronize(ws->stream));
  return 0;
}

int nka_hip_vec_d2h(nka_hip_vec_ws_t ws, int64_t n, double *dst_host, const double *src_dev) {
  if (!ws || n < 0) return nka_detail::set_error(NKA_HIP_EINVAL, "bad argument");
  HIP_TRYV(hipSetDevice(ws->device));
  HIP_TRYV(hipMemcpyAsync(dst_host, src_dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ws->stream));
  HIP_TRYV(hipStreamSynchronize(ws->stream));
  return 0;
}

}  // extern "C"
