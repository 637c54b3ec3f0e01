// tools/hbm_probe.hip -- what HBM rate can a streaming kernel reach on this chip?
// Measurement aid (not part of the product).  Reads S streams of `n` doubles
// (separate slots of one big allocation, like the NKA slot storage) with
// 16-B/lane loads and sums them; optional W streams written.  Variants:
// plain vs non-temporal loads/stores, tiles per iteration (loads in flight),
// grid size.  Prints GB/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ double2 ld2(const double *p) {
  if (NT) {
    double2 r;
    r.x = __builtin_nontemporal_load(p);
    r.y = __builtin_nontemporal_load(p + 1);
    return r;
  }
  return *reinterpret_cast<const double2 *>(p);
}
typedef double nd2 __attribute__((ext_vector_type(2)));
// store policy: 0 plain, 1 nt (builtin), 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt
template <int POL> __device__ __forceinline__ void st2(double *p, double2 v) {
  if (POL == 1) {
    nd2 q = {v.x, v.y};
    __builtin_nontemporal_store(q, reinterpret_cast<nd2 *>(p));   // one global_store_dwordx4 ... nt (two scalar nt stores merge into a PLAIN x4)
  } else if (POL == 5) {   // sc1 as a compiler-issued buffer store (counted by its s_waitcnt bookkeeping)
    typedef unsigned pu4 __attribute__((ext_vector_type(4)));
    nd2 q = {v.x, v.y};
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 16, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu4, q), r, 0, 0, 16);
  } else if (POL == 2) {
    nd2 q = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
  } else if (POL == 3) {
    nd2 q = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(q) : "memory");
  } else if (POL == 4) {
    nd2 q = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(q) : "memory");
  } else {
    *reinterpret_cast<double2 *>(p) = v;
  }
}

// S read streams, W write streams, T tiles (of 256 x 16 B) per block iteration.
template <int S, int W, int T, bool NTL, int NTS, int MAP>
__global__ __launch_bounds__(256) void k_stream(const double *base, double *wbase, size_t stride, size_t n, double *out) {
  const size_t ntile = n / 512 / T;
  double acc = 0.0;
  const size_t per = (ntile + gridDim.x - 1) / gridDim.x;
  const size_t tb = MAP ? blockIdx.x * per : blockIdx.x, te = MAP ? (tb + per < ntile ? tb + per : ntile) : ntile;
  const size_t step = MAP ? 1 : gridDim.x;
  for (size_t t = tb; t < te; t += step) {
    double2 v[S > 0 ? S : 1][T];
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
      for (int q = 0; q < T; q++) v[s][q] = ld2<NTL>(base + s * stride + (t * T + q) * 512 + threadIdx.x * 2);
    double2 sum = {0.0, 0.0};
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
      for (int q = 0; q < T; q++) { sum.x += v[s][q].x; sum.y += v[s][q].y; }
    acc += sum.x + sum.y;
#pragma unroll
    for (int w = 0; w < W; w++)
#pragma unroll
      for (int q = 0; q < T; q++) st2<NTS>(wbase + w * stride + (t * T + q) * 512 + threadIdx.x * 2, sum);
  }
  if (S > 0 && acc == 12345.678) out[0] = acc;  // keep the loads alive
}

// Pure-read variant with the arithmetic of the NKA dot pass: per element, stream 0 (f) and
// stream 1 (w1) give fq and d = w1 - f, every other stream x adds TWO fp64 FMAs
// (acc_d += d*x, acc_f += fq*x) into its own pair of accumulators, as k_dots does.
template <int S>
__global__ __launch_bounds__(256) void k_stream_fma(const double *base, size_t stride, size_t n, double *out) {
  const size_t ntile = n / 512;
  double acc[2 * S];
#pragma unroll
  for (int a = 0; a < 2 * S; a++) acc[a] = 0.0;
  for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
    double2 v[S];
#pragma unroll
    for (int s = 0; s < S; s++) v[s] = ld2<true>(base + s * stride + t * 512 + threadIdx.x * 2);
    const double fx = v[0].x, fy = v[0].y, dx = v[1].x - fx, dy = v[1].y - fy;
    acc[0] = fma(dx, dx, acc[0]); acc[0] = fma(dy, dy, acc[0]);
    acc[1] = fma(fx, dx, acc[1]); acc[1] = fma(fy, dy, acc[1]);
#pragma unroll
    for (int s = 2; s < S; s++) {
      acc[2 * s] = fma(dx, v[s].x, acc[2 * s]);     acc[2 * s] = fma(dy, v[s].y, acc[2 * s]);
      acc[2 * s + 1] = fma(fx, v[s].x, acc[2 * s + 1]); acc[2 * s + 1] = fma(fy, v[s].y, acc[2 * s + 1]);
    }
  }
  double tot = 0.0;
#pragma unroll
  for (int a = 0; a < 2 * S; a++) tot += acc[a];
  if (tot == 12345.678) out[0] = tot;
}

// Mixed read/write with a SMALL ROLLING WINDOW of loads: streams 0,1 are loaded a tile ahead,
// streams 2..S-1 go through a ring of WIN registers (a consumed slot is re-loaded at once, across
// the tile boundary), then W streams are stored per tile -- the shape of the NKA combine pass.
template <int S, int W, int WIN, int NTS>
__global__ __launch_bounds__(256) void k_stream_win(const double *base, double *wbase, size_t stride, size_t n, double *out) {
  constexpr int R = S - 2;
  static_assert(R % WIN == 0, "ring must divide the ringed streams");
  const size_t ntile = n / 512;
  double acc = 0.0;
  double2 a0, a1, ring[WIN];
  size_t t = blockIdx.x;
  if (t < ntile) {
    a0 = ld2<true>(base + t * 512 + threadIdx.x * 2);
    a1 = ld2<true>(base + stride + t * 512 + threadIdx.x * 2);
#pragma unroll
    for (int j = 0; j < WIN; j++) ring[j] = ld2<true>(base + (2 + j) * stride + t * 512 + threadIdx.x * 2);
  }
  for (; t < ntile; t += gridDim.x) {
    const size_t tn = (t + gridDim.x < ntile) ? t + gridDim.x : t;
    double2 sum = {a0.x + a1.x, a0.y + a1.y};
    __builtin_amdgcn_sched_barrier(0);
    a0 = ld2<true>(base + tn * 512 + threadIdx.x * 2);
    a1 = ld2<true>(base + stride + tn * 512 + threadIdx.x * 2);
#pragma unroll
    for (int j = 0; j < R; j++) {
      const double2 x = ring[j % WIN];
      __builtin_amdgcn_sched_barrier(0);
      if (j + WIN < R) ring[j % WIN] = ld2<true>(base + (2 + j + WIN) * stride + t * 512 + threadIdx.x * 2);
      else ring[j % WIN] = ld2<true>(base + (2 + j + WIN - R) * stride + tn * 512 + threadIdx.x * 2);
      __builtin_amdgcn_sched_barrier(0);
      sum.x += x.x; sum.y += x.y;
    }
    acc += sum.x + sum.y;
#pragma unroll
    for (int w = 0; w < W; w++) st2<NTS>(wbase + w * stride + t * 512 + threadIdx.x * 2, sum);
  }
  if (acc == 12345.678) out[0] = acc;
}

template <int S, int W, int WIN, int NTS>
void runw(const char *name, const double *base, double *wbase, size_t stride, size_t n, double *out, int grid) {
  static_assert(S <= 42 && W <= 6, "variant exceeds the allocated slots");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 3;
  hipLaunchKernelGGL((k_stream_win<S, W, WIN, NTS>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((k_stream_win<S, W, WIN, NTS>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)(S + W) * n * 8.0 * reps;
  printf("%-44s grid %5d  %8.1f GB/s  (%.3f ms/launch)\n", name, grid, bytes / (ms * 1e-3) / 1e9, ms / reps);
  fflush(stdout);
}


// ---- mode "d": does a STATIC tile->block mapping lose time in the tail? ----------------
// The rolling-window kernel again, (a) with the start / end time of every block stamped
// (wall_clock64, 100 MHz) so the spread of block end times can be read, (b) with DYNAMIC
// chunk claiming: a block takes its first chunk by blockIdx and every further chunk of
// `chunk_tiles` tiles from a global counter (claimed one chunk ahead by thread 0, so the
// rolling window can prefetch across the chunk boundary; one workgroup barrier per chunk).
template <int S, int W, int WIN, int NTS, bool DYN>
__global__ __launch_bounds__(256) void k_stream_win_d(const double *base, double *wbase, size_t stride, size_t n, double *out,
                                                      unsigned *counter, unsigned chunk_tiles, unsigned long long *stamps) {
  constexpr int R = S - 2;
  static_assert(R % WIN == 0, "ring must divide the ringed streams");
  const size_t ntile = n / 512;
  const unsigned nchunk = (unsigned)((ntile + chunk_tiles - 1) / chunk_tiles);
  __shared__ unsigned s_next[2];
  const unsigned long long t_start = wall_clock64();
  double acc = 0.0;
  double2 a0, a1, ring[WIN];
  unsigned c = blockIdx.x, par = 0;
  if (c < nchunk) {
    const size_t t = (size_t)c * chunk_tiles;
    a0 = ld2<true>(base + t * 512 + threadIdx.x * 2);
    a1 = ld2<true>(base + stride + t * 512 + threadIdx.x * 2);
#pragma unroll
    for (int j = 0; j < WIN; j++) ring[j] = ld2<true>(base + (2 + j) * stride + t * 512 + threadIdx.x * 2);
  }
  while (c < nchunk) {
    if (threadIdx.x == 0) s_next[par] = DYN ? atomicAdd(counter, 1u) : c + gridDim.x;
    const size_t t0 = (size_t)c * chunk_tiles;
    const size_t tend = (t0 + chunk_tiles < ntile) ? t0 + chunk_tiles : ntile;
    unsigned cn = 0xffffffffu;
    for (size_t t = t0; t < tend; ++t) {
      size_t tn = t + 1;
      if (tn >= tend) {                      // last tile of the chunk: learn the next chunk
        __syncthreads();
        cn = s_next[par];
        tn = (cn < nchunk) ? (size_t)cn * chunk_tiles : t;
      }
      double2 sum = {a0.x + a1.x, a0.y + a1.y};
      __builtin_amdgcn_sched_barrier(0);
      a0 = ld2<true>(base + tn * 512 + threadIdx.x * 2);
      a1 = ld2<true>(base + stride + tn * 512 + threadIdx.x * 2);
#pragma unroll
      for (int j = 0; j < R; j++) {
        const double2 x = ring[j % WIN];
        __builtin_amdgcn_sched_barrier(0);
        if (j + WIN < R) ring[j % WIN] = ld2<true>(base + (2 + j + WIN) * stride + t * 512 + threadIdx.x * 2);
        else ring[j % WIN] = ld2<true>(base + (2 + j + WIN - R) * stride + tn * 512 + threadIdx.x * 2);
        __builtin_amdgcn_sched_barrier(0);
        sum.x += x.x; sum.y += x.y;
      }
      acc += sum.x + sum.y;
#pragma unroll
      for (int w = 0; w < W; w++) st2<NTS>(wbase + w * stride + t * 512 + threadIdx.x * 2, sum);
    }
    c = cn;
    par ^= 1;
  }
  if (acc == 12345.678) out[0] = acc;
  if (stamps && threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t_start;
    stamps[2 * blockIdx.x + 1] = wall_clock64();
  }
}

template <int S, int W, int WIN, int NTS, bool DYN>
void rund(const char *name, const double *base, double *wbase, size_t stride, size_t n, double *out, int grid, unsigned chunk_tiles) {
  static_assert(S <= 42 && W <= 6, "variant exceeds the allocated slots");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 5;
  unsigned *ctr;
  unsigned long long *stamps;
  CK(hipMalloc(&ctr, sizeof(unsigned) * (reps + 1)));
  CK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * grid));
  std::vector<unsigned> init(reps + 1, (unsigned)grid);
  CK(hipMemcpy(ctr, init.data(), sizeof(unsigned) * (reps + 1), hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_stream_win_d<S, W, WIN, NTS, DYN>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out, ctr + reps, chunk_tiles, stamps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((k_stream_win_d<S, W, WIN, NTS, DYN>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out, ctr + r, chunk_tiles, stamps);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * grid);
  CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
  unsigned long long s0 = ~0ull;
  std::vector<double> ends(grid);
  for (int b = 0; b < grid; b++) s0 = h[2 * b] < s0 ? h[2 * b] : s0;
  for (int b = 0; b < grid; b++) ends[b] = (double)(h[2 * b + 1] - s0) * 0.01;   // us at 100 MHz
  std::sort(ends.begin(), ends.end());
  const double bytes = (double)(S + W) * n * 8.0 * reps;
  printf("%-40s chunk %4u grid %4d  %8.1f GB/s  (%.4f ms/launch)  block ends us: min %.1f p50 %.1f p90 %.1f max %.1f\n", name, chunk_tiles, grid,
         bytes / (ms * 1e-3) / 1e9, ms / reps, ends[0], ends[grid / 2], ends[grid * 9 / 10], ends[grid - 1]);
  fflush(stdout);
  CK(hipFree(ctr));
  CK(hipFree(stamps));
}

// Dynamic claiming with the ticket requested ONE CHUNK AHEAD: chunks 0 and 1 of a block are static
// (b, b + G); at the start of chunk i thread 0 requests the chunk after next (returning atomic, the
// value stays in a register), publishes it in LDS at the END of the chunk, one barrier per chunk.
// The atomic's latency hides behind a whole chunk of streaming.
template <int S, int W, int WIN, int NTS>
__global__ __launch_bounds__(256) void k_stream_win_d2(const double *base, double *wbase, size_t stride, size_t n, double *out,
                                                       unsigned *counter, unsigned chunk_tiles, unsigned long long *stamps, unsigned ng) {
  // ng counter groups (group = blockIdx % ng, one counter per group, 128 B apart): group g owns the chunks = g (mod ng)
  constexpr int R = S - 2;
  static_assert(R % WIN == 0, "ring must divide the ringed streams");
  const size_t ntile = n / 512;
  const unsigned nchunk = (unsigned)((ntile + chunk_tiles - 1) / chunk_tiles);
  __shared__ unsigned s_next[2];
  const unsigned long long t_start = wall_clock64();
  double acc = 0.0;
  double2 a0, a1, ring[WIN];
  const unsigned grp = blockIdx.x % ng;
  counter += grp * 32;
  unsigned c = blockIdx.x, cn = blockIdx.x + gridDim.x, par = 0;
  if (c < nchunk) {
    const size_t t = (size_t)c * chunk_tiles;
    a0 = ld2<true>(base + t * 512 + threadIdx.x * 2);
    a1 = ld2<true>(base + stride + t * 512 + threadIdx.x * 2);
#pragma unroll
    for (int j = 0; j < WIN; j++) ring[j] = ld2<true>(base + (2 + j) * stride + t * 512 + threadIdx.x * 2);
  }
  while (c < nchunk) {
    const bool claim = cn < nchunk;
    unsigned tk = 0xffffffffu;
    if (threadIdx.x == 0 && claim) tk = atomicAdd(counter, 1u) * ng + grp;   // a group's counter starts at 2G/ng
    const size_t t0 = (size_t)c * chunk_tiles;
    const size_t tend = (t0 + chunk_tiles < ntile) ? t0 + chunk_tiles : ntile;
    for (size_t t = t0; t < tend; ++t) {
      size_t tn = t + 1;
      if (tn >= tend) tn = claim ? (size_t)cn * chunk_tiles : t;
      double2 sum = {a0.x + a1.x, a0.y + a1.y};
      __builtin_amdgcn_sched_barrier(0);
      a0 = ld2<true>(base + tn * 512 + threadIdx.x * 2);
      a1 = ld2<true>(base + stride + tn * 512 + threadIdx.x * 2);
#pragma unroll
      for (int j = 0; j < R; j++) {
        const double2 x = ring[j % WIN];
        __builtin_amdgcn_sched_barrier(0);
        if (j + WIN < R) ring[j % WIN] = ld2<true>(base + (2 + j + WIN) * stride + t * 512 + threadIdx.x * 2);
        else ring[j % WIN] = ld2<true>(base + (2 + j + WIN - R) * stride + tn * 512 + threadIdx.x * 2);
        __builtin_amdgcn_sched_barrier(0);
        sum.x += x.x; sum.y += x.y;
      }
      acc += sum.x + sum.y;
#pragma unroll
      for (int w = 0; w < W; w++) st2<NTS>(wbase + w * stride + t * 512 + threadIdx.x * 2, sum);
    }
    if (threadIdx.x == 0) s_next[par] = tk;
    __syncthreads();
    c = cn;
    cn = s_next[par];
    par ^= 1;
  }
  if (acc == 12345.678) out[0] = acc;
  if (stamps && threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t_start;
    stamps[2 * blockIdx.x + 1] = wall_clock64();
  }
}

template <int S, int W, int WIN, int NTS>
void rund2(const char *name, const double *base, double *wbase, size_t stride, size_t n, double *out, int grid, unsigned chunk_tiles, unsigned ng = 1) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 5;
  const int CS = 32 * 16;     // counters of one launch: up to 16 groups, 128 B apart
  unsigned *ctr;
  unsigned long long *stamps;
  CK(hipMalloc(&ctr, sizeof(unsigned) * CS * (reps + 1)));
  CK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * grid));
  std::vector<unsigned> init(CS * (reps + 1), (unsigned)(2 * grid / ng));
  CK(hipMemcpy(ctr, init.data(), sizeof(unsigned) * init.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_stream_win_d2<S, W, WIN, NTS>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out, ctr + reps * CS, chunk_tiles, stamps, ng);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((k_stream_win_d2<S, W, WIN, NTS>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out, ctr + r * CS, chunk_tiles, stamps, ng);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * grid);
  CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
  unsigned long long s0 = ~0ull;
  std::vector<double> ends(grid);
  for (int b = 0; b < grid; b++) s0 = h[2 * b] < s0 ? h[2 * b] : s0;
  for (int b = 0; b < grid; b++) ends[b] = (double)(h[2 * b + 1] - s0) * 0.01;
  std::sort(ends.begin(), ends.end());
  const double bytes = (double)(S + W) * n * 8.0 * reps;
  printf("%-40s ng %2u chunk %4u grid %4d  %8.1f GB/s  (%.4f ms/launch)  block ends us: min %.1f p50 %.1f p90 %.1f max %.1f\n", name, ng, chunk_tiles, grid,
         bytes / (ms * 1e-3) / 1e9, ms / reps, ends[0], ends[grid / 2], ends[grid * 9 / 10], ends[grid - 1]);
  fflush(stdout);
  CK(hipFree(ctr));
  CK(hipFree(stamps));
}

// The rolling window with T 16-byte loads per stream per thread and tile (tile = T x 512 doubles):
// LAY 0: the T chunks of a thread are 4 KiB apart (each wave instruction covers 1 KiB, the block
// T x 4 KiB contiguous); LAY 1: each WAVE covers T KiB contiguous (chunks 1 KiB apart).
template <int S, int W, int WIN, int NTS, int T, int LAY>
__global__ __launch_bounds__(256) void k_stream_winT(const double *base, double *wbase, size_t stride, size_t n, double *out) {
  constexpr int R = S - 2;
  static_assert(R % WIN == 0, "ring must divide the ringed streams");
  const size_t ntile = n / (512 * T);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto off = [&](size_t t, int q) -> size_t {
    return LAY ? t * (512 * T) + (size_t)wave * (128 * T) + q * 128 + lane * 2 : t * (512 * T) + q * 512 + threadIdx.x * 2;
  };
  double acc = 0.0;
  double2 a0[T], a1[T], ring[WIN][T];
  size_t t = blockIdx.x;
  if (t < ntile) {
#pragma unroll
    for (int q = 0; q < T; q++) {
      a0[q] = ld2<true>(base + off(t, q));
      a1[q] = ld2<true>(base + stride + off(t, q));
    }
#pragma unroll
    for (int j = 0; j < WIN; j++)
#pragma unroll
      for (int q = 0; q < T; q++) ring[j][q] = ld2<true>(base + (2 + j) * stride + off(t, q));
  }
  for (; t < ntile; t += gridDim.x) {
    const size_t tn = (t + gridDim.x < ntile) ? t + gridDim.x : t;
    double2 sum[T];
#pragma unroll
    for (int q = 0; q < T; q++) sum[q] = {a0[q].x + a1[q].x, a0[q].y + a1[q].y};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < T; q++) {
      a0[q] = ld2<true>(base + off(tn, q));
      a1[q] = ld2<true>(base + stride + off(tn, q));
    }
#pragma unroll
    for (int j = 0; j < R; j++) {
      double2 x[T];
#pragma unroll
      for (int q = 0; q < T; q++) x[q] = ring[j % WIN][q];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < T; q++) {
        if (j + WIN < R) ring[j % WIN][q] = ld2<true>(base + (2 + j + WIN) * stride + off(t, q));
        else ring[j % WIN][q] = ld2<true>(base + (2 + j + WIN - R) * stride + off(tn, q));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < T; q++) { sum[q].x += x[q].x; sum[q].y += x[q].y; }
    }
#pragma unroll
    for (int q = 0; q < T; q++) acc += sum[q].x + sum[q].y;
#pragma unroll
    for (int w = 0; w < W; w++)
#pragma unroll
      for (int q = 0; q < T; q++) st2<NTS>(wbase + w * stride + off(t, q), sum[q]);
  }
  if (acc == 12345.678) out[0] = acc;
}

template <int S, int W, int WIN, int NTS, int T, int LAY>
void runwT(const char *name, const double *base, double *wbase, size_t stride, size_t n, double *out, int grid) {
  static_assert(S <= 42 && W <= 6, "variant exceeds the allocated slots");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 3;
  hipLaunchKernelGGL((k_stream_winT<S, W, WIN, NTS, T, LAY>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((k_stream_winT<S, W, WIN, NTS, T, LAY>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)(S + W) * n * 8.0 * reps;
  printf("%-44s grid %5d  %8.1f GB/s  (%.3f ms/launch)\n", name, grid, bytes / (ms * 1e-3) / 1e9, ms / reps);
  fflush(stdout);
}

// Rolling window + wide tiles + tile tickets (one tile per ticket, requested one tile ahead).
template <int S, int W, int WIN, int NTS, int T, int LAY>
__global__ __launch_bounds__(256) void k_stream_tix(const double *base, double *wbase, size_t stride, size_t n, double *out,
                                                    unsigned *counter, unsigned ng) {
  constexpr int R = S - 2;
  static_assert(R % WIN == 0, "ring must divide the ringed streams");
  const size_t ntile = n / (512 * T);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto off = [&](size_t t, int q) -> size_t {
    return LAY ? t * (512 * T) + (size_t)wave * (128 * T) + q * 128 + lane * 2 : t * (512 * T) + q * 512 + threadIdx.x * 2;
  };
  __shared__ unsigned s_next[2];
  const unsigned grp = blockIdx.x % ng;
  counter += grp * 32;
  double acc = 0.0;
  double2 a0[T], a1[T], ring[WIN][T];
  size_t t = blockIdx.x, tnext = t + gridDim.x;
  unsigned par = 0;
  if (t < ntile) {
#pragma unroll
    for (int q = 0; q < T; q++) {
      a0[q] = ld2<true>(base + off(t, q));
      a1[q] = ld2<true>(base + stride + off(t, q));
    }
#pragma unroll
    for (int j = 0; j < WIN; j++)
#pragma unroll
      for (int q = 0; q < T; q++) ring[j][q] = ld2<true>(base + (2 + j) * stride + off(t, q));
  }
  while (t < ntile) {
    const bool more = tnext < ntile;
    unsigned tk = 0xffffffffu;
    if (threadIdx.x == 0 && more) tk = atomicAdd(counter, 1u) * ng + grp;   // a group's counter starts at 2G/ng
    const size_t tn = more ? tnext : t;
    double2 sum[T];
#pragma unroll
    for (int q = 0; q < T; q++) sum[q] = {a0[q].x + a1[q].x, a0[q].y + a1[q].y};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < T; q++) {
      a0[q] = ld2<true>(base + off(tn, q));
      a1[q] = ld2<true>(base + stride + off(tn, q));
    }
#pragma unroll
    for (int j = 0; j < R; j++) {
      double2 x[T];
#pragma unroll
      for (int q = 0; q < T; q++) x[q] = ring[j % WIN][q];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < T; q++) {
        if (j + WIN < R) ring[j % WIN][q] = ld2<true>(base + (2 + j + WIN) * stride + off(t, q));
        else ring[j % WIN][q] = ld2<true>(base + (2 + j + WIN - R) * stride + off(tn, q));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < T; q++) { sum[q].x += x[q].x; sum[q].y += x[q].y; }
    }
#pragma unroll
    for (int q = 0; q < T; q++) acc += sum[q].x + sum[q].y;
#pragma unroll
    for (int w = 0; w < W; w++)
#pragma unroll
      for (int q = 0; q < T; q++) st2<NTS>(wbase + w * stride + off(t, q), sum[q]);
    if (threadIdx.x == 0) s_next[par] = tk;
    __syncthreads();
    const unsigned nx = s_next[par];
    par ^= 1;
    t = tnext;
    tnext = nx == 0xffffffffu ? ntile : nx;
  }
  if (acc == 12345.678) out[0] = acc;
}

template <int S, int W, int WIN, int NTS, int T, int LAY>
void runtix(const char *name, const double *base, double *wbase, size_t stride, size_t n, double *out, int grid, unsigned ng) {
  static_assert(S <= 42 && W <= 6, "variant exceeds the allocated slots");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 5;
  const int CS = 32 * 16;
  unsigned *ctr;
  CK(hipMalloc(&ctr, sizeof(unsigned) * CS * (reps + 1)));
  std::vector<unsigned> init(CS * (reps + 1), (unsigned)(2 * grid / ng));
  CK(hipMemcpy(ctr, init.data(), sizeof(unsigned) * init.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_stream_tix<S, W, WIN, NTS, T, LAY>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out, ctr + reps * CS, ng);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((k_stream_tix<S, W, WIN, NTS, T, LAY>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out, ctr + r * CS, ng);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)(S + W) * n * 8.0 * reps;
  printf("%-52s ng %u grid %4d  %8.1f GB/s  (%.4f ms/launch)\n", name, ng, grid, bytes / (ms * 1e-3) / 1e9, ms / reps);
  fflush(stdout);
  CK(hipFree(ctr));
}

// Burst variant: each block reads B tiles of S streams, THEN writes B tiles of W streams.
template <int S, int W, int B, int NTS>
__global__ __launch_bounds__(256) void k_burst(const double *base, double *wbase, size_t stride, size_t n, double *out) {
  const size_t ngroup = n / 512 / B;
  double acc = 0.0;
  for (size_t g = blockIdx.x; g < ngroup; g += gridDim.x) {
    double2 sum = {0.0, 0.0};
    for (int q = 0; q < B; q++) {
      double2 v[S];
#pragma unroll
      for (int s = 0; s < S; s++) v[s] = ld2<true>(base + s * stride + (g * B + q) * 512 + threadIdx.x * 2);
#pragma unroll
      for (int s = 0; s < S; s++) { sum.x += v[s].x; sum.y += v[s].y; }
    }
    acc += sum.x + sum.y;
    for (int q = 0; q < B; q++)
#pragma unroll
      for (int w = 0; w < W; w++) st2<NTS>(wbase + w * stride + (g * B + q) * 512 + threadIdx.x * 2, sum);
  }
  if (acc == 12345.678) out[0] = acc;
}

// Software-pipelined variant: the stores of tile t are issued AFTER the loads of
// tile t+1, so no load ever waits (vmcnt is in issue order) behind a store's ack.
template <int S, int W, int NTS>
__global__ __launch_bounds__(256) void k_pipe(const double *base, double *wbase, size_t stride, size_t n, double *out) {
  const size_t ntile = n / 512;
  double acc = 0.0;
  size_t t = blockIdx.x;
  if (t >= ntile) return;
  double2 v[S];
#pragma unroll
  for (int s = 0; s < S; s++) v[s] = ld2<true>(base + s * stride + t * 512 + threadIdx.x * 2);
  double2 sum = {0.0, 0.0};
#pragma unroll
  for (int s = 0; s < S; s++) { sum.x += v[s].x; sum.y += v[s].y; }
  for (;;) {
    const size_t tn = t + gridDim.x;
    const bool more = tn < ntile;
    if (more) {
#pragma unroll
      for (int s = 0; s < S; s++) v[s] = ld2<true>(base + s * stride + tn * 512 + threadIdx.x * 2);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int w = 0; w < W; w++) st2<NTS>(wbase + w * stride + t * 512 + threadIdx.x * 2, sum);
    __builtin_amdgcn_sched_barrier(0);
    acc += sum.x + sum.y;
    if (!more) break;
    sum.x = 0.0; sum.y = 0.0;
#pragma unroll
    for (int s = 0; s < S; s++) { sum.x += v[s].x; sum.y += v[s].y; }
    t = tn;
  }
  if (acc == 12345.678) out[0] = acc;
}

template <int S, int W, int NTS>
void runp(const char *name, const double *base, double *wbase, size_t stride, size_t n, double *out, int grid) {
  static_assert(S <= 42 && W <= 6, "variant exceeds the allocated slots");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 3;
  hipLaunchKernelGGL((k_pipe<S, W, NTS>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((k_pipe<S, W, NTS>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)(S + W) * n * 8.0 * reps;
  printf("%-44s grid %5d  %8.1f GB/s  (%.3f ms/launch)\n", name, grid, bytes / (ms * 1e-3) / 1e9, ms / reps);
  fflush(stdout);
}

template <int S, int W, int B, int NTS>
void runb(const char *name, const double *base, double *wbase, size_t stride, size_t n, double *out, int grid) {
  static_assert(S <= 42 && W <= 6, "variant exceeds the allocated slots");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 3;
  hipLaunchKernelGGL((k_burst<S, W, B, NTS>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((k_burst<S, W, B, NTS>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)(S + W) * n * 8.0 * reps;
  printf("%-44s grid %5d  %8.1f GB/s  (%.3f ms/launch)\n", name, grid, bytes / (ms * 1e-3) / 1e9, ms / reps);
  fflush(stdout);
}

template <int S, int W, int T, bool NTL, int NTS, int MAP>
void run(const char *name, const double *base, double *wbase, size_t stride, size_t n, double *out, int grid) {
  static_assert(S <= 42 && W <= 6, "variant exceeds the allocated slots");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 3;
  hipLaunchKernelGGL((k_stream<S, W, T, NTL, NTS, MAP>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((k_stream<S, W, T, NTL, NTS, MAP>), dim3(grid), dim3(256), 0, 0, base, wbase, stride, n, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)(S + W) * n * 8.0 * reps;
  printf("%-44s grid %5d  %8.1f GB/s  (%.3f ms/launch)\n", name, grid, bytes / (ms * 1e-3) / 1e9, ms / reps);
  fflush(stdout);
}

// fill with pseudo-random doubles in (-1,1): all-zero buffers flatter a bandwidth probe
// (no data toggling on the HBM interface)
__global__ void k_fill_random(double *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    p[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  }
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? (size_t)atof(argv[1]) : 100000000;
  const size_t stride = ((n + 31) / 32) * 32 + (argc > 2 ? atoi(argv[2]) / 8 : 0);
  const int NS = 42, NW = 6;   // slots allocated for reads / writes; every variant is checked against these
  double *rd, *wr, *out;
  CK(hipMalloc(&rd, stride * 8 * NS));
  CK(hipMalloc(&wr, stride * 8 * NW));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(rd, 0, stride * 8 * NS));
  CK(hipMemset(wr, 0, stride * 8 * NW));
  const bool random_data = argc > 3 && (argv[3][0] == 'r' || (argv[3][0] && argv[3][1] == 'r'));   // "r" or "wr"
  if (random_data) {
    hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, rd, stride * NS);
    CK(hipDeviceSynchronize());
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cu = prop.multiProcessorCount;
  printf("device %s, %d CUs, n=%zu, slot stride %zu B, read data %s\n", prop.gcnArchName, cu, n, stride * 8,
         random_data ? "random" : "zeros");
#define R(S, W, T, NTL, NTS, MAP, G) run<S, W, T, NTL, NTS, MAP>("S=" #S " W=" #W " T=" #T " ntl=" #NTL " st=" #NTS " map=" #MAP, rd, wr, stride, n, out, G)
#define RB(S, W, B, NTS, G) runb<S, W, B, NTS>("burst S=" #S " W=" #W " B=" #B " st=" #NTS, rd, wr, stride, n, out, G)
#define RP(S, W, NTS, G) runp<S, W, NTS>("pipelined S=" #S " W=" #W " st=" #NTS, rd, wr, stride, n, out, G)
  if (argc > 3 && argv[3][0] == 'f') {   // does the fp64 arithmetic of the dot pass cost bandwidth?  ("f" zeros, "fr" random)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
      R(22, 0, 1, true, 0, 0, cu);
      hipLaunchKernelGGL((k_stream_fma<22>), dim3(cu), dim3(256), 0, 0, rd, stride, n, out);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_stream_fma<22>), dim3(cu), dim3(256), 0, 0, rd, stride, n, out);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-44s grid %5d  %8.1f GB/s  (%.3f ms/launch)\n", "S=22 pure read + 84 fp64 FMAs per tile-thread", cu,
             22.0 * n * 8.0 * 3 / (ms * 1e-3) / 1e9, ms / 3);
    }
    return 0;
  }
#define RW(S, W, WIN, NTS, G) runw<S, W, WIN, NTS>("window S=" #S " W=" #W " WIN=" #WIN " st=" #NTS, rd, wr, stride, n, out, G)
  if (argc > 3 && argv[3][0] == 'm') {   // mixed read/write: all loads in flight vs a small rolling window ("m" zeros, "mr" random)
    for (int rep = 0; rep < 2; rep++)
      for (int g : {cu * 1, cu * 2}) {
        R(22, 5, 1, true, 1, 0, g);
        RP(22, 5, 1, g);
        RW(22, 5, 2, 1, g); RW(22, 5, 4, 1, g); RW(22, 5, 5, 1, g); RW(22, 5, 10, 1, g); RW(22, 5, 4, 2, g);
        R(42, 5, 1, true, 1, 0, g);
        RW(42, 5, 4, 1, g); RW(42, 5, 8, 1, g); RW(42, 5, 10, 1, g);
        RW(22, 0, 4, 1, g);
      }
    return 0;
  }
#define RD(S, W, WIN, NTS, DYN, G, CH) rund<S, W, WIN, NTS, DYN>("S=" #S " W=" #W " WIN=" #WIN " dyn=" #DYN, rd, wr, stride, n, out, G, CH)
  if (argc > 3 && argv[3][0] == 'd') {   // static vs dynamic tile assignment, block end-time spread ("d" zeros, "dr" random)
    for (int rep = 0; rep < 2; rep++) {
      RW(22, 0, 4, 1, cu);
      RD(22, 0, 4, 1, false, cu, 1); RD(22, 0, 4, 1, false, cu, 8);
      RD(22, 0, 4, 1, true, cu, 1); RD(22, 0, 4, 1, true, cu, 4); RD(22, 0, 4, 1, true, cu, 16); RD(22, 0, 4, 1, true, cu, 64);
      RW(22, 5, 4, 1, cu);
      RD(22, 5, 4, 1, false, cu, 1);
      RD(22, 5, 4, 1, true, cu, 1); RD(22, 5, 4, 1, true, cu, 4); RD(22, 5, 4, 1, true, cu, 16); RD(22, 5, 4, 1, true, cu, 64);
      RW(42, 5, 4, 1, cu);
      RD(42, 5, 4, 1, false, cu, 1);
      RD(42, 5, 4, 1, true, cu, 4); RD(42, 5, 4, 1, true, cu, 16);
    }
    return 0;
  }
#define RD2(S, W, WIN, NTS, G, CH) rund2<S, W, WIN, NTS>("S=" #S " W=" #W " WIN=" #WIN " dyn=ahead", rd, wr, stride, n, out, G, CH)
  if (argc > 3 && argv[3][0] == 'e') {   // dynamic claiming one chunk ahead ("e" zeros, "er" random)
    for (int rep = 0; rep < 2; rep++) {
      RW(22, 0, 4, 1, cu);
      RD2(22, 0, 4, 1, cu, 1); RD2(22, 0, 4, 1, cu, 2); RD2(22, 0, 4, 1, cu, 4);
      RW(22, 5, 4, 1, cu);
      RD(22, 5, 4, 1, true, cu, 1);
      RD2(22, 5, 4, 1, cu, 1); RD2(22, 5, 4, 1, cu, 2); RD2(22, 5, 4, 1, cu, 4); RD2(22, 5, 4, 1, cu * 2, 1);
      RW(42, 5, 4, 1, cu);
      RD(42, 5, 4, 1, true, cu, 1);
      RD2(42, 5, 4, 1, cu, 1); RD2(42, 5, 4, 1, cu, 2); RD2(42, 5, 4, 1, cu, 4);
      RW(12, 5, 2, 1, cu);
      RD2(12, 5, 2, 1, cu, 1); RD2(12, 5, 2, 1, cu, 2);
    }
    return 0;
  }
#define RD3(S, W, WIN, NTS, G, CH, NG) rund2<S, W, WIN, NTS>("S=" #S " W=" #W " WIN=" #WIN " dyn=ahead", rd, wr, stride, n, out, G, CH, NG)
  if (argc > 3 && argv[3][0] == 'g') {   // one ticket counter per group of blocks (blockIdx % ng) ("g" zeros, "gr" random)
    for (int rep = 0; rep < 2; rep++) {
      RW(22, 0, 4, 1, cu);
      RD3(22, 0, 4, 1, cu, 1, 1); RD3(22, 0, 4, 1, cu, 1, 2); RD3(22, 0, 4, 1, cu, 1, 4); RD3(22, 0, 4, 1, cu, 1, 8); RD3(22, 0, 4, 1, cu, 1, 16);
      RW(22, 5, 4, 1, cu);
      RD3(22, 5, 4, 1, cu, 1, 1); RD3(22, 5, 4, 1, cu, 1, 2); RD3(22, 5, 4, 1, cu, 1, 4); RD3(22, 5, 4, 1, cu, 1, 8); RD3(22, 5, 4, 1, cu, 1, 16);
      RW(42, 5, 4, 1, cu);
      RD3(42, 5, 4, 1, cu, 1, 1); RD3(42, 5, 4, 1, cu, 1, 2); RD3(42, 5, 4, 1, cu, 1, 8);
      RW(12, 5, 2, 1, cu);
      RD3(12, 5, 2, 1, cu, 1, 1); RD3(12, 5, 2, 1, cu, 1, 2); RD3(12, 5, 2, 1, cu, 1, 8);
      RW(7, 5, 1, 1, cu);
      RD3(7, 5, 1, 1, cu, 1, 2); RD3(7, 5, 1, 1, cu, 1, 8);
    }
    return 0;
  }
#define RT(S, W, WIN, NTS, T, LAY, G, NG) runtix<S, W, WIN, NTS, T, LAY>("tickets S=" #S " W=" #W " WIN=" #WIN " st=" #NTS " T=" #T " lay=" #LAY, rd, wr, stride, n, out, G, NG)
  if (argc > 3 && argv[3][0] == 'h') {   // tile tickets x store policy x ring depth x tile width ("h" zeros, "hr" random)
    for (int rep = 0; rep < 2; rep++) {
      RW(22, 5, 4, 1, cu);
      RT(22, 5, 4, 1, 1, 0, cu, 1);
      RT(22, 5, 4, 0, 1, 0, cu, 1); RT(22, 5, 4, 2, 1, 0, cu, 1); RT(22, 5, 4, 3, 1, 0, cu, 1); RT(22, 5, 4, 4, 1, 0, cu, 1);
      RT(22, 5, 2, 1, 1, 0, cu, 1); RT(22, 5, 5, 1, 1, 0, cu, 1); RT(22, 5, 10, 1, 1, 0, cu, 1);
      RT(22, 5, 4, 1, 2, 0, cu, 1); RT(22, 5, 4, 1, 2, 1, cu, 1); RT(22, 5, 2, 1, 2, 0, cu, 1); RT(22, 5, 2, 1, 2, 1, cu, 1);
      RT(22, 5, 2, 1, 4, 1, cu, 1); RT(22, 5, 1, 1, 4, 1, cu, 1);
      RT(22, 5, 4, 1, 1, 0, cu / 2, 1); RT(22, 5, 4, 1, 2, 1, cu / 2, 1);
      RW(42, 5, 4, 1, cu);
      RT(42, 5, 4, 1, 1, 0, cu, 1); RT(42, 5, 2, 1, 1, 0, cu, 1); RT(42, 5, 8, 1, 1, 0, cu, 1); RT(42, 5, 4, 1, 2, 1, cu, 1); RT(42, 5, 2, 1, 2, 1, cu, 1);
      RW(12, 5, 2, 1, cu);
      RT(12, 5, 2, 1, 1, 0, cu, 2); RT(12, 5, 2, 1, 2, 1, cu, 1); RT(12, 5, 2, 1, 2, 0, cu, 1); RT(12, 5, 5, 1, 2, 1, cu, 1);
      RW(7, 5, 1, 1, cu);
      RT(7, 5, 1, 1, 1, 0, cu, 2); RT(7, 5, 1, 1, 2, 1, cu, 1); RT(7, 5, 5, 1, 2, 1, cu, 1); RT(7, 5, 5, 1, 4, 1, cu, 1);
    }
    return 0;
  }
#define RWT(S, W, WIN, NTS, T, LAY, G) runwT<S, W, WIN, NTS, T, LAY>("window S=" #S " W=" #W " WIN=" #WIN " T=" #T " lay=" #LAY, rd, wr, stride, n, out, G)
  if (argc > 3 && argv[3][0] == 'k') {   // few write streams: does the compact front still pay? ("k" zeros, "kr" random)
    for (int rep = 0; rep < 2; rep++) {
      RW(22, 0, 4, 1, cu); RT(22, 0, 4, 1, 1, 0, cu, 2);
      RW(22, 1, 4, 1, cu); RT(22, 1, 4, 1, 1, 0, cu, 1); RT(22, 1, 4, 1, 1, 0, cu, 2);
      RW(22, 2, 4, 1, cu); RT(22, 2, 4, 1, 1, 0, cu, 1); RT(22, 2, 4, 1, 1, 0, cu, 2);
      RW(22, 3, 4, 1, cu); RT(22, 3, 4, 1, 1, 0, cu, 1);
      RW(42, 3, 4, 1, cu); RT(42, 3, 4, 1, 1, 0, cu, 1);
    }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 'j') {   // tile tickets: true nt vs plain vs sc1 stores ("j" zeros, "jr" random)
    for (int rep = 0; rep < 3; rep++) {
      RW(22, 5, 4, 0, cu); RW(22, 5, 4, 1, cu); RW(22, 5, 4, 2, cu);
      RT(22, 5, 4, 0, 1, 0, cu, 1); RT(22, 5, 4, 1, 1, 0, cu, 1); RT(22, 5, 4, 2, 1, 0, cu, 1); RT(22, 5, 4, 3, 1, 0, cu, 1); RT(22, 5, 4, 5, 1, 0, cu, 1);
      RW(42, 5, 4, 1, cu);
      RT(42, 5, 4, 0, 1, 0, cu, 1); RT(42, 5, 4, 1, 1, 0, cu, 1); RT(42, 5, 4, 2, 1, 0, cu, 1);
      RW(12, 5, 2, 1, cu);
      RT(12, 5, 2, 1, 2, 1, cu, 1); RT(12, 5, 2, 2, 2, 1, cu, 1); RT(12, 5, 2, 1, 1, 0, cu, 2); RT(12, 5, 2, 2, 1, 0, cu, 2);
    }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 'i') {   // tile tickets + write-through stores x tile width x ring ("i" zeros, "ir" random)
    for (int rep = 0; rep < 2; rep++) {
      RW(22, 5, 4, 1, cu); RW(22, 5, 4, 2, cu); RWT(22, 5, 4, 2, 2, 1, cu);
      RT(22, 5, 4, 1, 1, 0, cu, 1);
      RT(22, 5, 4, 2, 1, 0, cu, 1); RT(22, 5, 5, 2, 1, 0, cu, 1); RT(22, 5, 10, 2, 1, 0, cu, 1);
      RT(22, 5, 4, 2, 2, 1, cu, 1); RT(22, 5, 2, 2, 2, 1, cu, 1); RT(22, 5, 4, 2, 2, 0, cu, 1); RT(22, 5, 2, 2, 2, 0, cu, 1);
      RT(22, 5, 2, 2, 4, 1, cu, 1);
      RT(22, 5, 4, 2, 1, 0, cu, 2); RT(22, 5, 4, 2, 2, 1, cu, 2);
      RW(42, 5, 4, 1, cu); RW(42, 5, 4, 2, cu);
      RT(42, 5, 4, 2, 1, 0, cu, 1); RT(42, 5, 8, 2, 1, 0, cu, 1); RT(42, 5, 4, 2, 2, 1, cu, 1); RT(42, 5, 2, 2, 2, 1, cu, 1);
      RW(12, 5, 2, 1, cu);
      RT(12, 5, 2, 2, 2, 1, cu, 1); RT(12, 5, 5, 2, 2, 1, cu, 1); RT(12, 5, 2, 2, 1, 0, cu, 2); RT(12, 5, 2, 2, 4, 1, cu, 1);
      RW(7, 5, 1, 1, cu);
      RT(7, 5, 1, 2, 2, 1, cu, 1); RT(7, 5, 5, 2, 2, 1, cu, 1); RT(7, 5, 1, 2, 1, 0, cu, 2); RT(7, 5, 5, 2, 4, 1, cu, 1);
      RW(4, 5, 1, 1, cu);
      RT(4, 5, 2, 2, 2, 1, cu, 1); RT(4, 5, 2, 2, 4, 1, cu, 1);
    }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 't') {   // wider tiles in the rolling window ("t" zeros, "tr" random)
    for (int rep = 0; rep < 2; rep++)
      for (int g : {cu * 1, cu / 2}) {
        RW(22, 5, 4, 1, g);
        RWT(22, 5, 4, 1, 1, 0, g);
        RWT(22, 5, 2, 1, 2, 0, g); RWT(22, 5, 2, 1, 2, 1, g);
        RWT(22, 5, 4, 1, 2, 0, g); RWT(22, 5, 4, 1, 2, 1, g);
        RWT(22, 5, 2, 1, 4, 0, g); RWT(22, 5, 2, 1, 4, 1, g);
        RWT(22, 5, 1, 1, 4, 1, g);
        RW(42, 5, 4, 1, g);
        RWT(42, 5, 4, 1, 2, 1, g); RWT(42, 5, 2, 1, 2, 1, g); RWT(42, 5, 2, 1, 4, 1, g);
        RW(22, 0, 4, 1, g);
        RWT(22, 0, 4, 1, 2, 1, g); RWT(22, 0, 2, 1, 2, 1, g); RWT(22, 0, 2, 1, 4, 1, g);
      }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 'c') {   // the ceilings bench.py quotes in its JSON line ("c" zeros, "cr" random): pure read,
                                         // pure write, and the read/write mixes of PB (compact: 22 R + 5 W, two-vector: 42 R + 5 W)
    for (int rep = 0; rep < 2; rep++) {
      RW(22, 0, 4, 1, cu);
      R(0, 4, 1, true, 2, 0, cu);
      RT(22, 5, 4, 1, 1, 0, cu, 1);
      RT(42, 5, 4, 1, 1, 0, cu, 1);
    }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 's') {   // what would ONE store stream less buy PB (an out-of-place update that lends f_out)?  ("s", "sr")
    for (int rep = 0; rep < 3; rep++) {
      RT(22, 5, 4, 1, 1, 0, cu, 1); RT(22, 4, 4, 1, 1, 0, cu, 1); RT(22, 3, 4, 1, 1, 0, cu, 1);
      RT(42, 5, 4, 1, 1, 0, cu, 1); RT(42, 4, 4, 1, 1, 0, cu, 1);
    }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 'w') {   // pure-write study: store policy x tiles per iteration x blocks per CU
    for (int g : {cu * 1, cu * 2, cu * 4, cu * 8}) {
      R(0, 4, 1, true, 0, 0, g); R(0, 4, 1, true, 1, 0, g); R(0, 4, 1, true, 2, 0, g); R(0, 4, 1, true, 3, 0, g); R(0, 4, 1, true, 4, 0, g);
      R(0, 4, 4, true, 0, 0, g); R(0, 4, 4, true, 1, 0, g); R(0, 4, 4, true, 4, 0, g);
      R(0, 1, 4, true, 1, 0, g); R(0, 1, 4, true, 1, 1, g);
    }
    return 0;
  }
  for (int rep = 0; rep < 2; rep++)
  for (int g : {cu * 1, cu * 2}) {
    R(22, 5, 1, true, 1, 0, g);  RP(22, 5, 1, g); RP(22, 5, 0, g);
    R(41, 5, 1, true, 1, 0, g);  RP(41, 5, 1, g);
    R(22, 0, 1, true, 0, 0, g);
  }
  return 0;
}
