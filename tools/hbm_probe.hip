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
    __builtin_nontemporal_store(v.x, p);
    __builtin_nontemporal_store(v.y, p + 1);
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
