// Micro-benchmark (tools/micro): latency of DEPENDENT v_add_f64 / v_add_f32 / DPP steps on one wavefront of an MI355X, alone and with
// every other compute unit busy (does the shader clock depend on the load?).  hipcc --offload-arch=gfx950 -O3 dep_add.hip -o dep_add
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void dep_add64(double *out, const double *in, int iters) {
  double a = in[0], p = in[1];
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 64; u++) a = a + p;
  }
  if (a == 12345.678) out[0] = a;     // (keeps the chain alive)
  out[threadIdx.x + blockIdx.x * blockDim.x + 1] = a;
}
__global__ void dep_add32(float *out, const float *in, int iters) {
  float a = in[0], p = in[1];
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 64; u++) a = a + p;
  }
  out[threadIdx.x + blockIdx.x * blockDim.x + 1] = a;
}
__global__ void indep_add64(double *out, const double *in, int iters) {
  double a[8];
  for (int k = 0; k < 8; k++) a[k] = in[k];
  const double p = in[9];
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int k = 0; k < 8; k++) a[k] = a[k] + p;
  }
  double s = 0;
  for (int k = 0; k < 8; k++) s += a[k];
  out[threadIdx.x + blockIdx.x * blockDim.x + 1] = s;
}
__global__ void dep_dpp(double *out, const double *in, int iters) {
  union { double d; int u[2]; } a, b;
  a.d = in[threadIdx.x & 7];
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 64; u++) {
      b.u[0] = __builtin_amdgcn_update_dpp(0, a.u[0], 0x111, 0xf, 0xf, false);
      b.u[1] = __builtin_amdgcn_update_dpp(0, a.u[1], 0x111, 0xf, 0xf, false);
      a.d = a.d + b.d;
    }
  }
  out[threadIdx.x + blockIdx.x * blockDim.x + 1] = a.d;
}

template <typename K, typename T>
static int run(const char *name, K kern, T *out, const T *in, int blocks, int threads, int iters, double ops_per_iter) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, in, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, in, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-34s blocks=%4d threads=%4d: %8.3f ms  %7.3f ns per step\n", name, blocks, threads, ms, 1e6 * ms / (iters * ops_per_iter));
  return 0;
}

int main() {
  double *out, *in; float *outf, *inf;
  CHECK(hipMalloc(&out, 8 * (1 << 20))); CHECK(hipMalloc(&in, 8 * 64));
  CHECK(hipMalloc(&outf, 4 * (1 << 20))); CHECK(hipMalloc(&inf, 4 * 64));
  double h[64]; float hf[64];
  for (int i = 0; i < 64; i++) { h[i] = 1e-3 * (i + 1); hf[i] = 1e-3f * (i + 1); }
  CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice)); CHECK(hipMemcpy(inf, hf, sizeof hf, hipMemcpyHostToDevice));
  const int it = 20000;
  for (int blocks : {1, 256, 1024}) {
    for (int threads : {64, 256, 512}) {
      if (run("dependent v_add_f64", dep_add64, out, in, blocks, threads, it, 64)) return 1;
    }
    if (run("dependent v_add_f32", dep_add32, outf, inf, blocks, 64, it, 64)) return 1;
    if (run("8 independent chains v_add_f64", indep_add64, out, in, blocks, 64, it, 64)) return 1;
    if (run("dependent dpp+dpp+add f64", dep_dpp, out, in, blocks, 64, it, 64)) return 1;
  }
  return 0;
}
