// tools/wave_sum_check.hip -- is the register-only wave reduction (v_permlane32_swap / v_permlane16_swap / DPP row_shl,
// gfx950) the SAME TREE as the __shfl_down butterfly it replaced?  Lane 0's sum must agree bit for bit.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../nka_amd/csrc wave_sum_check.hip -o wave_sum_check && ./wave_sum_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "nka_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ double wave_sum_shfl(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

__global__ void k_check(const double *in, double *a, double *b, int nwaves) {
  const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  if (w >= nwaves) return;
  const double x = in[(size_t)w * 64 + (threadIdx.x & 63)];
  const double ra = wave_sum_shfl(x), rb = nka::wave_sum(x);
  if ((threadIdx.x & 63) == 0) { a[w] = ra; b[w] = rb; }
}

// the block reduction as it was: one __shfl_down butterfly per accumulator, then waves 0..3
template <int NACC>
__device__ void block_reduce_store_shfl(const double (&acc)[NACC], double *partials, int G) {
  __shared__ double sm[4][NACC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < NACC; a++) {
    const double r = wave_sum_shfl(acc[a]);
    if (lane == 0) sm[wv][a] = r;
  }
  __syncthreads();
  for (int a = threadIdx.x; a < NACC; a += 256) {
    double r = sm[0][a];
    for (int q = 1; q < 4; q++) r += sm[q][a];
    partials[(size_t)a * G + blockIdx.x] = r;
  }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_block(const double *in, double *pa, double *pb) {
  double acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; a++) acc[a] = in[((size_t)blockIdx.x * NACC + a) * 256 + threadIdx.x];
  block_reduce_store_shfl<NACC>(acc, pa, gridDim.x);
  __syncthreads();
  nka::block_reduce_store<NACC>(acc, pb, gridDim.x);
}

template <int NACC>
int check_block(const std::vector<double> &h, double *din) {
  const int G = 64;
  double *pa, *pb;
  CK(hipMalloc(&pa, (size_t)NACC * G * 8));
  CK(hipMalloc(&pb, (size_t)NACC * G * 8));
  CK(hipMemset(pa, 0xff, (size_t)NACC * G * 8));
  CK(hipMemset(pb, 0, (size_t)NACC * G * 8));
  k_block<NACC><<<G, 256>>>(din, pa, pb);
  std::vector<double> ha((size_t)NACC * G), hb((size_t)NACC * G);
  CK(hipMemcpy(ha.data(), pa, ha.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), pb, hb.size() * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (size_t i = 0; i < ha.size(); i++) bad += memcmp(&ha[i], &hb[i], 8) != 0;
  printf("block_reduce_store<%d>: %d of %zu block sums differ from the per-accumulator butterflies\n", NACC, bad, ha.size());
  CK(hipFree(pa));
  CK(hipFree(pb));
  return bad;
}

int main() {
  const int nwaves = 1 << 16;
  std::vector<double> h((size_t)nwaves * 64);
  srand(7);
  for (auto &v : h) {
    const int e = rand() % 40 - 20;
    v = ((double)rand() / RAND_MAX - 0.5) * pow(2.0, e);
  }
  double *din, *da, *db;
  CK(hipMalloc(&din, h.size() * 8));
  CK(hipMalloc(&da, nwaves * 8));
  CK(hipMalloc(&db, nwaves * 8));
  CK(hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  k_check<<<nwaves / 4, 256>>>(din, da, db, nwaves);
  std::vector<double> ha(nwaves), hb(nwaves);
  CK(hipMemcpy(ha.data(), da, nwaves * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), db, nwaves * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < nwaves; i++) bad += memcmp(&ha[i], &hb[i], 8) != 0;
  printf("wave_sum: %d of %d wavefront sums differ from the __shfl_down butterfly\n", bad, nwaves);
  bad += check_block<1>(h, din) + check_block<2>(h, din) + check_block<3>(h, din) + check_block<5>(h, din) + check_block<7>(h, din) +
         check_block<12>(h, din) + check_block<13>(h, din) + check_block<22>(h, din) + check_block<41>(h, din) +
         check_block<42>(h, din) + check_block<49>(h, din) + check_block<66>(h, din);
  return bad != 0;
}
