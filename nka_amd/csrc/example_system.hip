// example_system.hip -- device-resident caller of the hot path:
// the finite-volume system of the reference's example program,
// /root/reference/src-F08/nka_example.F90:86-179, as HIP kernels behind the C ABI
// of include/nka_example_dev.h.  Compiled with -ffp-contract=off: every
// expression rounds like the Fortran statement it restates, so a solve driven
// through these kernels reproduces the reference's printed tables.
#include "../../include/nka_example_dev.h"
#include "../../include/nka_hip.h"

#include <hip/hip_runtime.h>

#include <string>

namespace nka_detail {
int set_error(int code, const std::string &msg);
int check_device_span(const void *p, int64_t n, const char *what);
void invalidate_span_cache();
}

struct nka_ex_system {
  int device = 0;
  hipStream_t stream = nullptr;
  int nx = 0, ny = 0;
  double a = 0.0, hx = 0.0, hy = 0.0;
  double *ax = nullptr;  // (nx+1) x ny : ax(j,k) at [(j-1) + (k-1)*(nx+1)]
  double *ay = nullptr;  // nx x (ny+1) : ay(j,k) at [(j-1) + (k-1)*nx]
  double *ac = nullptr;  // nx x ny
  double *z = nullptr;   // (nx+2) x (ny+2) work array of pc_ssor
};

namespace {

#define EX_TRY(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return nka_detail::set_error(e_ == hipErrorOutOfMemory ? NKA_HIP_ENOMEM : NKA_HIP_EHIP, \
                                   std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

constexpr int kT = 256;

// Two layouts of a cell-centred field with its boundary ring:
//   GRID = false: the natural (nx+2) x (ny+2) array, u(j,k) at [j + k*(nx+2)]
//   GRID = true : the device grid vector of nka_amd/fortran/vector/hip_grid_vector_type.F90 --
//                 the nx*ny interior values packed first, u(j,k) at [(j-1) + (k-1)*nx], then the
//                 ring: row k = 0 (j = 0..nx+1), row k = ny+1, column j = 0 (k = 1..ny), column
//                 j = nx+1.  Reductions of the vector hooks then run over a dense prefix.
template <bool GRID>
__device__ __forceinline__ double u_at(const double *__restrict__ u, int nx, int ny, int j, int k) {
  if (!GRID) return u[j + (int64_t)k * (nx + 2)];
  if (j >= 1 && j <= nx && k >= 1 && k <= ny) return u[(j - 1) + (int64_t)(k - 1) * nx];
  const double *ring = u + (int64_t)nx * ny;
  if (k == 0) return ring[j];
  if (k == ny + 1) return ring[(nx + 2) + j];
  if (j == 0) return ring[2 * (nx + 2) + (k - 1)];
  return ring[2 * (nx + 2) + ny + (k - 1)];
}

// nka_example.F90:122-142.  The reference accumulates t*h^2 of the two cells next to
// a face into a zeroed array (k outer, j inner: the lower-index cell first) and then
// takes 2/sum: one thread per face does the same two additions in the same order.
template <bool GRID>
__global__ __launch_bounds__(kT) void k_ex_faces(int nx, int ny, double a, double hx2, double hy2,
                                                 const double *__restrict__ uext, double *__restrict__ ax,
                                                 double *__restrict__ ay) {
  const int64_t nax = (int64_t)(nx + 1) * ny, nay = (int64_t)nx * (ny + 1);
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < nax + nay; i += (int64_t)gridDim.x * kT) {
    if (i < nax) {
      const int j = (int)(i % (nx + 1)) + 1, k = (int)(i / (nx + 1)) + 1;   // face between cells (j-1,k) and (j,k)
      double sum = 0.0;
      if (j - 1 >= 1) sum = sum + ((1.0 / (a + u_at<GRID>(uext, nx, ny, j - 1, k))) * hx2);
      if (j <= nx) sum = sum + ((1.0 / (a + u_at<GRID>(uext, nx, ny, j, k))) * hx2);
      ax[i] = 2.0 / sum;
    } else {
      const int64_t q = i - nax;
      const int j = (int)(q % nx) + 1, k = (int)(q / nx) + 1;               // face between cells (j,k-1) and (j,k)
      double sum = 0.0;
      if (k - 1 >= 1) sum = sum + ((1.0 / (a + u_at<GRID>(uext, nx, ny, j, k - 1))) * hy2);
      if (k <= ny) sum = sum + ((1.0 / (a + u_at<GRID>(uext, nx, ny, j, k))) * hy2);
      ay[q] = 2.0 / sum;
    }
  }
}

// nka_example.F90:143-145 (ac) and :112-118 (the residual), q = 1 (:100)
template <bool GRID>
__global__ __launch_bounds__(kT) void k_ex_residual(int nx, int ny, const double *__restrict__ uext,
                                                    const double *__restrict__ ax, const double *__restrict__ ay,
                                                    double *__restrict__ ac, double *__restrict__ r) {
  const int64_t n = (int64_t)nx * ny, ldx = nx + 1;
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) {
    const int j = (int)(i % nx) + 1, k = (int)(i / nx) + 1;
    const double axl = ax[(j - 1) + (k - 1) * ldx], axr = ax[j + (k - 1) * ldx];
    const double ayl = ay[(j - 1) + (int64_t)(k - 1) * nx], ayr = ay[(j - 1) + (int64_t)k * nx];
    const double c = axl + axr + ayl + ayr;
    ac[i] = c;
    r[i] = c * u_at<GRID>(uext, nx, ny, j, k) - axl * u_at<GRID>(uext, nx, ny, j - 1, k) -
           axr * u_at<GRID>(uext, nx, ny, j + 1, k) - ayl * u_at<GRID>(uext, nx, ny, j, k - 1) -
           ayr * u_at<GRID>(uext, nx, ny, j, k + 1) - 1.0;
  }
}

// nka_example.F90:147-179 as anti-diagonal wavefronts inside ONE workgroup.  The
// update of z(j,k) in the forward sweep reads the NEW z(j-1,k), z(j,k-1) -- both on
// diagonal j+k-1, finished before the barrier -- and the OLD z(j+1,k), z(j,k+1) on
// diagonal j+k+1, untouched so far: exactly what the lexicographic loops read.
// The backward sweep is the mirror image.  z lives in global memory (L2 resident).
constexpr int kSsorThreads = 1024;
__global__ __launch_bounds__(kSsorThreads) void k_ex_ssor(int nx, int ny, int nsweep, double omega,
                                                          const double *__restrict__ ax, const double *__restrict__ ay,
                                                          const double *__restrict__ ac, double *r, double *z) {
  const int64_t ldz = nx + 2, ldx = nx + 1;
  const int64_t nz = ldz * (ny + 2);
  for (int64_t i = threadIdx.x; i < nz; i += kSsorThreads) z[i] = 0.0;
  __syncthreads();
  const double om1 = 1 - omega;
  auto relax = [&](int j, int k) {
    const int64_t c = j + k * ldz;
    const double s = r[(j - 1) + (int64_t)(k - 1) * nx] + ax[(j - 1) + (k - 1) * ldx] * z[c - 1] +
                     ax[j + (k - 1) * ldx] * z[c + 1] + ay[(j - 1) + (int64_t)(k - 1) * nx] * z[c - ldz] +
                     ay[(j - 1) + (int64_t)k * nx] * z[c + ldz];
    z[c] = om1 * z[c] + omega * s / ac[(j - 1) + (int64_t)(k - 1) * nx];
  };
  for (int it = 0; it < nsweep; it++) {
    for (int d = 2; d <= nx + ny; d++) {                 // forward: k = 1..ny, j = 1..nx
      const int jlo = d - ny > 1 ? d - ny : 1, jhi = d - 1 < nx ? d - 1 : nx;
      for (int j = jlo + threadIdx.x; j <= jhi; j += kSsorThreads) relax(j, d - j);
      __syncthreads();
    }
    for (int d = nx + ny; d >= 2; d--) {                 // backward: k = ny..1, j = nx..1
      const int jlo = d - ny > 1 ? d - ny : 1, jhi = d - 1 < nx ? d - 1 : nx;
      for (int j = jlo + threadIdx.x; j <= jhi; j += kSsorThreads) relax(j, d - j);
      __syncthreads();
    }
  }
  const int64_t n = (int64_t)nx * ny;
  for (int64_t i = threadIdx.x; i < n; i += kSsorThreads) {       // r = z(1:nx,1:ny)   :178
    const int j = (int)(i % nx) + 1, k = (int)(i / nx) + 1;
    r[i] = z[j + k * ldz];
  }
}

// u = u - r on the interior of uext   (nka_example.F90:248)
__global__ __launch_bounds__(kT) void k_ex_update(int nx, int ny, double *__restrict__ uext,
                                                  const double *__restrict__ r) {
  const int64_t n = (int64_t)nx * ny, ldu = nx + 2;
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) {
    const int j = (int)(i % nx) + 1, k = (int)(i / nx) + 1;
    uext[j + k * ldu] = uext[j + k * ldu] - r[i];
  }
}

int grid_of(int64_t n) {
  int64_t g = (n + kT - 1) / kT;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

namespace {
template <bool GRID>
int residual_impl(nka_ex_t s, const double *u, double *r, const char *who) {
  if (!s) return nka_detail::set_error(NKA_HIP_EINVAL, "null system");
  EX_TRY(hipSetDevice(s->device));
  const int64_t n = (int64_t)s->nx * s->ny, next = (int64_t)(s->nx + 2) * (s->ny + 2);
  if (int rc = nka_detail::check_device_span(u, next, who)) return rc;
  if (int rc = nka_detail::check_device_span(r, GRID ? next : n, who)) return rc;
  const int64_t nfaces = (int64_t)(s->nx + 1) * s->ny + (int64_t)s->nx * (s->ny + 1);
  hipLaunchKernelGGL(k_ex_faces<GRID>, dim3(grid_of(nfaces)), dim3(kT), 0, s->stream, s->nx, s->ny, s->a, s->hx * s->hx,
                     s->hy * s->hy, u, s->ax, s->ay);
  hipLaunchKernelGGL(k_ex_residual<GRID>, dim3(grid_of(n)), dim3(kT), 0, s->stream, s->nx, s->ny, u, s->ax, s->ay, s->ac, r);
  EX_TRY(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" {

int nka_ex_create(nka_ex_t *out, int32_t nx, int32_t ny, double a, int32_t device, void *stream) {
  if (!out) return nka_detail::set_error(NKA_HIP_EINVAL, "nka_ex_create: out is NULL");
  *out = nullptr;
  if (!(a > 0.0) || nx < 3 || ny < 3)   // nka_example.F90:90-92
    return nka_detail::set_error(NKA_HIP_EINVAL, "nka_ex_create: need a > 0, nx >= 3, ny >= 3");
  int ndev = 0;
  EX_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return nka_detail::set_error(NKA_HIP_EINVAL, "nka_ex_create: no such HIP device");
  EX_TRY(hipSetDevice(device));
  auto *s = new nka_ex_system();
  s->device = device;
  s->stream = (hipStream_t)stream;
  s->nx = nx;
  s->ny = ny;
  s->a = a;
  s->hx = 1.0 / nx;
  s->hy = 1.0 / ny;
  hipError_t e = hipMalloc((void **)&s->ax, sizeof(double) * (size_t)(nx + 1) * ny);
  if (e == hipSuccess) e = hipMalloc((void **)&s->ay, sizeof(double) * (size_t)nx * (ny + 1));
  if (e == hipSuccess) e = hipMalloc((void **)&s->ac, sizeof(double) * (size_t)nx * ny);
  if (e == hipSuccess) e = hipMalloc((void **)&s->z, sizeof(double) * (size_t)(nx + 2) * (ny + 2));
  if (e != hipSuccess) {
    (void)hipGetLastError();   // reported here: not left for a later hipGetLastError() to find
    nka_ex_destroy(s);
    return nka_detail::set_error(NKA_HIP_ENOMEM, std::string("nka_ex_create: ") + hipGetErrorString(e));
  }
  *out = s;
  return 0;
}

int nka_ex_destroy(nka_ex_t s) {
  if (!s) return 0;
  nka_detail::invalidate_span_cache();
  hipSetDevice(s->device);
  hipStreamSynchronize(s->stream);
  hipFree(s->ax);
  hipFree(s->ay);
  hipFree(s->ac);
  hipFree(s->z);
  delete s;
  return 0;
}

int nka_ex_residual(nka_ex_t s, const double *uext, double *r) { return residual_impl<false>(s, uext, r, "nka_ex_residual"); }

// grid-vector layout: the interior prefix of r is written, its ring is left alone
// (src-F08-vector/nka_example.F90:103-120 assigns r(1:nx,1:ny) only)
int nka_ex_residual_grid(nka_ex_t s, const double *u_grid, double *r_grid) {
  return residual_impl<true>(s, u_grid, r_grid, "nka_ex_residual_grid");
}

int nka_ex_pc_ssor(nka_ex_t s, int32_t nsweep, double omega, double *r) {
  if (!s) return nka_detail::set_error(NKA_HIP_EINVAL, "null system");
  if (nsweep < 1 || !(omega > 0.0)) return nka_detail::set_error(NKA_HIP_EINVAL, "pc_ssor: nsweep >= 1, omega > 0");  // :158-159
  EX_TRY(hipSetDevice(s->device));
  if (int rc = nka_detail::check_device_span(r, (int64_t)s->nx * s->ny, "nka_ex_pc_ssor: r")) return rc;
  hipLaunchKernelGGL(k_ex_ssor, dim3(1), dim3(kSsorThreads), 0, s->stream, s->nx, s->ny, nsweep, omega, s->ax, s->ay,
                     s->ac, r, s->z);
  EX_TRY(hipGetLastError());
  return 0;
}

// r(:,:) = z with z = 0 on the ring (src-F08-vector/nka_example.F90:150, 178): the interior
// prefix through the same sweeps, the ring zeroed
int nka_ex_pc_ssor_grid(nka_ex_t s, int32_t nsweep, double omega, double *r_grid) {
  if (!s) return nka_detail::set_error(NKA_HIP_EINVAL, "null system");
  const int64_t n = (int64_t)s->nx * s->ny, next = (int64_t)(s->nx + 2) * (s->ny + 2);
  EX_TRY(hipSetDevice(s->device));
  if (int rc = nka_detail::check_device_span(r_grid, next, "nka_ex_pc_ssor_grid: r")) return rc;
  if (int rc = nka_ex_pc_ssor(s, nsweep, omega, r_grid)) return rc;
  EX_TRY(hipMemsetAsync(r_grid + n, 0, sizeof(double) * (size_t)(next - n), s->stream));
  return 0;
}

int nka_ex_update_solution(nka_ex_t s, double *uext, const double *r) {
  if (!s) return nka_detail::set_error(NKA_HIP_EINVAL, "null system");
  EX_TRY(hipSetDevice(s->device));
  const int64_t n = (int64_t)s->nx * s->ny, next = (int64_t)(s->nx + 2) * (s->ny + 2);
  if (int rc = nka_detail::check_device_span(uext, next, "nka_ex_update_solution: uext")) return rc;
  if (int rc = nka_detail::check_device_span(r, n, "nka_ex_update_solution: r")) return rc;
  hipLaunchKernelGGL(k_ex_update, dim3(grid_of(n)), dim3(kT), 0, s->stream, s->nx, s->ny, uext, r);
  EX_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
