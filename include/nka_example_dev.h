/*
 * nka_example_dev.h -- device-resident versions of the CALLER of the hot path in
 * the reference's example program: the finite-volume system
 * of /root/reference/src-F08/nka_example.F90
 *     update_system  :122-145     residual  :103-120     pc_ssor  :147-179
 *     u = u - r      :248
 * so that a whole nonlinear solve keeps u and r in HBM and hands r to
 * nka_hip_accel_update (device pointer) -- nothing crosses PCIe per iteration.
 *
 * Every expression is evaluated in the reference's order without fused
 * multiply-add, and the SSOR sweeps -- lexicographic Gauss-Seidel in the
 * reference -- run as anti-diagonal WAVEFRONTS: all points with j+k = d depend
 * only on new values of diagonal d-1 (forward sweep; d+1 backward) and old
 * values of the other side, so the wavefront order produces the SAME bits as the
 * sequential loops.  The device solve therefore prints the reference_output tables.
 *
 * Arrays (device memory, column-major like the Fortran):
 *   uext : (nx+2) x (ny+2), solution with its boundary ring, uext(j,k) at [j + k*(nx+2)]
 *   r    : nx x ny,          r(j,k) at [(j-1) + (k-1)*nx]
 * Part of libnka_hip.so; returns 0 or a negative NKA_HIP_E* code (nka_hip_last_error()).
 */
#ifndef NKA_EXAMPLE_DEV_H
#define NKA_EXAMPLE_DEV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nka_ex_system *nka_ex_t;

/* system%init(a, nx, ny)  (nka_example.F90:86-101): a > 0, nx, ny >= 3; q = 1. */
int nka_ex_create(nka_ex_t *out, int32_t nx, int32_t ny, double a, int32_t device, void *stream);
int nka_ex_destroy(nka_ex_t s);
/* call sys%residual(uext, r)  (:103-120; rebuilds the coefficients from uext first, :122-145). */
int nka_ex_residual(nka_ex_t s, const double *uext_dev, double *r_dev);
/* call sys%pc_ssor(nsweep, omega, r)  (:147-179): r <- SSOR-preconditioned r. */
int nka_ex_pc_ssor(nka_ex_t s, int32_t nsweep, double omega, double *r_dev);
/* u = u - r on the interior of uext  (:248). */
int nka_ex_update_solution(nka_ex_t s, double *uext_dev, const double *r_dev);

/* The same system on DEVICE GRID VECTORS (the abstract-vector flavour of the example,
 * /root/reference/src-F08-vector/nka_example.F90:103-120, 147-179, whose u and r are
 * grid_vector objects: (nx+2) x (ny+2) values, elementwise operations cover the ghost
 * ring, reductions do not -- grid_vector_type.F90:104-195).  Layout of a device grid
 * vector (nka_amd/fortran/vector/hip_grid_vector_type.F90): the nx*ny interior values
 * first, x(j,k) at [(j-1) + (k-1)*nx], then the ring -- row k = 0 (j = 0..nx+1), row
 * k = ny+1, column j = 0 (k = 1..ny), column j = nx+1 -- so every reduction of the
 * vector hooks runs over a dense, aligned prefix.
 *   residual_grid: r(1:nx,1:ny) <- residual(u); the ring of r is left alone (:112-118)
 *   pc_ssor_grid : r(:,:) <- z, i.e. the preconditioned interior and a ZERO ring (:178) */
int nka_ex_residual_grid(nka_ex_t s, const double *u_grid_dev, double *r_grid_dev);
int nka_ex_pc_ssor_grid(nka_ex_t s, int32_t nsweep, double omega, double *r_grid_dev);

#ifdef __cplusplus
}
#endif
#endif /* NKA_EXAMPLE_DEV_H */
