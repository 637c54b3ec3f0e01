/*
 * nka_c_compat.h -- the nine functions of the reference's C API
 * (/root/reference/src-C/nonlinear_krylov_accelerator.h:3-12) over libnka_hip.so
 * (SURVEY.md 8 row f2).  Header only; link with -lnka_hip.
 *
 * Same names, argument order and meaning.  `f` is HOST memory, as in the
 * reference (nka_accel_update copies it to the GPU and back); a caller whose
 * vectors already live in HBM uses nka_accel_update_dev.  The reference's `dp`
 * argument (.c:196, 211, 227-231): NULL selects the device sums (the fast
 * path; a multi-rank run then installs nka_hip_set_allreduce /
 * nka_hip_comm_init_rank on the returned handle); a non-NULL dp is honoured
 * through nka_hip_set_host_dot -- every inner product of every update is then
 * evaluated by dp on host copies of the operands in the reference's own order
 * (slow compatibility path, see include/nka_hip.h).  Failed preconditions abort
 * like the reference's assert() (.c:216-218).
 *
 * Two ways to use it: include this header (everything static inline), or keep the
 * reference's own nonlinear_krylov_accelerator.h and link libnka_c_compat.so
 * (nka_amd/csrc/nka_c_compat_lib.c: this header compiled with external linkage,
 * exporting exactly the nine reference symbols) -- then not even the include line
 * of a caller changes (oracle/Makefile: dropin_example_c builds the reference's
 * unchanged src-C/nka_example.c that way).
 */
#ifndef NKA_C_COMPAT_H
#define NKA_C_COMPAT_H

#include <stdio.h>
#include <stdlib.h>

#include "nka_hip.h"

typedef nka_hip_t NKA;

#ifndef NKA_C_COMPAT_API
#define NKA_C_COMPAT_API static inline
#endif

static inline void nka_compat_check_(int rc, const char *what) {
  if (rc != 0) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, nka_hip_last_error());
    abort();
  }
}

typedef double (*nka_compat_dp_fn_)(int, double *, double *);

/* nka_hip_host_dot_fn over the reference's dp(int, double*, double*): ctx is dp itself */
static inline double nka_compat_dp_trampoline_(void *ctx, int64_t n, const double *x, const double *y) {
  union { void *p; nka_compat_dp_fn_ f; } u;
  u.p = ctx;
  return u.f((int)n, (double *)x, (double *)y);
}

NKA_C_COMPAT_API NKA nka_init(int vlen, int mvec, double vtol, double (*dp)(int, double *, double *)) {
  NKA a = 0;
  nka_compat_check_(nka_hip_create(&a, vlen, mvec, vtol, NKA_HIP_FLAVOR_C, 0, 0), "nka_init");
  if (dp != 0) {                                   /* .c:227-231 */
    union { void *p; nka_compat_dp_fn_ f; } u;
    u.f = dp;
    nka_compat_check_(nka_hip_set_host_dot(a, nka_compat_dp_trampoline_, u.p), "nka_init(dp)");
  }
  return a;
}
NKA_C_COMPAT_API void nka_delete(NKA a) { nka_hip_destroy(a); }
NKA_C_COMPAT_API void nka_accel_update(NKA a, double *f) { nka_compat_check_(nka_hip_accel_update_host(a, f), "nka_accel_update"); }
NKA_C_COMPAT_API void nka_accel_update_dev(NKA a, double *f_dev) { nka_compat_check_(nka_hip_accel_update(a, f_dev), "nka_accel_update_dev"); }
NKA_C_COMPAT_API void nka_restart(NKA a) { nka_compat_check_(nka_hip_restart(a), "nka_restart"); }
NKA_C_COMPAT_API void nka_relax(NKA a) { nka_compat_check_(nka_hip_relax(a), "nka_relax"); }
NKA_C_COMPAT_API int nka_num_vec(NKA a) { return nka_hip_num_vec(a); }
NKA_C_COMPAT_API int nka_max_vec(NKA a) { return nka_hip_max_vec(a); }
NKA_C_COMPAT_API int nka_vec_len(NKA a) { return (int)nka_hip_vec_len(a); }
NKA_C_COMPAT_API double nka_vec_tol(NKA a) { return nka_hip_vec_tol(a); }

#endif
