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
 */
#ifndef NKA_C_COMPAT_H
#define NKA_C_COMPAT_H

#include <stdio.h>
#include <stdlib.h>

#include "nka_hip.h"

typedef nka_hip_t NKA;

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

static inline NKA nka_init(int vlen, int mvec, double vtol, double (*dp)(int, double *, double *)) {
  NKA a = 0;
  nka_compat_check_(nka_hip_create(&a, vlen, mvec, vtol, NKA_HIP_FLAVOR_C, 0, 0), "nka_init");
  if (dp != 0) {                                   /* .c:227-231 */
    union { void *p; nka_compat_dp_fn_ f; } u;
    u.f = dp;
    nka_compat_check_(nka_hip_set_host_dot(a, nka_compat_dp_trampoline_, u.p), "nka_init(dp)");
  }
  return a;
}
static inline void nka_delete(NKA a) { nka_hip_destroy(a); }
static inline void nka_accel_update(NKA a, double *f) { nka_compat_check_(nka_hip_accel_update_host(a, f), "nka_accel_update"); }
static inline void nka_accel_update_dev(NKA a, double *f_dev) { nka_compat_check_(nka_hip_accel_update(a, f_dev), "nka_accel_update_dev"); }
static inline void nka_restart(NKA a) { nka_compat_check_(nka_hip_restart(a), "nka_restart"); }
static inline void nka_relax(NKA a) { nka_compat_check_(nka_hip_relax(a), "nka_relax"); }
static inline int nka_num_vec(NKA a) { return nka_hip_num_vec(a); }
static inline int nka_max_vec(NKA a) { return nka_hip_max_vec(a); }
static inline int nka_vec_len(NKA a) { return (int)nka_hip_vec_len(a); }
static inline double nka_vec_tol(NKA a) { return nka_hip_vec_tol(a); }

#endif
