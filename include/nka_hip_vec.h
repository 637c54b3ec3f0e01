/*
 * nka_hip_vec.h -- device implementations of the hooks of the reference's ABSTRACT VECTOR class (libnka_hip.so).
 *
 * The boundary of the abstract-vector flavour: what a concrete `vector` type
 * (/root/reference/src-F08-vector/vector_class.F90:90-228; model grid_vector_type.F90:44-197) calls from its deferred
 * procedures when its data lives on the GPU, the batched / fused forms behind the optional hooks of
 * nka_amd/fortran/vector/vector_class.F90, and the parallel-aware reductions.  Used by hip_block_vector_type.F90 and
 * hip_grid_vector_type.F90; a caller of the array flavours never needs this header.
 */
#ifndef NKA_HIP_VEC_H
#define NKA_HIP_VEC_H

#include "nka_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- vector primitives for the abstract-vector hooks -------------------- */
/* Device implementations of the deferred procedures a concrete `vector` must
 * supply (F08V vector_class.F90:92-108; model: grid_vector_type.F90:86-197).
 * x, y, z are device pointers to n doubles; `ws` is a workspace obtained from
 * nka_hip_vec_workspace_create (holds the reduction partials).  Elementwise
 * results are rounded exactly like the Fortran expressions they replace. */
typedef struct nka_hip_vec_ws *nka_hip_vec_ws_t;
int nka_hip_vec_workspace_create(nka_hip_vec_ws_t *out, int32_t device, void *stream);
int nka_hip_vec_workspace_destroy(nka_hip_vec_ws_t ws);
int nka_hip_vec_alloc(nka_hip_vec_ws_t ws, int64_t n, double **out_dev);      /* clone: allocate */
int nka_hip_vec_free(nka_hip_vec_ws_t ws, double *dev);
int nka_hip_vec_copy(nka_hip_vec_ws_t ws, int64_t n, double *dst, const double *src);          /* copy_   */
int nka_hip_vec_setval(nka_hip_vec_ws_t ws, int64_t n, double *x, double val);                  /* setval  */
int nka_hip_vec_scale(nka_hip_vec_ws_t ws, int64_t n, double *x, double a);                     /* scale: a*x */
int nka_hip_vec_update1(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x);  /* a*x + z */
int nka_hip_vec_update2(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x, double b); /* a*x + b*z */
int nka_hip_vec_update3(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x,
                        double b, const double *y);                                             /* a*x + b*y + z */
int nka_hip_vec_update4(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x,
                        double b, const double *y, double c);                                   /* a*x + b*y + c*z */
int nka_hip_vec_dot(nka_hip_vec_ws_t ws, int64_t n, const double *x, const double *y, double *host_result); /* dot_ */
int nka_hip_vec_norm2(nka_hip_vec_ws_t ws, int64_t n, const double *x, double *host_result);    /* norm2   */
/* Batched forms (optional overrides of vector%dot_many / vector%update_many,
 * nka_amd/fortran/vector/vector_class.F90): ys / xs are HOST arrays of `count`
 * device pointers.  dot_many: vals[j] = <x, ys[j]> with x read once per 24
 * vectors.  update_many: z <- (a[j]*xs[j] + b[j]*ys[j]) + z for j = 0..count-1
 * in order -- the rounding of `count` successive update3_ calls -- with z read
 * and written once per 24 pairs. */
int nka_hip_vec_dot_many(nka_hip_vec_ws_t ws, int64_t n, const double *x, const double *const *ys,
                         int32_t count, double *host_vals);
/* Both rows of the Gram update in one pass: vals0[j] = <x0, ys[j]>,
 * vals1[j] = <x1, ys[j]>, *cross = <x0, x1> (override of vector%dot_pair_many). */
int nka_hip_vec_dot_pair_many(nka_hip_vec_ws_t ws, int64_t n, const double *x0, const double *x1,
                              const double *const *ys, int32_t count, double *host_vals0,
                              double *host_vals1, double *host_cross);
int nka_hip_vec_update_many(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a,
                            const double *const *xs, const double *b, const double *const *ys, int32_t count);
/* z <- a[j]*xs[j] + z for j = 0..count-1 in order (override of vector%axpy_many). */
int nka_hip_vec_axpy_many(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a,
                          const double *const *xs, int32_t count);
/* Fused stages of the vector-flavour update (optional overrides of
 * vector%update_norm2 / scale_dot_pair_many / update_many_keep / axpy_many_keep;
 * each default body is the reference's own hook sequence, F08V:237-238, 255-264 +
 * 347, 336 + 374 + 382).  Elementwise results are rounded like those hook calls.
 *   update_norm2:        *host_norm = ||a*x + z||_2 ; store != 0: z <- a*x + z, store == 0:
 *                        z untouched -- the update is then applied by the next stage
 *   scale_dot_pair_many: [pre != 0: w <- pre_a*f + w ;] w <- a*w ; v <- a*v (subtract:
 *                        v <- (-1)*w + v) ; then with the new w: vals_w[j] = <w,ys[j]>,
 *                        vals_f[j] = <f,ys[j]>, *cross = <f,w>
 *   update_many_keep:    keep_in <- z ; z <- (a[j]*xs[j] + b[j]*ys[j]) + z in order ;
 *                        keep_out <- z          (keep_in / keep_out may be NULL)
 *   axpy_many_keep:      the same with z <- a[j]*xs[j] + z
 * With the norm stage deferring its store an update of the abstract path moves
 * 8n(11+3m) bytes in 3 passes: exactly the contract's figure. */
int nka_hip_vec_update_norm2(nka_hip_vec_ws_t ws, int64_t n, double *z, double a, const double *x,
                             int32_t store, double *host_norm);
int nka_hip_vec_scale_dot_pair_many(nka_hip_vec_ws_t ws, int64_t n, double *w, double *v, double a,
                                    int32_t subtract, int32_t pre, double pre_a, const double *f,
                                    const double *const *ys, int32_t count, double *host_vals_w,
                                    double *host_vals_f, double *host_cross);
int nka_hip_vec_update_many_keep(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a,
                                 const double *const *xs, const double *b, const double *const *ys,
                                 int32_t count, double *keep_in, double *keep_out);
int nka_hip_vec_axpy_many_keep(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a,
                               const double *const *xs, int32_t count, double *keep_in, double *keep_out);
/* The scale-and-dot stage as a PURE READ, and the combine stage normalising the new pair itself
 * ("pending pair": entry 0 of its lists is the raw pair the first call left untouched):
 *   dot_pair_many_scaled:   with w' = a*(pre_a*f + w) [pre != 0; else a*w] formed in registers only:
 *                           vals_w[j] = <w',ys[j]>, vals_f[j] = <f,ys[j]>, *cross = <f,w'> ; any count (balanced groups of <= 24)
 *   update_many_keep_pend:  update_many_keep with xs[0] = w, ys[0] = v rewritten on the way as
 *                           w <- a*(pre_a*z_in + w) [pre], v <- a*v [, subtract: v <- (-1)*w + v]
 *   axpy_many_keep_pend:    axpy_many_keep for compact storage: xs[0] = v, pend_w = w, subtract implied
 * Same expressions as scale_dot_pair_many, hence the same bits; the new pair is read raw once more by
 * the combine instead of being written and re-read normalised: 8n(10+3m) bytes (8n(8+3m) with the fused
 * norm stage nka_hip_vec_diff_norm_dot_pair_many), the scale-and-dot
 * stage without a store stream. */
int nka_hip_vec_dot_pair_many_scaled(nka_hip_vec_ws_t ws, int64_t n, const double *w, double a, int32_t pre,
                                     double pre_a, const double *f, const double *const *ys, int32_t count,
                                     double *host_vals_w, double *host_vals_f, double *host_cross);
/* The norm stage and the scale-and-dot stage as ONE pure-read pass (override of
 * vector%update_norm2_dots): with d = a*x + z formed in registers only,
 *   *host_dd = <d,d>, vals_z[j] = <d,ys[j]>, vals_x[j] = <x,ys[j]>, *cross = <x,d>     (RAW sums; any count: balanced groups of <= 24 vectors,
 *                           each forming d in registers again)
 * The accelerator takes s = sqrt(<d,d>) and scales the d-sums by 1/s itself -- the Gram row of the normalised
 * pair as fl(<d,w_k>/s) instead of the sum of fl(d_i/s)*w_k,i, like pass PA of the array flavours: last-bit
 * differences, decisions and tolerance unaffected -- and hands the whole pending normalisation to the combine
 * stage (update_many_keep_pend / axpy_many_keep_pend with pre): 8n(8+3m) bytes and TWO reductions per update. */
int nka_hip_vec_diff_norm_dot_pair_many(nka_hip_vec_ws_t ws, int64_t n, const double *z, double a, const double *x,
                                        const double *const *ys, int32_t count, double *host_dd, double *host_vals_z,
                                        double *host_vals_x, double *host_cross);
int nka_hip_vec_update_many_keep_pend(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a,
                                      const double *const *xs, const double *b, const double *const *ys,
                                      int32_t count, double *keep_in, double *keep_out, double pend_a,
                                      int32_t pend_pre, double pend_pre_a, int32_t pend_subtract);
int nka_hip_vec_axpy_many_keep_pend(nka_hip_vec_ws_t ws, int64_t n, double *z, const double *a,
                                    const double *const *xs, int32_t count, double *keep_in, double *keep_out,
                                    double *pend_w, double pend_a, int32_t pend_pre, double pend_pre_a);
/* ---- parallel-aware reductions of the device vector types ----------------
 * The vector flavour of the reference is distributed THROUGH the vector class: "the
 * implementation of the vector base class reduction methods will necessarily be
 * parallel-aware" (src-F08-vector/README.md:16-22).  For the device vector types of this
 * build (hip_block_vector, hip_grid_vector) that means: every sum a reduction hands back
 * to the host -- dot_, norm2 (before its square root), dot_many, dot_pair_many and the 1
 * resp. 2L+1 sums of the fused stages update_norm2 / scale_dot_pair_many -- is first
 * summed over all ranks.  The hooks hang on the WORKSPACE the vectors share:
 *   nka_hip_vec_set_allreduce       fn sums `count` doubles at a DEVICE address in place,
 *                                   ordered on the given hipStream_t (the type of
 *                                   nka_hip_set_allreduce); the sums are written there in a
 *                                   layout that depends only on the list length, never on
 *                                   the kernel variant a rank happened to take
 *   nka_hip_vec_comm_init_rank      built-in: an RCCL all-reduce on the workspace stream
 *                                   (unique id from nka_hip_comm_unique_id)
 *   nka_hip_vec_set_host_allreduce  fn sums `count` doubles in HOST memory in place, after
 *                                   the stream has been synchronised -- the natural place
 *                                   for an MPI_Allreduce of a caller that has no device-
 *                                   aware communication library
 * Either kind, both, or none may be installed (device hook first, then the host hook).
 * The result must carry the same bits on every rank: the Gram/Cholesky matrix and the
 * lists of the vector flavour live on the host of each rank and take the drop decisions
 * independently (F08V:269-321).  A rank whose slice is empty (n = 0) still takes part in
 * every collective.  A failing hook makes the reduction return NKA_HIP_ECOMM.
 * nka_hip_vec_allreduce_now runs the installed hooks once on `count` host values
 * (count <= 50), so that a launcher can prove the communicator before the first update. */
typedef int (*nka_hip_host_allreduce_fn)(void *ctx, double *host_vals, int32_t count);
/* Sums of the vector hooks in the REFERENCE'S ORDER (the abstract-vector counterpart of nka_hip_set_sum_order).  With
 * NKA_HIP_SUMS_REFERENCE_ORDER nka_hip_vec_dot -- and nka_hip_vec_norm2, which is its square root -- sums element after
 * element, one rounding per product and per addition, as `sum(x*y)` over the elements does
 * (/root/reference/src-F08-vector/grid_vector_type.F90:170-197); the batched and stage reductions of this library sum in
 * blocks and return NKA_HIP_ESTATE then: a vector type that honours the switch (hip_block_vector, hip_grid_vector) runs
 * the default bodies of the batched / stage hooks, i.e. the reference's own sequence of deferred hook calls
 * (vector_class.F90), and the vector flavour of the accelerator returns the bits of the reference on the same vector
 * type.  Works with the parallel-aware reductions too (ordered partial sums per rank, summed by the hook -- the
 * reference's own parallel contract).  n sequential additions per dot product: a validation mode.
 * NKA_HIP_SUMS_BLOCKED (= _AUTO, the default): the fast reductions.  NKA_HIP_SUMS_BLOCKED_ROUNDED: the fast reductions, but the device
 * vector types keep the norm stage a pass of its own, so that the Gram row is summed on the ROUNDED pair (see nka_hip_set_sum_order). */
int nka_hip_vec_set_sum_order(nka_hip_vec_ws_t ws, int32_t order);
int nka_hip_vec_get_sum_order(nka_hip_vec_ws_t ws);      /* NKA_HIP_SUMS_REFERENCE_ORDER or NKA_HIP_SUMS_BLOCKED; <0 on error */
int nka_hip_vec_set_allreduce(nka_hip_vec_ws_t ws, nka_hip_allreduce_fn fn, void *ctx);
int nka_hip_vec_set_host_allreduce(nka_hip_vec_ws_t ws, nka_hip_host_allreduce_fn fn, void *ctx);
int nka_hip_vec_comm_init_rank(nka_hip_vec_ws_t ws, const void *id128, int32_t nranks, int32_t rank);
int nka_hip_vec_comm_destroy(nka_hip_vec_ws_t ws);
int nka_hip_vec_allreduce_now(nka_hip_vec_ws_t ws, double *host_vals, int32_t count);

int nka_hip_vec_h2d(nka_hip_vec_ws_t ws, int64_t n, double *dst_dev, const double *src_host);
int nka_hip_vec_d2h(nka_hip_vec_ws_t ws, int64_t n, double *dst_host, const double *src_dev);

#ifdef __cplusplus
}
#endif
#endif /* NKA_HIP_VEC_H */
