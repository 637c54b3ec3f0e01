/*
 * oracle/ref_c_shim.c -- compiles the reference's C accelerator *where it lies*
 * (the #include below pulls /root/reference/src-C/nonlinear_krylov_accelerator.c
 * into this translation unit; REF_SRC_C is set by oracle/Makefile) and adds a
 * state dump, since `struct nka_state` is private to that file
 * (src-C/nonlinear_krylov_accelerator.c:179-197).
 * TEST INFRASTRUCTURE ONLY; the product is built without it.
 *
 * The dump converts to the Fortran numbering used by the oracle and the HIP
 * library: slot = C slot + 1, end-of-list -1 -> 0.  The C code keeps the raw
 * inner product of the newest vector with k in h[first][k] (.c:323-324) and the
 * factor entry in h[k][j] (.c:347-356): the same (row, column) roles as the
 * Fortran h(first,k) / h(k,j), so only the index base differs.
 */
#include <stdlib.h>
#define _NKA_STR(x) #x
#define _NKA_XSTR(x) _NKA_STR(x)
#include _NKA_XSTR(REF_SRC_C/nonlinear_krylov_accelerator.c)

void ref_c_get_state(NKA a, int *subspace, int *pending, int *first, int *last, int *free_,
                     int *next, int *prev, double *h) {
  int n = a->mvec + 1;
  *subspace = a->subspace;
  *pending = a->pending;
  *first = a->first + 1;
  *last = a->last + 1;
  *free_ = a->free + 1;
  for (int k = 0; k < n; k++) {
    next[k] = a->next[k] + 1;
    prev[k] = a->prev[k] + 1;
  }
  /* column-major (i-1) + (j-1)*n, like nka_oracle_get_state */
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) h[i + (size_t)j * n] = a->h[i][j];
}

/* The C API fixes vtol at construction (.c:211); the Fortran flavours can change
 * it mid-stream (set_vec_tol, F08:202-207).  The drop test reads state->vtol on
 * every update (.c:362), so scenarios with a mid-stream change are replayed on
 * the C reference by writing the field. */
void ref_c_set_vec_tol(NKA a, double vtol) { a->vtol = vtol; }

const double *ref_c_w(NKA a, int slot1) { return a->w[slot1 - 1]; }
const double *ref_c_v(NKA a, int slot1) { return a->v[slot1 - 1]; }
