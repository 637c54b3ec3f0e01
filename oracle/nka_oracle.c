/*
 * oracle/nka_oracle.c -- CPU restatement of the reference NKA accelerator.
 * TEST INFRASTRUCTURE ONLY (see nka_oracle.h).  Compile with
 *   gcc -O2 -ffp-contract=off
 * so that no multiply-add is fused: the reference build on x86-64 has no FMA
 * (SURVEY.md 7.2), and the drop decisions are compared bit for bit.
 *
 * Slot numbering is the Fortran one: slots 1..mvec+1, 0 terminates a list.
 * Arrays are allocated with one unused leading entry so slot numbers index
 * them directly.
 */
#include "nka_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ONE source, TWO instruments.  Compiled as it stands this file is the restatement in IEEE double
 * (real_t = double: the expressions, their order and their rounding are the reference's; pinned bit for
 * bit to the compiled reference by tests/test_oracle_golden.py).  oracle/nka_oracle_exact.c includes it
 * with NKA_ORACLE_EXTENDED defined: the SAME statements and the SAME list logic in x87 extended
 * precision (real_t = long double, 64-bit significand, unit roundoff 5.4e-20; every stored vector,
 * inner product, factor entry and coefficient) under the names nka_oraclex_* -- the "exact" trajectory
 * against which the tests measure err(reference) and err(device) (tests/parity_util.py). */
#ifdef NKA_ORACLE_EXTENDED
typedef long double real_t;
#define R_SQRT sqrtl
#define NM(x) nka_oraclex_##x
typedef struct nka_oraclex nka_oraclex;
#define nka_oracle nka_oraclex
#else
typedef double real_t;
#define R_SQRT sqrt
#define NM(x) nka_oracle_##x
#endif

typedef real_t (*real_dot_fn)(void *ctx, int64_t n, const real_t *x, const real_t *y);

struct nka_oracle {
  int subspace, pending;
  int64_t vlen;
  int mvec, nslot; /* nslot = mvec + 1 */
  int flavor;
  real_t vtol;
  real_dot_fn dot;
  void *dot_ctx;
  real_t *v, *w;   /* nslot vectors of vlen each, slot k at (k-1)*vlen */
  real_t *h;       /* (nslot+1)^2, h[i*(nslot+1)+j] is the reference's h(i,j) */
  real_t *c;       /* nslot+1 */
  int first, last, free_;
  int *next, *prev; /* nslot+1 */
  /* ERROR ATTRIBUTION (never set by the parity tests' reference runs): the Gram row of the normalised new
   * vector and the first projection from the RAW sums <d,w_k>/s, <f,d>/s with d = w1 - f, as the device
   * path forms them (deviation ii), instead of summing fl(d_i/s)*w_k,i. */
  int gram_from_raw_sums;
  real_t *fx;      /* extended build: the caller's f in extended precision */
};

#define H(a, i, j) ((a)->h[(size_t)(i) * ((a)->nslot + 1) + (j)])
#define WV(a, k) ((a)->w + (size_t)((k) - 1) * (size_t)(a)->vlen)
#define VV(a, k) ((a)->v + (size_t)((k) - 1) * (size_t)(a)->vlen)

/* Default dot product: a single sequential accumulation, the order of the C
 * reference's dot_product (src-C/nonlinear_krylov_accelerator.c:200-208) and of
 * an unvectorised DOT_PRODUCT intrinsic (src-F08/nka_type.F90:216-219). */
static real_t seq_dot(void *ctx, int64_t n, const real_t *x, const real_t *y) {
  (void)ctx;
  real_t s = 0.0;
  for (int64_t i = 0; i < n; i++) s += x[i] * y[i];
  return s;
}

/* src-F08/nka_type.F90:422-436 */
void NM(restart)(nka_oracle *a) {
  a->subspace = 0;
  a->pending = 0;
  a->first = 0;
  a->last = 0;
  a->free_ = 1;
  for (int k = 1; k < a->nslot; k++) a->next[k] = k + 1;
  a->next[a->nslot] = 0;
}

/* src-F08/nka_type.F90:185-200 (init also resets vtol and dp to the defaults) */
nka_oracle *NM(init)(int64_t vlen, int mvec, int flavor) {
  if (mvec <= 0 || vlen < 0) return NULL;
  nka_oracle *a = (nka_oracle *)calloc(1, sizeof *a);
  if (!a) return NULL;
  a->vlen = vlen;
  a->mvec = mvec;
  a->nslot = mvec + 1;
  a->flavor = flavor;
  a->vtol = 0.01; /* src-F08/nka_type.F90:160 */
  a->dot = seq_dot;
  size_t nv = (size_t)a->nslot * (size_t)(vlen > 0 ? vlen : 1);
  a->v = (real_t *)malloc(nv * sizeof(real_t));
  a->w = (real_t *)malloc(nv * sizeof(real_t));
  a->h = (real_t *)calloc((size_t)(a->nslot + 1) * (a->nslot + 1), sizeof(real_t));
  a->c = (real_t *)calloc((size_t)a->nslot + 1, sizeof(real_t));
#ifdef NKA_ORACLE_EXTENDED
  a->fx = (real_t *)malloc((size_t)(vlen > 0 ? vlen : 1) * sizeof(real_t));
  if (!a->fx) {
    NM(delete)(a);
    return NULL;
  }
#endif
  a->next = (int *)calloc((size_t)a->nslot + 1, sizeof(int));
  a->prev = (int *)calloc((size_t)a->nslot + 1, sizeof(int));
  if (!a->v || !a->w || !a->h || !a->c || !a->next || !a->prev) {
    NM(delete)(a);
    return NULL;
  }
  NM(restart)(a);
  return a;
}

void NM(delete)(nka_oracle *a) {
  if (!a) return;
  free(a->v);
  free(a->w);
  free(a->h);
  free(a->c);
  free(a->next);
  free(a->prev);
  free(a->fx);
  free(a);
}

/* src-F08/nka_type.F90:202-207 */
void NM(set_vec_tol)(nka_oracle *a, double vtol) { a->vtol = vtol; }

/* src-F08/nka_type.F90:209-214 */
#ifndef NKA_ORACLE_EXTENDED
void NM(set_dot_prod)(nka_oracle *a, nka_oracle_dot_fn fn, void *ctx) {
  a->dot = fn ? fn : seq_dot;
  a->dot_ctx = ctx;
}
/* error attribution only (see struct nka_oracle and nka_oracle_probe.c) */
void nka_oracle_set_gram_from_raw_sums(nka_oracle *a, int on) { a->gram_from_raw_sums = on; }
#endif

/* src-F08/nka_type.F90:439-457 */
void NM(relax)(nka_oracle *a) {
  if (!a->pending) return;
  int dropped = a->first;
  a->first = a->next[dropped];
  if (a->first == 0)
    a->last = 0;
  else
    a->prev[a->first] = 0;
  a->next[dropped] = a->free_;
  a->free_ = dropped;
  a->pending = 0;
}

/* src-F08/nka_type.F90:221-231 */
int NM(num_vec)(const nka_oracle *a) {
  int n = 0;
  for (int k = a->first; k != 0; k = a->next[k]) n++;
  return a->pending ? n - 1 : n;
}
int NM(max_vec)(const nka_oracle *a) { return a->mvec; }
int64_t NM(vec_len)(const nka_oracle *a) { return a->vlen; }
double NM(vec_tol)(const nka_oracle *a) { return (double)a->vtol; }

/* src-F08/nka_type.F90:460-524 -- structural invariants of the two lists. */
int NM(defined)(const nka_oracle *a) {
  if (!a || a->mvec < 1 || !a->v || !a->w || !a->h || !a->next || !a->prev) return 0;
  if (a->vtol <= 0.0) return 0;
  int n = a->nslot;
  for (int k = 1; k <= n; k++)
    if (a->next[k] < 0 || a->next[k] > n) return 0;
  if (a->first < 0 || a->first > n) return 0;
  if (a->free_ < 0 || a->free_ > n) return 0;
  char *tag = (char *)calloc((size_t)n + 1, 1);
  int ok = 0;
  do {
    if (a->first == 0) {
      if (a->last != 0) break;
    } else {
      int k = a->first;
      if (a->prev[k] != 0) break;
      tag[k] = 1;
      int bad = 0;
      while (a->next[k] != 0) {
        if (a->prev[a->next[k]] != k) { bad = 1; break; }
        k = a->next[k];
        if (tag[k]) { bad = 1; break; }
        tag[k] = 1;
      }
      if (bad || a->last != k) break;
    }
    int bad = 0;
    for (int k = a->free_; k != 0; k = a->next[k]) {
      if (tag[k]) { bad = 1; break; }
      tag[k] = 1;
    }
    if (bad) break;
    ok = 1;
    for (int k = 1; k <= n; k++)
      if (!tag[k]) ok = 0;
  } while (0);
  free(tag);
  return ok;
}

/* Row-by-row Cholesky factorisation of the Gram matrix in list order with the
 * capacity drop and the dependence drops: src-F08/nka_type.F90:295-351.
 * For list entries j newer than k, h(j,k) is the raw inner product and h(k,j)
 * the factor entry; h(k,k) is the pivot.  The order of the subtractions in the
 * inner loop is part of the contract (decision parity). */
static void factor_with_drops(nka_oracle *a) {
  H(a, a->first, a->first) = 1.0;
  int k = a->next[a->first];
  int nvec = 1;
  while (k != 0) {
    nvec++;
    if (nvec > a->mvec) { /* capacity: k is necessarily the last entry */
      a->next[a->last] = a->free_;
      a->free_ = k;
      a->last = a->prev[k];
      a->next[a->last] = 0;
      break;
    }
    real_t hkk = 1.0;
    for (int j = a->first; j != k; j = a->next[j]) {
      real_t hkj = H(a, j, k);
      for (int i = a->first; i != j; i = a->next[i]) hkj = hkj - H(a, k, i) * H(a, j, i);
      hkj = hkj / H(a, j, j);
      hkk = hkk - hkj * hkj;
      H(a, k, j) = hkj;
    }
    if (hkk > a->vtol * a->vtol) {
      H(a, k, k) = R_SQRT(hkk);
    } else { /* w_k is (nearly) in the span of the newer vectors: unlink it */
      int p = a->prev[k], nx = a->next[k];
      a->next[p] = nx;
      if (nx == 0)
        a->last = p;
      else
        a->prev[nx] = p;
      a->next[k] = a->free_;
      a->free_ = k;
      k = p;
      nvec--;
    }
    k = a->next[k];
  }
  a->subspace = 1;
  a->pending = 0;
}

/* Forward then backward substitution on c (holding the right-hand side b on
 * entry, indexed by slot): src-F08/nka_type.F90:369-392. */
static void solve_normal_equations(nka_oracle *a) {
  for (int j = a->first; j != 0; j = a->next[j]) {
    real_t cj = a->c[j];
    for (int i = a->first; i != j; i = a->next[i]) cj = cj - H(a, j, i) * a->c[i];
    a->c[j] = cj / H(a, j, j);
  }
  for (int j = a->last; j != 0; j = a->prev[j]) {
    real_t cj = a->c[j];
    for (int i = a->last; i != j; i = a->prev[i]) cj = cj - H(a, i, j) * a->c[i];
    a->c[j] = cj / H(a, j, j);
  }
}

/* src-F08/nka_type.F90:406-417 */
static void prepend(nka_oracle *a, int slot) {
  a->prev[slot] = 0;
  a->next[slot] = a->first;
  if (a->first == 0)
    a->last = slot;
  else
    a->prev[a->first] = slot;
  a->first = slot;
  a->pending = 1;
}

/* src-F08/nka_type.F90:249-419 */
void NM(accel_update)(nka_oracle *a, double *f_io) {
  const int64_t n = a->vlen;
  real_t s = 0.0;
#ifdef NKA_ORACLE_EXTENDED
  real_t *f = a->fx;        /* the caller's doubles, exactly; every statement below in extended precision */
  for (int64_t i = 0; i < n; i++) f[i] = f_io[i];
#else
  double *f = f_io;
#endif
  real_t *hraw = NULL;      /* error attribution only (gram_from_raw_sums): <d,w_k> by slot, and <f,d> in entry 0 */

  if (a->pending) {
    real_t *w1 = WV(a, a->first);
    if (a->flavor == NKA_ORACLE_F08_VECTOR) {
      /* update1_: a*x + this with a = -1 (src-F08-vector/nka_type.F90:237) */
      for (int64_t i = 0; i < n; i++) w1[i] = (-1.0) * f[i] + w1[i];
    } else {
      for (int64_t i = 0; i < n; i++) w1[i] = w1[i] - f[i];
    }
    s = R_SQRT(a->dot(a->dot_ctx, n, w1, w1));
    if (s == 0.0) NM(relax)(a);
  }

  if (a->pending) {
    real_t *w1 = WV(a, a->first), *v1 = VV(a, a->first);
    if (a->gram_from_raw_sums) {
      hraw = (real_t *)calloc((size_t)a->nslot + 1, sizeof(real_t));
      hraw[0] = a->dot(a->dot_ctx, n, f, w1);
      for (int k = a->next[a->first]; k != 0; k = a->next[k]) hraw[k] = a->dot(a->dot_ctx, n, w1, WV(a, k));
    }
    if (a->flavor == NKA_ORACLE_F08_VECTOR) {
      const real_t r = 1.0 / s; /* scale(1/s): src-F08-vector/nka_type.F90:255-256 */
      for (int64_t i = 0; i < n; i++) v1[i] = r * v1[i];
      for (int64_t i = 0; i < n; i++) w1[i] = r * w1[i];
    } else {
      for (int64_t i = 0; i < n; i++) v1[i] = v1[i] / s;
      for (int64_t i = 0; i < n; i++) w1[i] = w1[i] / s;
    }
    for (int k = a->next[a->first]; k != 0; k = a->next[k])
      H(a, a->first, k) = hraw ? (a->flavor == NKA_ORACLE_F08_VECTOR ? (1.0 / s) * hraw[k] : hraw[k] / s)
                               : a->dot(a->dot_ctx, n, w1, WV(a, k));
    factor_with_drops(a);
  }

  int slot = a->free_;
  a->free_ = a->next[slot];
  memcpy(WV(a, slot), f, (size_t)n * sizeof(real_t));

  if (a->subspace) {
    const int newest = hraw ? a->first : 0;      /* (factor_with_drops never drops the first entry) */
    for (int j = a->first; j != 0; j = a->next[j])
      a->c[j] = (j == newest) ? (a->flavor == NKA_ORACLE_F08_VECTOR ? (1.0 / s) * hraw[0] : hraw[0] / s)
                              : a->dot(a->dot_ctx, n, f, WV(a, j));
    solve_normal_equations(a);
    for (int k = a->first; k != 0; k = a->next[k]) {
      const real_t ck = a->c[k];
      const real_t *wk = WV(a, k), *vk = VV(a, k);
      switch (a->flavor) {
      case NKA_ORACLE_F08_VECTOR: { /* update3_(-c,w,c,v): a*x + b*y + this */
        const real_t mck = -ck;
        for (int64_t i = 0; i < n; i++) f[i] = (mck * wk[i] + ck * vk[i]) + f[i];
        break;
      }
      case NKA_ORACLE_C: /* src-C/...c:419-424 */
        for (int64_t i = 0; i < n; i++) f[i] += ck * (vk[i] - wk[i]);
        break;
      default: /* src-F08/nka_type.F90:397 */
        for (int64_t i = 0; i < n; i++) f[i] = (f[i] - ck * wk[i]) + ck * vk[i];
      }
    }
  }
  free(hraw);

  memcpy(VV(a, slot), f, (size_t)n * sizeof(real_t));
  prepend(a, slot);
#ifdef NKA_ORACLE_EXTENDED
  for (int64_t i = 0; i < n; i++) f_io[i] = (double)f[i];   /* ONE rounding, of the result */
#endif
}

#ifndef NKA_ORACLE_EXTENDED
/* The scalar part of one update with the dot products supplied from outside
 * (what the device "solve" kernel does between the streaming passes).
 *   had_pending  -- value of `pending` at entry of the update
 *   s            -- the norm of the new difference (ignored if !had_pending)
 *   hrow_by_slot -- <w1', w_k> indexed by slot k (entries 1..mvec+1; only the
 *                   list entries after `first` are read)
 *   b_by_slot    -- <f, w_j> indexed by slot j (w_first already normalised)
 * On return a->c holds the coefficients by slot, *new_slot the slot that
 * receives the new pair, and the list has `new_slot` prepended. */
void NM(scalar_step)(nka_oracle *a, int had_pending, double s,
                            const double *hrow_by_slot, const double *b_by_slot,
                            int *new_slot) {
  (void)had_pending;
  if (a->pending && s == 0.0) NM(relax)(a);
  if (a->pending) {
    for (int k = a->next[a->first]; k != 0; k = a->next[k]) H(a, a->first, k) = hrow_by_slot[k];
    factor_with_drops(a);
  }
  int slot = a->free_;
  a->free_ = a->next[slot];
  if (a->subspace) {
    for (int j = a->first; j != 0; j = a->next[j]) a->c[j] = b_by_slot[j];
    solve_normal_equations(a);
  }
  prepend(a, slot);
  *new_slot = slot;
}

#endif /* !NKA_ORACLE_EXTENDED */

void NM(get_state)(const nka_oracle *a, int *subspace, int *pending,
                          int *first, int *last, int *free_, int *next, int *prev,
                          double *h, double *c) {
  if (subspace) *subspace = a->subspace;
  if (pending) *pending = a->pending;
  if (first) *first = a->first;
  if (last) *last = a->last;
  if (free_) *free_ = a->free_;
  for (int k = 1; k <= a->nslot; k++) {
    if (next) next[k - 1] = a->next[k];
    if (prev) prev[k - 1] = a->prev[k];
    if (c) c[k - 1] = (double)a->c[k];
  }
  if (h)
    for (int j = 1; j <= a->nslot; j++)
      for (int i = 1; i <= a->nslot; i++) h[(i - 1) + (size_t)(j - 1) * a->nslot] = (double)H(a, i, j);
}

#ifndef NKA_ORACLE_EXTENDED
const double *nka_oracle_w(const nka_oracle *a, int slot) { return WV(a, slot); }
const double *nka_oracle_v(const nka_oracle *a, int slot) { return VV(a, slot); }
#endif
