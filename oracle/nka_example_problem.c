/*
 * oracle/nka_example_problem.c -- the caller of the hot path in BASELINE config 1,
 * restated in C.  TEST INFRASTRUCTURE ONLY.
 *
 * Finite-volume discretisation of  -div((a+u) grad u) = q  on the unit square
 * with u = 0 on the boundary, solved by the fixed-point iteration
 *     r <- SSOR(residual(u));  [accelerate r];  u <- u - r
 * following /root/reference/src-F08/nka_example.F90:
 *   system init      :86-101     residual        :103-120
 *   update_system    :122-145    pc_ssor         :147-179
 *   solve loop       :226-256    defaults        :273-274 (nx=ny=50, nsweep=2,
 *                                                 a=0.02, omega=1.4)
 * The accelerator is a callback so the same driver runs the CPU oracle, the
 * compiled reference (oracle/_ref) and the HIP library (through its C ABI).
 * Arrays are column-major with explicit index macros so the loop order -- and
 * hence every rounding -- is the reference's.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef void (*nka_example_accel_fn)(void *ctx, double *r, int64_t n);

typedef struct {
  int nx, ny;
  double a, hx, hy;
  double *ax; /* (nx+1) x ny   : ax(j,k), j=1..nx+1, k=1..ny   */
  double *ay; /* nx x (ny+1)   : ay(j,k), j=1..nx,   k=1..ny+1 */
  double *ac; /* nx x ny */
  double *q;  /* nx x ny */
} example_system;

#define AX(s, j, k) ((s)->ax[((j)-1) + (size_t)((k)-1) * ((s)->nx + 1)])
#define AY(s, j, k) ((s)->ay[((j)-1) + (size_t)((k)-1) * (s)->nx])
#define AC(s, j, k) ((s)->ac[((j)-1) + (size_t)((k)-1) * (s)->nx])
#define QQ(s, j, k) ((s)->q[((j)-1) + (size_t)((k)-1) * (s)->nx])
#define R2(r, s, j, k) ((r)[((j)-1) + (size_t)((k)-1) * (s)->nx])
/* extended arrays (0..nx+1, 0..ny+1) */
#define UE(u, s, j, k) ((u)[(j) + (size_t)(k) * ((s)->nx + 2)])

static void system_init(example_system *s, double a, int nx, int ny) {
  s->a = a;
  s->nx = nx;
  s->ny = ny;
  s->hx = 1.0 / nx;
  s->hy = 1.0 / ny;
  s->ax = (double *)calloc((size_t)(nx + 1) * ny, sizeof(double));
  s->ay = (double *)calloc((size_t)nx * (ny + 1), sizeof(double));
  s->ac = (double *)calloc((size_t)nx * ny, sizeof(double));
  s->q = (double *)malloc((size_t)nx * ny * sizeof(double));
  for (size_t i = 0; i < (size_t)nx * ny; i++) s->q[i] = 1.0;
}

static void system_free(example_system *s) {
  free(s->ax);
  free(s->ay);
  free(s->ac);
  free(s->q);
}

/* harmonic-mean face coefficients: nka_example.F90:122-145 */
static void update_system(example_system *s, const double *uext) {
  const int nx = s->nx, ny = s->ny;
  memset(s->ax, 0, (size_t)(nx + 1) * ny * sizeof(double));
  memset(s->ay, 0, (size_t)nx * (ny + 1) * sizeof(double));
  const double hx2 = s->hx * s->hx, hy2 = s->hy * s->hy;
  for (int k = 1; k <= ny; k++)
    for (int j = 1; j <= nx; j++) {
      double t = 1.0 / (s->a + UE(uext, s, j, k));
      AX(s, j, k) = AX(s, j, k) + (t * hx2);
      AX(s, j + 1, k) = AX(s, j + 1, k) + (t * hx2);
      AY(s, j, k) = AY(s, j, k) + (t * hy2);
      AY(s, j, k + 1) = AY(s, j, k + 1) + (t * hy2);
    }
  for (size_t i = 0; i < (size_t)(nx + 1) * ny; i++) s->ax[i] = 2.0 / s->ax[i];
  for (size_t i = 0; i < (size_t)nx * (ny + 1); i++) s->ay[i] = 2.0 / s->ay[i];
  for (int k = 1; k <= ny; k++)
    for (int j = 1; j <= nx; j++)
      AC(s, j, k) = AX(s, j, k) + AX(s, j + 1, k) + AY(s, j, k) + AY(s, j, k + 1);
}

/* nka_example.F90:103-120 */
static void residual(example_system *s, const double *uext, double *r) {
  update_system(s, uext);
  for (int k = 1; k <= s->ny; k++)
    for (int j = 1; j <= s->nx; j++)
      R2(r, s, j, k) = AC(s, j, k) * UE(uext, s, j, k) - AX(s, j, k) * UE(uext, s, j - 1, k) -
                       AX(s, j + 1, k) * UE(uext, s, j + 1, k) - AY(s, j, k) * UE(uext, s, j, k - 1) -
                       AY(s, j, k + 1) * UE(uext, s, j, k + 1) - QQ(s, j, k);
}

/* nka_example.F90:147-179 */
static void pc_ssor(const example_system *s, int nsweep, double omega, double *r, double *z) {
  const int nx = s->nx, ny = s->ny;
  memset(z, 0, (size_t)(nx + 2) * (ny + 2) * sizeof(double));
  for (int it = 0; it < nsweep; it++) {
    for (int k = 1; k <= ny; k++)
      for (int j = 1; j <= nx; j++)
        UE(z, s, j, k) = (1 - omega) * UE(z, s, j, k) +
                         omega * (R2(r, s, j, k) + AX(s, j, k) * UE(z, s, j - 1, k) +
                                  AX(s, j + 1, k) * UE(z, s, j + 1, k) + AY(s, j, k) * UE(z, s, j, k - 1) +
                                  AY(s, j, k + 1) * UE(z, s, j, k + 1)) / AC(s, j, k);
    for (int k = ny; k >= 1; k--)
      for (int j = nx; j >= 1; j--)
        UE(z, s, j, k) = (1 - omega) * UE(z, s, j, k) +
                         omega * (R2(r, s, j, k) + AX(s, j, k) * UE(z, s, j - 1, k) +
                                  AX(s, j + 1, k) * UE(z, s, j + 1, k) + AY(s, j, k) * UE(z, s, j, k - 1) +
                                  AY(s, j, k + 1) * UE(z, s, j, k + 1)) / AC(s, j, k);
  }
  for (int k = 1; k <= ny; k++)
    for (int j = 1; j <= nx; j++) R2(r, s, j, k) = UE(z, s, j, k);
}

static double norm2(const double *x, size_t n) {
  double t = 0.0;
  for (size_t i = 0; i < n; i++) t += x[i] * x[i];
  return sqrt(t);
}

/*
 * Run the solve loop (nka_example.F90:226-256).  `accel` may be NULL (no
 * acceleration).  On return rnorm[0..niter] holds the residual norms
 * (rnorm[0] = initial), and the function value is niter, the index of the
 * last iteration performed (<= maxitr).  rnorm must have maxitr+1 entries.
 * u_out, if not NULL, receives the nx*ny interior solution values.
 */
int nka_example_solve(int nx, int ny, double a, int nsweep, double omega, int maxitr, double tol,
                      nka_example_accel_fn accel, void *accel_ctx, double *rnorm, double *u_out) {
  example_system s;
  system_init(&s, a, nx, ny);
  const size_t n = (size_t)nx * ny, next = (size_t)(nx + 2) * (ny + 2);
  double *uext = (double *)calloc(next, sizeof(double));
  double *z = (double *)calloc(next, sizeof(double));
  double *r = (double *)calloc(n, sizeof(double));
  residual(&s, uext, r);
  rnorm[0] = norm2(r, n);
  int itr;
  for (itr = 1; itr <= maxitr; itr++) {
    pc_ssor(&s, nsweep, omega, r, z);
    if (accel) accel(accel_ctx, r, (int64_t)n);
    for (int k = 1; k <= ny; k++)
      for (int j = 1; j <= nx; j++) UE(uext, &s, j, k) = UE(uext, &s, j, k) - R2(r, &s, j, k);
    residual(&s, uext, r);
    rnorm[itr] = norm2(r, n);
    if (rnorm[itr] < tol * rnorm[0]) break;
  }
  if (itr > maxitr) itr = maxitr;
  if (u_out)
    for (int k = 1; k <= ny; k++)
      for (int j = 1; j <= nx; j++) R2(u_out, &s, j, k) = UE(uext, &s, j, k);
  free(uext);
  free(z);
  free(r);
  system_free(&s);
  return itr;
}
