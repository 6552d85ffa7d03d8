/* TEST INFRASTRUCTURE — CPU oracle for the SLQ hot path. NOT product code.
 *
 * A plain-C restatement of the reference's algorithm (peekxc/primate @ 2025-01-03): the Lanczos
 * recurrence + modified Gram-Schmidt of src/primate/include/lanczos.h:43-149, the operator
 * matvecs of src/primate/include/eigen_operators.h, Golub-Welsch quadrature
 * (src/primate/integrate.py:57-76), FTTR (src/primate/fttr.py) and the spectral-function registry
 * (src/primate/special.py:78-107).
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — as the
 * checker / the timed CPU baseline, never as the thing shipped. primate_amd/ must not import,
 * link or call anything in this directory.
 *
 * Parity status: PINNED. The reference's compiled path is unbuildable here (Eigen is an empty
 * submodule) but its Python is importable; tests/golden/make_golden.py ran the
 * reference's own NumPy twins (src/primate/lanczos.py:196-238), quadrature(), isotropic(),
 * MatrixFunction and hutch() in this container and the captured vectors under tests/golden/ are
 * what tests/test_oracle_golden.py checks this file against.
 */
#ifndef SLQ_ORACLE_H
#define SLQ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_OP_CSR = 0, ORACLE_OP_CSC = 1, ORACLE_OP_DENSE = 2, ORACLE_OP_CALLBACK = 3 };

/* Spectral functions (src/primate/special.py:78-107). params meaning per id:
 *   EXP: {t}            f(x) = exp(t x)
 *   SMOOTHSTEP: {a, b}  y = clip((x-a)/d, 0, 1), d = b-a (1 if a == b); f = 3y^2 - 2y^3
 *   STEP: {c, nonneg}   x' = |x| if nonneg else x;  f = 0 if x' < c else 1   (numrank = {1e-6, 1})
 *   SOFTSIGN: {q}       x' = clip(x,-1,1); f = sum_{i<=q} x'(1-x'^2)^i J_i
 *   LOG:                f(x) = log(max(x, eps_f64))
 */
enum {
  ORACLE_FUN_IDENTITY = 0,
  ORACLE_FUN_ABS = 1,
  ORACLE_FUN_SQRT = 2,
  ORACLE_FUN_LOG = 3,
  ORACLE_FUN_INV = 4,
  ORACLE_FUN_EXP = 5,
  ORACLE_FUN_SMOOTHSTEP = 6,
  ORACLE_FUN_STEP = 7,
  ORACLE_FUN_SOFTSIGN = 8
};

#define ORACLE_DECLARE(F, S)                                                                      \
  typedef struct {                                                                                \
    int32_t kind;                                                                                 \
    int32_t _pad;                                                                                 \
    int64_t nrows, ncols;                                                                         \
    const int32_t *ptr;                                                                           \
    const int32_t *ind;                                                                           \
    const F *vals;                                                                                \
    int64_t lda;                                                                                  \
    int (*matvec)(void *ctx, const F *x, F *y);                                                   \
    void *ctx;                                                                                    \
  } oracle_operator##S;                                                                           \
  void oracle_csr_matvec##S(int64_t n, const int32_t *rowptr, const int32_t *colind,              \
                            const F *vals, const F *x, F *y);                                     \
  void oracle_csc_matvec##S(int64_t nrows, int64_t ncols, const int32_t *colptr,                  \
                            const int32_t *rowind, const F *vals, const F *x, F *y);              \
  void oracle_dense_matvec##S(int64_t nrows, int64_t ncols, const F *A, int64_t lda, const F *x,  \
                              F *y);                                                              \
  void oracle_orth_vector##S(F *v, const F *U, int64_t n, int m, int start_idx, int p,            \
                             int reverse);                                                        \
  int oracle_lanczos_recurrence##S(const oracle_operator##S *A, F *q, int deg, F rtol, int orth,  \
                                   F *alpha, F *beta, F *V, int64_t ncv);                         \
  int oracle_tridiag_ql##S(int n, F *d, F *e, F *z, int zrows, int maxiter);                      \
  int oracle_quadrature_gw##S(int deg, const F *d, const F *e, F *nodes, F *weights, F *work);    \
  void oracle_fttr##S(const F *theta, const F *alpha, const F *beta, int n, int k, F *weights,    \
                      F *p);                                                                      \
  int oracle_quad_form##S(const oracle_operator##S *A, const F *x, int deg, F rtol, int orth,     \
                          int fun_id, const double *fun_params, F *alpha, F *beta, F *Q,          \
                          int64_t ncv, F *nodes, F *weights, F *work, double *out);               \
  int oracle_quad_batch##S(const oracle_operator##S *A, const F *X, int64_t ldx, int nprobes,     \
                           int deg, F rtol, int orth, int fun_id, const double *fun_params,       \
                           int fresh_q, int nthreads, double *quad_out, F *nodes_out,            \
                           F *weights_out, int *steps_out);

#ifndef SLQ_ORACLE_IMPL
ORACLE_DECLARE(float, _f32)
ORACLE_DECLARE(double, _f64)
#endif

double oracle_apply_fun(int fun_id, const double *params, double x);

#ifdef __cplusplus
}
#endif
#endif
