/* TEST INFRASTRUCTURE — CPU oracle, not product code.
 *
 * Precision-generic body of the oracle; included twice by slq_oracle.c with
 *   F      = float | double
 *   FN(x)  = x##_f32 | x##_f64
 *   F_EPS  = FLT_EPSILON | DBL_EPSILON,  F_SQRT / F_FABS / F_HYPOT = the libm forms for F
 * Every function cites the reference lines (relative to /root/reference) that it restates.
 * Nothing here is copied from the reference: the reference is Eigen/C++20 + NumPy; this is plain
 * C99 with explicit loops.
 */

/* y = A x, CSR gather form (scipy.sparse CSR @ vector; the layout the HIP path consumes). */
void FN(oracle_csr_matvec)(int64_t n, const int32_t *rowptr, const int32_t *colind,
                           const F *vals, const F *x, F *y) {
  for (int64_t i = 0; i < n; ++i) {
    F acc = (F)0;
    for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) acc += vals[p] * x[colind[p]];
    y[i] = acc;
  }
}

/* y = A x, CSC scatter form. Restates src/primate/include/eigen_operators.h:66-77: the reference
 * holds an Eigen::SparseMatrix<F> (ColMajor == CSC; pybind11 converts any SciPy sparse input with
 * csc_matrix()) and Eigen's sparse*dense product walks columns and scatters into y. */
void FN(oracle_csc_matvec)(int64_t nrows, int64_t ncols, const int32_t *colptr,
                           const int32_t *rowind, const F *vals, const F *x, F *y) {
  for (int64_t i = 0; i < nrows; ++i) y[i] = (F)0;
  for (int64_t j = 0; j < ncols; ++j) {
    const F xj = x[j];
    for (int32_t p = colptr[j]; p < colptr[j + 1]; ++p) y[rowind[p]] += vals[p] * xj;
  }
}

/* y = A x for a dense column-major matrix (src/primate/include/eigen_operators.h:24-30; Eigen's
 * DenseMatrix<F> is column-major and pybind11 re-lays NumPy input to it). Column-axpy order. */
void FN(oracle_dense_matvec)(int64_t nrows, int64_t ncols, const F *A, int64_t lda, const F *x,
                             F *y) {
  for (int64_t i = 0; i < nrows; ++i) y[i] = (F)0;
  for (int64_t j = 0; j < ncols; ++j) {
    const F xj = x[j];
    const F *col = A + j * lda;
    for (int64_t i = 0; i < nrows; ++i) y[i] += col[i] * xj;
  }
}

/* Operator plugin: the C restatement of the `LinearOperator` concept
 * (src/primate/include/linear_operator.h:25-29: matvec(const F*, F*) + shape()). */
typedef struct {
  int32_t kind; /* ORACLE_OP_* */
  int32_t _pad;
  int64_t nrows, ncols;
  const int32_t *ptr; /* rowptr (CSR) / colptr (CSC) */
  const int32_t *ind; /* colind (CSR) / rowind (CSC) */
  const F *vals;      /* nnz values, or dense column-major matrix */
  int64_t lda;
  /* ORACLE_OP_CALLBACK: restates src/primate/include/pylinop.h:32-40 (the reference re-enters
   * Python for every matvec); returns nonzero on failure. */
  int (*matvec)(void *ctx, const F *x, F *y);
  void *ctx;
} FN(oracle_operator);

static int FN(op_apply)(const FN(oracle_operator) * A, const F *x, F *y) {
  switch (A->kind) {
    case ORACLE_OP_CSR: FN(oracle_csr_matvec)(A->nrows, A->ptr, A->ind, A->vals, x, y); return 0;
    case ORACLE_OP_CSC:
      FN(oracle_csc_matvec)(A->nrows, A->ncols, A->ptr, A->ind, A->vals, x, y);
      return 0;
    case ORACLE_OP_DENSE:
      FN(oracle_dense_matvec)(A->nrows, A->ncols, A->vals, A->lda, x, y);
      return 0;
    case ORACLE_OP_CALLBACK: return A->matvec(A->ctx, x, y);
    default: return -1;
  }
}

/* Inner product. The reference's are Eigen's `dot` / `squaredNorm` (src/primate/include/lanczos.h:59-63,129,139), which Eigen
 * 3.4 evaluates packet-wise: a packet of 8 floats (AVX) per accumulator, four accumulators unrolled, the four added pairwise,
 * the eight lanes of the result by a tree, the tail element by element. fp32 (F_BLOCKED_DOT): restated in that shape - 32
 * running sums - because ONE sequential fp32 accumulator is 0.7 % off at n = 2e6 (the oracle, not the thing
 * checked, was the noisy side at configs[3]'s size), which the reference is not. fp64 keeps the sequential sum the golden
 * vectors were pinned with (its rounding is far below every tolerance in use). */
static F FN(dot)(int64_t n, const F *a, const F *b) {
#ifdef F_BLOCKED_DOT
  F acc[32];
  for (int k = 0; k < 32; ++k) acc[k] = (F)0;
  int64_t i = 0;
  for (; i + 32 <= n; i += 32)
    for (int k = 0; k < 32; ++k) acc[k] += a[i + k] * b[i + k];
  F lane[8];
  for (int l = 0; l < 8; ++l) lane[l] = (acc[l] + acc[8 + l]) + (acc[16 + l] + acc[24 + l]);
  F s = ((lane[0] + lane[1]) + (lane[2] + lane[3])) + ((lane[4] + lane[5]) + (lane[6] + lane[7]));
  for (; i < n; ++i) s += a[i] * b[i];
  return s;
#else
  F s = (F)0;
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
#endif
}

/* Modified Gram-Schmidt of v against p cyclic columns of U (n x m, column-major), starting at
 * start_idx and walking backwards when reverse != 0.
 * Restates src/primate/include/lanczos.h:43-66 (NumPy twin: src/primate/lanczos.py:196-207):
 *   tol = 2*eps*sqrt(n); a column is skipped unless ||u||^2 > tol AND |<v,u>| > tol. */
void FN(oracle_orth_vector)(F *v, const F *U, int64_t n, int m, int start_idx, int p,
                            int reverse) {
  const F tol = (F)2 * F_EPS * F_SQRT((F)n);
  const int diff = reverse ? -1 : 1;
  int i = oracle_pymod(start_idx, m);
  for (int c = 0; c < p; ++c, i = oracle_pymod(i + diff, m)) {
    const F *u = U + (int64_t)i * n;
    const F u_norm = FN(dot)(n, u, u);
    const F s_proj = FN(dot)(n, v, u);
    if (u_norm > tol && F_FABS(s_proj) > tol) {
      const F g = s_proj / u_norm;
      for (int64_t r = 0; r < n; ++r) v[r] -= g * u[r];
    }
  }
}

/* k-step Lanczos recurrence, Paige's A27 ordering.
 * Restates src/primate/include/lanczos.h:92-149 (NumPy twin: src/primate/lanczos.py:211-238):
 *   - q is the work vector and is clobbered (lanczos.h:114,127-130);
 *   - V is n x ncv column-major, used as a ring; column ncv-1 is zeroed and column 0 receives
 *     q/||q|| (lanczos.h:118-121); the other columns are NOT touched on entry (so stale columns
 *     are visible to the reorthogonalisation, exactly as in the reference);
 *   - residual_tol = sqrt(n)*rtol (lanczos.h:110); stop when beta[j+1] < tol or j+1 == deg
 *     (lanczos.h:140-142) BEFORE the next column is written;
 *   - ring rotation pos = {c, n, mod(j+2, ncv)} (lanczos.h:146-147).
 * alpha and beta must hold deg+1 entries. Returns the number of steps executed (j+1 at the
 * break), or a negative value if the operator callback failed. */
int FN(oracle_lanczos_recurrence)(const FN(oracle_operator) * A, F *q, int deg, F rtol, int orth,
                                  F *alpha, F *beta, F *V, int64_t ncv) {
  const int64_t n = A->nrows;
  const int64_t m = A->ncols;
  const F residual_tol = F_SQRT((F)n) * rtol;
  int pos[3] = {(int)ncv - 1, 0, 1};
  for (int64_t r = 0; r < n; ++r) V[(int64_t)pos[0] * n + r] = (F)0;
  {
    const F nrm = F_SQRT(FN(dot)(m, q, q));
    for (int64_t r = 0; r < n; ++r) V[r] = q[r] / nrm;
  }
  beta[0] = (F)0;
  int steps = 0;
  for (int j = 0; j < deg; ++j) {
    const F *qp = V + (int64_t)pos[0] * n;
    const F *qc = V + (int64_t)pos[1] * n;
    F *qn = V + (int64_t)pos[2] * n;
    if (FN(op_apply)(A, qc, q) != 0) return -1;
    for (int64_t r = 0; r < n; ++r) q[r] -= beta[j] * qp[r];
    alpha[j] = FN(dot)(n, qc, q);
    for (int64_t r = 0; r < n; ++r) q[r] -= alpha[j] * qc[r];
    if (orth > 0) FN(oracle_orth_vector)(q, V, n, (int)ncv, pos[1], orth, 1);
    beta[j + 1] = F_SQRT(FN(dot)(n, q, q));
    steps = j + 1;
    if (beta[j + 1] < residual_tol || (j + 1) == deg) break;
    for (int64_t r = 0; r < n; ++r) qn[r] = q[r] / beta[j + 1];
    pos[0] = pos[1];
    pos[1] = pos[2];
    pos[2] = oracle_pymod(j + 2, (int)ncv);
  }
  return steps;
}

/* Symmetric tridiagonal eigen-decomposition by implicit QL with Wilkinson shifts.
 *   d[0..n)  diagonal in / eigenvalues out (unsorted),
 *   e[0..n)  sub-diagonal in the reference's convention e[0] = 0, e[i] couples i-1 and i
 *            (src/primate/tridiag.py:25-43; src/primate/tqli.py:32-36); destroyed,
 *   z        if zrows > 0: zrows x n row-major block that receives the same plane rotations
 *            (pass the identity for eigenvectors, or a 1 x n e_1^T for first components only).
 * Role in the reference: the `tqli` fallback solver (src/primate/tqli.py:15-90) and, through the
 * public eigh_tridiag(), the thing LAPACK ?stemr computes (src/primate/tridiag.py:10-11).
 * This is a fresh statement of the textbook algorithm (Numerical Recipes §11.3 / EISPACK tql2),
 * NOT a transcription of tqli.py, whose sign() helper deviates from the textbook (SURVEY.md
 * Appendix B.6); results are pinned against scipy.linalg.eigh_tridiagonal in tests/.
 * Returns 0, or the (1-based) index of an eigenvalue that failed to converge in maxiter sweeps. */
int FN(oracle_tridiag_ql)(int n, F *d, F *e, F *z, int zrows, int maxiter) {
  if (n <= 0) return 0;
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = (F)0;
  for (int l = 0; l < n; ++l) {
    int iter = 0;
    for (;;) {
      int m = l;
      for (; m < n - 1; ++m) {
        const F dd = F_FABS(d[m]) + F_FABS(d[m + 1]);
        if (F_FABS(e[m]) <= F_EPS * dd) break;
      }
      if (m == l) break;
      if (iter++ >= maxiter) return l + 1;
      F g = (d[l + 1] - d[l]) / ((F)2 * e[l]);
      F r = F_HYPOT(g, (F)1);
      g = d[m] - d[l] + e[l] / (g + (g >= (F)0 ? F_FABS(r) : -F_FABS(r)));
      F s = (F)1, c = (F)1, p = (F)0;
      int i = m - 1;
      int underflow = 0;
      for (; i >= l; --i) {
        F f = s * e[i];
        const F b = c * e[i];
        r = F_HYPOT(f, g);
        e[i + 1] = r;
        if (r == (F)0) {
          d[i + 1] -= p;
          e[m] = (F)0;
          underflow = 1;
          break;
        }
        s = f / r;
        c = g / r;
        g = d[i + 1] - p;
        r = (d[i] - g) * s + (F)2 * c * b;
        p = s * r;
        d[i + 1] = g + p;
        g = c * r - b;
        for (int k = 0; k < zrows; ++k) {
          F *zk = z + (int64_t)k * n;
          f = zk[i + 1];
          zk[i + 1] = s * zk[i] + c * f;
          zk[i] = c * zk[i] - s * f;
        }
      }
      if (underflow) continue;
      d[l] -= p;
      e[l] = g;
      e[m] = (F)0;
    }
  }
  return 0;
}

/* Gauss quadrature rule of the Jacobi matrix T(d, e): nodes = eigenvalues ascending, weights =
 * squared first components of the normalised eigenvectors (Golub-Welsch).
 * Restates src/primate/integrate.py:57-64,70-76 with the QL solver above standing in for LAPACK.
 * d, e hold deg entries with e[0] = 0 (asserted in the reference, integrate.py:59); not modified.
 * work must hold 3*deg entries. */
int FN(oracle_quadrature_gw)(int deg, const F *d, const F *e, F *nodes, F *weights, F *work) {
  F *dd = work, *ee = work + deg, *z = work + 2 * deg;
  for (int i = 0; i < deg; ++i) {
    dd[i] = d[i];
    ee[i] = e[i];
    z[i] = (F)0;
  }
  ee[0] = (F)0;
  z[0] = (F)1;
  const int rc = FN(oracle_tridiag_ql)(deg, dd, ee, z, 1, 60);
  /* insertion sort by node, carrying the first components */
  for (int i = 1; i < deg; ++i) {
    const F dv = dd[i], zv = z[i];
    int j = i - 1;
    for (; j >= 0 && dd[j] > dv; --j) {
      dd[j + 1] = dd[j];
      z[j + 1] = z[j];
    }
    dd[j + 1] = dv;
    z[j + 1] = zv;
  }
  for (int i = 0; i < deg; ++i) {
    nodes[i] = dd[i];
    weights[i] = z[i] * z[i];
  }
  return rc;
}

/* Forward three-term-recurrence weights. Restates src/primate/fttr.py:5-29 (ortho_poly + fttr):
 * mu_0 = sum |theta[:k]|; p_0 = 1/sqrt(mu_0); p_1 = (x-a_0)p_0/b_1;
 * p_i = ((x-a_{i-1})p_{i-1} - b_{i-1}p_{i-2})/b_i; w = 1/(sum p^2)/mu_0.
 * alpha, beta hold n entries (beta[0] = 0); p is scratch of n entries. */
void FN(oracle_fttr)(const F *theta, const F *alpha, const F *beta, int n, int k, F *weights,
                     F *p) {
  F mu_0 = (F)0;
  for (int i = 0; i < k; ++i) mu_0 += F_FABS(theta[i]);
  const F mu_sqrt_rec = (F)1 / F_SQRT(mu_0);
  for (int i = 0; i < k; ++i) {
    const F x = theta[i];
    p[0] = mu_sqrt_rec;
    if (n > 1) p[1] = (x - alpha[0]) * p[0] / beta[1];
    for (int t = 2; t < n; ++t) {
      const F s = (x - alpha[t - 1]) / beta[t];
      const F u = -beta[t - 1] / beta[t];
      p[t] = s * p[t - 1] + u * p[t - 2];
    }
    F ss = (F)0;
    for (int t = 0; t < n; ++t) ss += p[t] * p[t];
    weights[i] = ((F)1 / ss) / mu_0;
  }
}

/* Built-in spectral functions, restating src/primate/special.py:78-107 (param_callable) and the
 * helpers it names: exp (:62-66), smoothstep (:33-55), step/numrank (:69-74,103-105),
 * softsign (:10-30), log clamped at eps(float64) (:89-90). params: see ORACLE_FUN_* in
 * slq_oracle.h. Evaluated in double, as NumPy does on the float64 nodes. */
double oracle_apply_fun(int fun_id, const double *params, double x);

/* One SLQ quadratic form x^T f(A) x: Lanczos, Gauss quadrature, sum f(nodes)*weights * ||x||^2.
 * Restates the loop body of MatrixFunction.quad (src/primate/operators.py:145-150). x is copied
 * (the reference passes a column view which pybind11 copies, operators.py:146-148); alpha, beta
 * (deg+1) and Q (n x ncv) are caller-owned and reused across calls exactly like the reference's
 * self._alpha/_beta/_Q (operators.py:69-77) — including their stale contents. work: n + 5*deg. */
int FN(oracle_quad_form)(const FN(oracle_operator) * A, const F *x, int deg, F rtol, int orth,
                         int fun_id, const double *fun_params, F *alpha, F *beta, F *Q,
                         int64_t ncv, F *nodes, F *weights, F *work, double *out) {
  const int64_t n = A->nrows;
  F *v = work;
  F *qwork = work + n;
  F nrm2 = (F)0;
  for (int64_t i = 0; i < n; ++i) {
    v[i] = x[i];
    nrm2 += x[i] * x[i];
  }
  /* operators.py:147 takes np.linalg.norm(xc)**2 */
  const F nrm = F_SQRT(nrm2);
  const int steps = FN(oracle_lanczos_recurrence)(A, v, deg, rtol, orth, alpha, beta, Q, ncv);
  if (steps < 0) return steps;
  const int rc = FN(oracle_quadrature_gw)(deg, alpha, beta, nodes, weights, qwork);
  double s = 0.0;
  for (int i = 0; i < deg; ++i)
    s += oracle_apply_fun(fun_id, fun_params, (double)nodes[i]) * (double)weights[i];
  *out = s * (double)(nrm * nrm);
  return rc;
}
