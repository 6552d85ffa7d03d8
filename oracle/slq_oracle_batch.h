/* TEST INFRASTRUCTURE — CPU oracle, not product code. Included by slq_oracle.c per precision.
 *
 * Batched SLQ quadratic forms: the loop of MatrixFunction.quad over the columns of X
 * (src/primate/operators.py:138-151), i.e. what hutch() calls once per batch
 * (src/primate/trace.py:97,107-115).
 *
 *   X         n x nprobes column-major, leading dimension ldx (the reference's F-ordered probes,
 *             src/primate/random.py:76)
 *   fresh_q   0: alpha/beta/Q are allocated zeroed ONCE and reused across probes without
 *                clearing, exactly as MatrixFunction does with self._alpha/_beta/_Q
 *                (operators.py:69-77,148) — stale ring columns are visible to the MGS sweep.
 *                Only meaningful with nthreads == 1 (the reference is single-threaded).
 *             1: alpha/beta/Q are zeroed before every probe (what a first call sees, and what
 *                MatrixFunction._matvec enforces, operators.py:116). Order independent.
 *   nthreads  > 1 parallelises over probes with OpenMP: a generous CPU upper bound; the reference
 *             has no such mode (src/primate/include/omp_support.h:4-10).
 *   ncv is always deg, as MatrixFunction fixes it (operators.py:75-77).
 *   nodes_out / weights_out: nprobes x deg row-major, or NULL. steps_out: nprobes ints or NULL.
 */
int FN(oracle_quad_batch)(const FN(oracle_operator) * A, const F *X, int64_t ldx, int nprobes,
                          int deg, F rtol, int orth, int fun_id, const double *fun_params,
                          int fresh_q, int nthreads, double *quad_out, F *nodes_out,
                          F *weights_out, int *steps_out) {
  const int64_t n = A->nrows;
  if (deg > n) deg = (int)n; /* operators.py:68 */
  if (orth < 0 || orth > deg) orth = deg; /* operators.py:80 */
  int status = 0;
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    F *alpha = (F *)calloc((size_t)deg + 1, sizeof(F));
    F *beta = (F *)calloc((size_t)deg + 1, sizeof(F));
    F *Q = (F *)calloc((size_t)n * (size_t)deg, sizeof(F));
    F *nodes = (F *)calloc((size_t)deg, sizeof(F));
    F *weights = (F *)calloc((size_t)deg, sizeof(F));
    F *work = (F *)malloc(((size_t)n + 3 * (size_t)deg) * sizeof(F));
    if (!alpha || !beta || !Q || !nodes || !weights || !work) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
      status = -2;
    } else {
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
      for (int j = 0; j < nprobes; ++j) {
        if (fresh_q) {
          memset(alpha, 0, ((size_t)deg + 1) * sizeof(F));
          memset(beta, 0, ((size_t)deg + 1) * sizeof(F));
          memset(Q, 0, (size_t)n * (size_t)deg * sizeof(F));
        }
        const F *x = X + (int64_t)j * ldx;
        F *v = work;
        F nrm2 = (F)0;
        for (int64_t i = 0; i < n; ++i) {
          v[i] = x[i];
          nrm2 += x[i] * x[i];
        }
        const F nrm = F_SQRT(nrm2); /* operators.py:147: np.linalg.norm(xc) ** 2 */
        const int steps =
            FN(oracle_lanczos_recurrence)(A, v, deg, rtol, orth, alpha, beta, Q, (int64_t)deg);
        int rc = steps < 0 ? steps : 0;
        if (rc == 0) rc = FN(oracle_quadrature_gw)(deg, alpha, beta, nodes, weights, work + n);
        double s = 0.0;
        for (int i = 0; i < deg; ++i)
          s += oracle_apply_fun(fun_id, fun_params, (double)nodes[i]) * (double)weights[i];
        quad_out[j] = s * (double)(nrm * nrm);
        if (nodes_out) memcpy(nodes_out + (size_t)j * deg, nodes, (size_t)deg * sizeof(F));
        if (weights_out) memcpy(weights_out + (size_t)j * deg, weights, (size_t)deg * sizeof(F));
        if (steps_out) steps_out[j] = steps;
        if (rc != 0) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
          status = rc;
        }
      }
    }
    free(alpha);
    free(beta);
    free(Q);
    free(nodes);
    free(weights);
    free(work);
  }
  return status;
}
