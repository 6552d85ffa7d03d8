/* slq.h — C-ABI of the MI355X stochastic-Lanczos-quadrature engine (libslq).
 *
 * This is the drop-in boundary for the reference's native hot path. Plain pointers, sizes and
 * opaque handles only; no C++ or torch types cross it. All entry points return 0 on success or a
 * negative SLQ_E* code (never throw); slq_last_error() gives the message for the calling thread.
 *
 * Reference interfaces replaced (paths relative to the reference repo root):
 *   - primate._lanczos.lanczos(A, v, deg, rtol, orth, alpha, beta, Q)
 *       src/primate/_lanczos.cpp:88-99 (six overloads registered at :102-112)
 *       -> slq_lanczos_f64 / slq_lanczos_f32
 *   - the native kernels behind it, src/primate/include/lanczos.h:43-66 (orth_vector) and
 *       :92-149 (lanczos_recurrence)              -> the plan's device loop (slq_plan_run)
 *   - the operator plugin concept, src/primate/include/linear_operator.h:25-29
 *       (matvec(const F*, F*) + shape())          -> slq_operator (CSR device fast path,
 *       dense, host-callback fallback, GPU-resident device callback;
 *       src/primate/include/eigen_operators.h:17-104, src/primate/include/pylinop.h:16-73)
 *   - the per-probe Python loop of MatrixFunction.quad, src/primate/operators.py:138-151,
 *       with integrate.quadrature (src/primate/integrate.py:57-76) and the LAPACK call in
 *       src/primate/tridiag.py:10-11               -> slq_quad_batch (one call for P probes)
 *   - MatrixFunction._matvec, src/primate/operators.py:102-124 -> slq_fAv_batch / slq_plan_fun_action
 *   - eigh_tridiag / eigvalsh_tridiag, src/primate/tridiag.py:25-62 -> slq_eigh_tridiag_batch
 *   - random.isotropic, src/primate/random.py:22-41,47-80 -> slq_plan_generate_probes
 *   - the spectral-function registry, src/primate/special.py:78-107 -> SLQ_FUN_* ids
 *
 * Conventions kept from the reference: alpha/beta have deg+1 entries, beta[0] = 0
 * (lanczos.h:121); rtol is scaled by sqrt(n) (lanczos.h:110); `orth` = number of most recent
 * Lanczos vectors (current one included) to re-orthogonalise against (lanczos.h:135);
 * out-of-range orth (<0 or >deg) means deg (src/primate/operators.py:80); probes are column-major
 * n x P (src/primate/random.py:76); nodes ascending with weights = squared first eigenvector
 * components (integrate.py:63-64).
 */
#ifndef SLQ_H
#define SLQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLQ_VERSION 100

/* ---- status codes --------------------------------------------------------------------------- */
enum {
  SLQ_OK = 0,
  SLQ_EINVAL = -1,   /* invalid argument (maps to Python AssertionError / ValueError)           */
  SLQ_ENOMEM = -2,   /* host or device allocation failed                                         */
  SLQ_EHIP = -3,     /* a HIP runtime call or kernel failed                                      */
  SLQ_ENODEV = -4,   /* no usable gfx950 device                                                  */
  SLQ_ECALLBACK = -5, /* a host-callback operator returned non-zero                              */
  SLQ_ENOTCONV = -6   /* tridiagonal QL did not converge for at least one probe                   */
};

enum { SLQ_F32 = 0, SLQ_F64 = 1 };

/* Spectral functions; ids and parameter meaning follow src/primate/special.py:78-107:
 *   EXP {t}: exp(t x) | SMOOTHSTEP {a,b}: y=clip((x-a)/d,0,1), d=b-a (1 if a==b), 3y^2-2y^3 |
 *   STEP {c, nonnegative}: (|x| or x) < c ? 0 : 1  (numrank = {1e-6, 1}) |
 *   SOFTSIGN {q} | LOG: log(max(x, eps_f64)) | NONE: skip the reduction (nodes/weights only). */
enum {
  SLQ_FUN_IDENTITY = 0,
  SLQ_FUN_ABS = 1,
  SLQ_FUN_SQRT = 2,
  SLQ_FUN_LOG = 3,
  SLQ_FUN_INV = 4,
  SLQ_FUN_EXP = 5,
  SLQ_FUN_SMOOTHSTEP = 6,
  SLQ_FUN_STEP = 7,
  SLQ_FUN_SOFTSIGN = 8,
  SLQ_FUN_NONE = -1
};

/* Probe distributions (src/primate/random.py:12-18). */
enum { SLQ_PDF_RADEMACHER = 0, SLQ_PDF_NORMAL = 1, SLQ_PDF_SPHERE = 2 };

typedef struct slq_context slq_context;   /* one per (process, GPU): device id + HIP stream      */
typedef struct slq_operator slq_operator; /* a symmetric linear operator resident on that GPU    */
typedef struct slq_plan slq_plan;         /* workspace + state of one batched Lanczos run        */
typedef struct slq_diag slq_diag;         /* device-resident accumulators of the diagonal estimator */
typedef struct slq_dmat slq_dmat;         /* column-major n x m fp64 matrix resident on the device  */

/* Host-callback operator: y = A x on HOST memory (the fallback for arbitrary Python
 * LinearOperators; mirrors PyLinearOperator::matvec, src/primate/include/pylinop.h:32-40).
 * x has ncols entries, y nrows entries, both of the operator's dtype. Return 0 on success. */
typedef int (*slq_matvec_fn)(void *user, const void *x, void *y);
/* Device plugin: Y = A X for ncols columns; d_X, d_Y are DEVICE pointers to column-major n x ncols arrays of
 * the operator dtype (leading dimension n). The callee enqueues its work on `stream` (a hipStream_t), or on
 * any stream provided it has completed or been ordered after/before `stream` when it returns. Nonzero = error. */
typedef int (*slq_matmat_device_fn)(void *user, const void *d_X, void *d_Y, int64_t n, int ncols, void *stream);

/* ---- context -------------------------------------------------------------------------------- */
const char *slq_last_error(void);
int slq_version(void);
int slq_device_count(int *count);
/* device < 0: use the current HIP device. stream == NULL: the context creates its own stream. */
int slq_context_create(int device, void *hip_stream, slq_context **out);
int slq_context_destroy(slq_context *ctx);
int slq_context_synchronize(slq_context *ctx);
/* bytes free / total on the context's device */
int slq_context_meminfo(slq_context *ctx, size_t *free_bytes, size_t *total_bytes);
/* the HIP device ordinal the context is bound to (device = -1 at creation resolves to the current device) */
int slq_context_device(slq_context *ctx, int *device);

/* ---- operators ------------------------------------------------------------------------------ */
/* CSR, int32 indices, host arrays: copied to the device once (the reference copies the matrix
 * at least three times PER PROBE at its FFI, src/primate/_lanczos.cpp:88-93 +
 * src/primate/include/eigen_operators.h:64). */
int slq_csr_create(slq_context *ctx, int dtype, int64_t n, int64_t nnz, const int32_t *rowptr,
                   const int32_t *colind, const void *vals, slq_operator **out);
/* Same, arrays already on the device. The operator is built exactly as by slq_csr_create (validated, reordered, upper
 * triangle, tiles): those decisions are taken on the host, so one copy of the arrays is read back (once), and the caller
 * may free its arrays after the call. */
int slq_csr_create_device(slq_context *ctx, int dtype, int64_t n, int64_t nnz,
                          const int32_t *d_rowptr, const int32_t *d_colind, const void *d_vals,
                          slq_operator **out);
/* Gram operator x -> A^T (A x) of a rectangular CSR matrix A (mrows x ncols, host arrays): the operator Lanczos sees is
 * ncols x ncols (src/primate/include/eigen_operators.h:57-72, SparseEigenLinearOperator<F, true>; unbound in the reference's
 * Python module, src/primate/_lanczos.cpp:104-111). */
int slq_csr_gram_create(slq_context *ctx, int dtype, int64_t mrows, int64_t ncols, int64_t nnz, const int32_t *rowptr,
                        const int32_t *colind, const void *vals, slq_operator **out);
/* Affine operator A + t B of two n x n CSR matrices (eigen_operators.h:106-137, SparseEigenAffineOperator); t = 0 until
 * slq_operator_set_parameter (eigen_operators.h:134-136) changes it. Stored on the union pattern as an ordinary CSR operator. */
int slq_csr_affine_create(slq_context *ctx, int dtype, int64_t n, int64_t nnz_a, const int32_t *rowptr_a, const int32_t *colind_a,
                          const void *vals_a, int64_t nnz_b, const int32_t *rowptr_b, const int32_t *colind_b, const void *vals_b,
                          slq_operator **out);
int slq_operator_set_parameter(slq_operator *op, double t);
/* Dense symmetric n x n, column-major with leading dimension lda (host array, copied). */
int slq_dense_create(slq_context *ctx, int dtype, int64_t n, const void *A, int64_t lda,
                     slq_operator **out);
/* Host-callback operator (device <-> host round trip per Lanczos step and probe). */
/* A GPU-resident LinearOperator plugin (e.g. a torch module, a user's HIP kernel): the Lanczos vectors never
 * leave HBM. Same place in the reference: any object satisfying the LinearOperator concept
 * (src/primate/include/linear_operator.h:25-29), here with device buffers. */
int slq_device_callback_create(slq_context *ctx, int dtype, int64_t n, slq_matmat_device_fn fn, void *user,
                               slq_operator **out);
int slq_callback_create(slq_context *ctx, int dtype, int64_t n, slq_matvec_fn fn, void *user,
                        slq_operator **out);
int slq_operator_destroy(slq_operator *op);
int slq_operator_shape(const slq_operator *op, int64_t *nrows, int64_t *ncols, int64_t *nnz,
                       int *dtype);
/* Y = A X for a column-major host panel (n x b, ld); exercises the device SpMM on its own. */
int slq_operator_matmat(slq_operator *op, const void *X, int64_t ldx, void *Y, int64_t ldy,
                        int b);

/* ---- plan: batched lock-step Lanczos over P probes --------------------------------------------- */
/* keep_basis != 0 retains all deg Lanczos vectors (ncv = deg, as MatrixFunction fixes it for the
 * f(A)v action, src/primate/operators.py:75-77); otherwise only max(orth,2)(+1) ring slots are
 * resident (ncv = clip(orth, 2, deg), src/primate/lanczos.py:89). */
int slq_plan_create(slq_context *ctx, slq_operator *op, int nprobes, int deg, int orth,
                    int keep_basis, slq_plan **out);
int slq_plan_destroy(slq_plan *plan);
/* Device bytes this plan holds / would hold. */
int slq_plan_workspace_bytes(const slq_plan *plan, size_t *bytes);
int slq_plan_query_bytes(int dtype, int64_t n, int nprobes, int deg, int orth, int keep_basis,
                         size_t *bytes);
/* What the plan decided (panel geometry and launch sequence): for byte models and records. */
typedef struct {
  int panel_width;   /* PW: probes per panel row                                                        */
  int panels;        /* NP                                                                              */
  int ring_slots;    /* S                                                                               */
  int sequence;      /* steps with r_j <= 8: 0 store-and-revisit sweeps, 1 fused recompute passes,      */
                     /* 2 fused passes with the intermediate stored (operators without gather locality),  */
                     /* 3 sweeps with the fp32 archive ring (SLQ_RING32 opt-in), 4 ring-fed passes with the  */
                     /* projections taken from Gram rows of the update passes (one gather pass fewer panel reads) */
  int pipelined;     /* dots/update passes use the pipelined row loop                                   */
  int reordered;     /* rows stored in the XCD-aware reverse Cuthill-McKee order                        */
  int upper_alpha;   /* alpha pass walks the upper triangle (exactly symmetric CSR)                     */
  double far_per_row;/* stored nonzeros per row further than 4096 rows from the diagonal                */
  int tiles;         /* fused passes run on LDS workgroup tiles: 0 no, 1 behind barriers, 2 ring-fed (SLQ_TILES) */
  int fused_alpha;   /* sequence 4 only: the update pass of step j also takes step j + 1's alpha dot (no alpha-only pass after step 0) */
} slq_plan_info;
int slq_plan_describe(const slq_plan *plan, slq_plan_info *out);

/* Parity mode: host probes, column-major n x nprobes of the plan's dtype (ld >= n). */
int slq_plan_set_probes(slq_plan *plan, const void *X, int64_t ldx);
/* Same, probes already on the device (column-major n x nprobes, contiguous: ldx == n), on the
 * context's stream: nothing crosses PCIe. */
int slq_plan_set_probes_device(slq_plan *plan, const void *d_X, int64_t ldx);
/* Throughput mode: counter-based Philox4x32-10 on the device; the stream of probe `i` depends
 * only on (seed, probe_offset + i), so results do not depend on how probes are sharded. */
int slq_plan_generate_probes(slq_plan *plan, int pdf, uint64_t seed, uint64_t probe_offset);
/* Copy the current probes back as a column-major n x nprobes host panel. */
int slq_plan_get_probes(slq_plan *plan, void *X, int64_t ldx);

/* deg Lanczos steps for all probes (asynchronous on the context stream). */
int slq_plan_run(slq_plan *plan, double rtol);
/* alpha, beta: nprobes x (deg+1) row-major of the plan dtype; steps: nprobes ints. Any may be
 * NULL. Synchronises. */
int slq_plan_get_tridiag(slq_plan *plan, void *alpha, void *beta, int32_t *steps);
/* Gauss quadrature of every probe's Jacobi matrix on the device, then
 * quad[i] = sum_k f(nodes[i,k]) * weights[i,k] * ||v_i||^2 (src/primate/operators.py:149-150).
 * quad: nprobes doubles or NULL; nodes/weights: nprobes x deg row-major doubles or NULL.
 * Synchronises. */
int slq_plan_quadrature(slq_plan *plan, int fun_id, const double *fun_params, double *quad,
                        double *nodes, double *weights);
/* Lanczos basis of probe `probe` as a column-major n x deg host array (normalised columns;
 * columns past an early stop are zero). Requires keep_basis. */
int slq_plan_get_basis(slq_plan *plan, int probe, void *Q, int64_t ldq);
/* Y[:, i] = f(A) x_i ~= ||x_i|| Q_i Y_i (f(theta_i) * Y_i[0,:])  (src/primate/operators.py:113-124);
 * column-major n x nprobes host output. Requires keep_basis and a completed run. */
int slq_plan_fun_action(slq_plan *plan, int fun_id, const double *fun_params, void *Y,
                        int64_t ldy);

/* Stand-alone Gauss quadrature of nb Jacobi matrices on the device (the C-ABI form of
 * integrate.quadrature(d, e, deg, quad="gw"), src/primate/integrate.py:57-64): d, e are
 * nb x deg row-major doubles, e[:,0] is ignored (must be 0 in the reference, integrate.py:59) and
 * e[:,i] couples i-1 and i. quad[i] = sum_k f(nodes[i,k]) weights[i,k]. Any output may be NULL. */
int slq_quadrature_batch(slq_context *ctx, int nb, int deg, const double *d, const double *e,
                         int fun_id, const double *fun_params, double *quad, double *nodes,
                         double *weights);

/* Full eigendecomposition of nb symmetric tridiagonals (eigh_tridiag / eigvalsh_tridiag,
 * src/primate/tridiag.py:25-62; what rayleigh_ritz and MatrixFunction._matvec call): d, e as above
 * (e[:,0] ignored). w: nb x deg ascending eigenvalues. Z: nb x deg x deg row-major, eigenvectors in the
 * COLUMNS of each matrix, or NULL for eigenvalues only. deg <= 512 (eigenvectors on chip up to 141). */
int slq_eigh_tridiag_batch(slq_context *ctx, int nb, int deg, const double *d, const double *e, double *w, double *Z);

/* Tall-skinny dense algebra for the exchangeable estimators (xtrace / hutch++: the host-side
 * np.linalg.qr, Q.T @ W, Z.T @ W, ... of src/primate/trace.py:160-176,199-227,296-302) on the matrix
 * cores (fp64 MFMA). Matrices are column-major n x cols with leading dimension n.
 *   gemm_tn: C (ma x mb, row-major, host) = A[:, a0:a0+ma]^T  B[:, b0:b0+mb]
 *   gemm_nn: OUT[:, o0:o0+mb] = beta * OUT[:, o0:o0+mb] + alpha * A[:, a0:a0+ma] * C  (C: ma x mb row-major host)
 *   slq_plan_fun_action_dmat: f(A) X of a completed keep_basis run straight into OUT's columns;
 *   slq_dmat_ptr + slq_plan_set_probes_device feed a matrix's columns back in as probes. */
int slq_dmat_create(slq_context *ctx, int64_t n, int cols, slq_dmat **out);
int slq_dmat_destroy(slq_dmat *m);
int slq_dmat_set(slq_dmat *m, int c0, int nc, const double *host, int64_t ld);
int slq_dmat_get(slq_dmat *m, int c0, int nc, double *host, int64_t ld);
int slq_dmat_ptr(slq_dmat *m, int c0, void **dptr);
/* Isotropic probes straight into columns [c0, c0+nc): element (seed, probe id = probe_offset + column, row) of
 * the same Philox stream as slq_plan_generate_probes; sphere columns have norm sqrt(n). The batch filler of
 * src/primate/random.py:100-142 (class Isotropic) without a host array. */
int slq_dmat_generate(slq_dmat *m, int c0, int nc, int pdf, uint64_t seed, uint64_t probe_offset);
int slq_dmat_copy(slq_dmat *dst, int d0, slq_dmat *src, int s0, int nc); /* dst[:, d0:d0+nc] = src[:, s0:s0+nc], on the device */
/* dst[dr0:dr0+nrows, d0:d0+nc] = src[sr0:sr0+nrows, s0:s0+nc] between matrices of different heights (device to device): how a
 * row-sharded sketch takes its rows out of full columns and back (xtrace with row-sharded sketches: trace.py:296-302 at scale) */
int slq_dmat_copy_rows(slq_dmat *dst, int d0, int64_t dr0, slq_dmat *src, int s0, int64_t sr0, int64_t nrows, int nc);
int slq_dmat_gemm_tn(slq_dmat *A, int a0, int ma, slq_dmat *B, int b0, int mb, double *C_host);
int slq_dmat_gemm_nn(slq_dmat *OUT, int o0, slq_dmat *A, int a0, int ma, const double *C_host, int mb,
                     double alpha, double beta);
int slq_plan_fun_action_dmat(slq_plan *plan, int fun_id, const double *fun_params, slq_dmat *OUT, int o0);
/* The plan's current probes (set or device-generated, not yet consumed by a run), as the estimators use
 * them (sphere draws scaled to norm sqrt(n), src/primate/random.py:36-41), into OUT's columns
 * [o0, o0 + nprobes): lets xtrace keep its sample matrix W on the device without a host draw. */
int slq_plan_get_probes_dmat(slq_plan *plan, slq_dmat *OUT, int o0);

/* Device bandwidth probe with the access shape of the sweeps (16 B/lane, one contiguous window):
 * mode 0 = two read streams, 1 = in-place triad (2 reads + 1 write), 2 = copy. Reports GB/s. Used by
 * bench.py to quote the roofline fraction against the measured rate as well as the 8 TB/s spec. */
int slq_measure_stream(slq_context *ctx, int mode, size_t bytes_per_stream, int reps, double *gbps);

/* FTTR quadrature weights on the device (integrate.quadrature(..., quad="fttr"),
 * src/primate/integrate.py:65-69 -> src/primate/fttr.py:17-29): theta, weights are nb x k row-major;
 * alpha, beta nb x n row-major (beta[:,0] unused); values as the reference's fttr() returns them. */
int slq_fttr_batch(slq_context *ctx, int nb, int n, int k, const double *theta, const double *alpha,
                   const double *beta, double *weights);

/* Diagonal estimator (the loop body of diag(), src/primate/diagonal.py:74-79): for every probe of a
 * completed keep_basis run, in order, numer += f(A)v * v, denom += v*v, and the running mean of
 * numer/denom (the reference's estimate). Everything stays on the device between updates. */
int slq_diag_create(slq_context *ctx, int64_t n, slq_diag **out);
int slq_diag_destroy(slq_diag *d);
int slq_diag_update(slq_diag *d, slq_plan *plan, int fun_id, const double *fun_params);
/* any of numer / denom / running_mean (n doubles each) and count may be NULL */
int slq_diag_get(slq_diag *d, double *numer, double *denom, double *running_mean, int64_t *count);

/* Per-kernel device time accumulated by HIP events on the context stream (for bench.py's
 * roofline line). enable != 0 turns event recording on for subsequent slq_plan_run calls. */
enum {
  SLQ_K_SPMM = 0,    /* sweep A: panel SpMM + three-term update + alpha partials                  */
  SLQ_K_AXPY_NORM,   /* sweep B (orth = 0): w -= alpha q_c, ||w||^2 partials                       */
  SLQ_K_REORTH_DOT,  /* sweep B (orth > 0): w -= alpha q_c, c = Q_r^T w partials                   */
  SLQ_K_REORTH_UPD,  /* sweep C: w -= Q_r c, ||w||^2 partials                                      */
  SLQ_K_FINALIZE,    /* all per-step scalar kernels (partials -> alpha/beta/coefficients)           */
  SLQ_K_PROBES,      /* probe generation / layout                                                   */
  SLQ_K_QUADRATURE,  /* tridiagonal eigensolve + f reduction                                        */
  SLQ_K_COMBINE,     /* f(A)x = sum_t g_t W_t over the kept basis (slq_plan_fun_action; not a recurrence sweep) */
  SLQ_K_COUNT
};
typedef struct {
  double ms[SLQ_K_COUNT];       /* summed device time per class                                   */
  int64_t launches[SLQ_K_COUNT];
} slq_profile;
int slq_plan_profile_enable(slq_plan *plan, int enable);
/* Byte accounting of the deep-window update sweep (orth > 8), which reads a ring column only when some probe of the panel has a non-zero projection on it - the
 * reference skips a projection per probe below its threshold, src/primate/include/lanczos.h:62 -: columns read / columns offered, summed over launches and panels
 * since the last reset. Synchronises. */
int slq_plan_sweep_columns(slq_plan *plan, uint64_t *read, uint64_t *offered, int reset);
int slq_plan_profile_read(slq_plan *plan, slq_profile *out, int reset);

/* Failure reporting of the ring-fed tile pass (k_csr_ring_pass). Every wait inside that kernel is bounded; a workgroup
 * whose wait runs out raises a device word and leaves, and every accessor that hands results of a run to the host
 * (slq_plan_get_tridiag, _quadrature, _get_basis, _fun_action[_dmat], slq_diag_update, the one-shot entries) then returns
 * SLQ_EHIP instead of undefined numbers (the reference's kernel has no failure mode of its own:
 * src/primate/include/lanczos.h:92-149 is noexcept host code). Two hooks for the tests:
 *   slq_debug_ring_flag_status   the flag -> status translation those accessors share (no device work: CPU-testable)
 *   slq_debug_plan_poke_ring_flag  sets a plan's device word as an aborting workgroup would */
int slq_debug_ring_flag_status(int flag);
int slq_debug_plan_poke_ring_flag(slq_plan *plan, int value);

/* ---- one-shot entries ---------------------------------------------------------------------------- */
/* P probes in one call: the batched counterpart of the Python loop at
 * src/primate/operators.py:145-150. X: host column-major n x nprobes, or NULL to draw probes on
 * the device (pdf, seed, probe_offset). quad_out: nprobes doubles. nodes_out / weights_out:
 * nprobes x deg row-major doubles or NULL. */
int slq_quad_batch(slq_context *ctx, slq_operator *op, const void *X, int64_t ldx, int pdf,
                   uint64_t seed, uint64_t probe_offset, int nprobes, int deg, double rtol,
                   int orth, int fun_id, const double *fun_params, double *quad_out,
                   double *nodes_out, double *weights_out);

/* Y[:, i] = f(A) X[:, i] for nvec columns in one call: the batched counterpart of
 * MatrixFunction._matvec (src/primate/operators.py:102-124: Lanczos with the full basis kept,
 * eigh_tridiagonal, Q (Y (f(theta) * Y[0,:])) ||x||). X, Y: host column-major, operator dtype. Built-in
 * fun ids only. Columns are processed in as few lock-step batches as the free device memory allows. */
int slq_fAv_batch(slq_context *ctx, slq_operator *op, const void *X, int64_t ldx, int nvec, int deg,
                  double rtol, int orth, int fun_id, const double *fun_params, void *Y, int64_t ldy);

/* Single-vector drop-in for primate._lanczos.lanczos (src/primate/_lanczos.cpp:88-99), host
 * pointers, same in/out contract: v (n) is scratch and is clobbered; alpha, beta (deg+1) and
 * Q (n x ncv column-major) are written in place; beta[0] = 0. Q's incoming contents take part in
 * the re-orthogonalisation exactly as in the reference (columns other than ncv-1 and 0 are not
 * cleared, src/primate/include/lanczos.h:118-121). Returns the number of executed steps (>= 1)
 * or a negative error code. */
int slq_lanczos_f64(slq_context *ctx, slq_operator *op, double *v, int deg, double rtol, int orth,
                    double *alpha, double *beta, double *Q, size_t ncv);
int slq_lanczos_f32(slq_context *ctx, slq_operator *op, float *v, int deg, float rtol, int orth,
                    float *alpha, float *beta, float *Q, size_t ncv);

#ifdef __cplusplus
}
#endif
#endif /* SLQ_H */
