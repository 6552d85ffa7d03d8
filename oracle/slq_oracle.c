/* TEST INFRASTRUCTURE — CPU oracle for the SLQ hot path. NOT product code. See slq_oracle.h. */
#define SLQ_ORACLE_IMPL
#include "slq_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Python-style modulus; restates src/primate/include/lanczos.h:34-36. */
static inline int oracle_pymod(int a, int b) { return (b + (a % b)) % b; }

/* src/primate/special.py:78-107 and the helpers it dispatches to. */
double oracle_apply_fun(int fun_id, const double *params, double x) {
  switch (fun_id) {
    case ORACLE_FUN_IDENTITY: return x;
    case ORACLE_FUN_ABS: return fabs(x);
    case ORACLE_FUN_SQRT: return sqrt(x);
    case ORACLE_FUN_LOG: return log(x > DBL_EPSILON ? x : DBL_EPSILON); /* special.py:89-90 */
    case ORACLE_FUN_INV: return 1.0 / x;
    case ORACLE_FUN_EXP: return exp(params[0] * x); /* special.py:62-66 */
    case ORACLE_FUN_SMOOTHSTEP: {                    /* special.py:33-55 */
      const double a = params[0], b = params[1];
      const double d = (a != b) ? (b - a) : 1.0;
      double y = (x - a) / d;
      y = y < 0.0 ? 0.0 : (y > 1.0 ? 1.0 : y);
      return 3.0 * y * y - 2.0 * y * y * y;
    }
    case ORACLE_FUN_STEP: { /* special.py:69-74; numrank = step(c=1e-6, nonnegative) :103-105 */
      const double xx = (params[1] != 0.0) ? fabs(x) : x;
      return xx < params[0] ? 0.0 : 1.0;
    }
    case ORACLE_FUN_SOFTSIGN: { /* special.py:10-30 */
      const int q = (int)params[0];
      const double xc = x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x);
      double J = 1.0, pw = 1.0, s = 0.0;
      for (int i = 0; i <= q; ++i) {
        if (i > 0) {
          J *= (2.0 * i - 1.0) / (2.0 * i);
          pw *= (1.0 - xc * xc);
        }
        s += xc * pw * J;
      }
      return s;
    }
    default: return NAN;
  }
}

#define F double
#define FN(x) x##_f64
#define F_EPS DBL_EPSILON
#define F_SQRT sqrt
#define F_FABS fabs
#define F_HYPOT hypot
#include "slq_oracle_impl.h"
#include "slq_oracle_batch.h"
#undef F
#undef FN
#undef F_EPS
#undef F_SQRT
#undef F_FABS
#undef F_HYPOT
#undef F_BLOCKED_DOT

#define F float
#define FN(x) x##_f32
#define F_EPS FLT_EPSILON
#define F_BLOCKED_DOT 1 /* fp32 reductions blocked the way Eigen's packet reductions are (slq_oracle_impl.h: dot) */
#define F_SQRT sqrtf
#define F_FABS fabsf
#define F_HYPOT hypotf
#include "slq_oracle_impl.h"
#include "slq_oracle_batch.h"
#undef F
#undef FN
#undef F_EPS
#undef F_SQRT
#undef F_FABS
#undef F_HYPOT
#undef F_BLOCKED_DOT
