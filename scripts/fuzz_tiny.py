"""Tiny operators (n = 4..40), orth = 1..4: the regime where the reference's second-pass projection on the CURRENT
Lanczos vector can exceed its 2 eps sqrt(n) threshold, which the merged alpha+dots pass replaces by gamma_0 = 0.
Compares alpha/beta of the device run with the oracle, probe by probe, on well-posed runs."""
import sys
from pathlib import Path
import numpy as np, scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import oracle
from primate_amd import engine as eng
oracle.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst, cases, skipped = 0.0, 0, 0
for it in range(1500):
    n = int(rng.integers(4, 41))
    B = rng.standard_normal((n, n)); A = B @ B.T / n + np.diag(rng.uniform(0.1, 2.0, n)); A = (A + A.T) / 2
    A[np.abs(A) < 0.3] = 0.0; A = (A + A.T) / 2 + np.eye(n) * 2
    M = sp.csr_matrix(A); M.sort_indices()
    deg = int(rng.integers(2, min(n, 12) + 1)); orth = int(rng.integers(1, 5))
    X = np.asfortranarray(rng.standard_normal((n, 5)))
    op = eng.DeviceOperator(M); plan = eng.LanczosPlan(op, 5, deg, orth); plan.set_probes(X); plan.run()
    a, b, st = plan.tridiag()
    for c in range(5):
        ar, br, Q = np.zeros(deg + 1), np.zeros(deg + 1), np.zeros((n, max(orth, 2)), order="F")
        s = oracle.lanczos(M, X[:, c].copy(), deg, 1e-8, orth, ar, br, Q)
        if s < deg or st[c] < deg or (deg > 1 and np.min(np.abs(br[1:deg])) < 1e-3 * np.max(np.abs(br[1:deg]))):
            skipped += 1; continue
        err = max(np.max(np.abs(a[c][:deg] - ar[:deg])), np.max(np.abs(b[c][1:deg] - br[1:deg]))) / np.max(np.abs(ar[:deg]))
        worst = max(worst, err); cases += 1
        if err > 1e-10: print("large", n, deg, orth, err, flush=True)
    plan.close(); op.close()
print(f"{cases} probe runs compared, {skipped} ill-posed skipped, worst |alpha/beta difference| / max|alpha| = {worst:.2e}")
