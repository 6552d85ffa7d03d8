"""The Gram sequence through LOST orthogonality: recurrences of 100-300 steps on small grids, where Ritz
values converge long before the run ends and q_{j+1}.q_{j-s} is no longer O(eps) for the window columns.
Per-probe quadrature values of the device (Gram sequence, and the merged sequence SLQ_GRAM=0 on the same tiles) against the
CPU oracle on identical probes, next to the oracle's OWN sensitivity to a 1-ulp perturbation of the probes (partial
reorthogonalisation over hundreds of steps amplifies rounding in any implementation: that number is the yardstick).
usage: python scripts/gram_long.py [quick]
Reference semantics: src/primate/include/lanczos.h:43-66,133-136; tests/test_lanczos.py:11-20."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d, laplacian_3d
from oracle import oracle
from primate_amd import engine as eng

oracle.build()
os.environ["SLQ_TILES"] = "2"
QUICK = len(sys.argv) > 1 and sys.argv[1] == "quick"
FUNS = [("log", {}), ("exp", {"t": -0.1}), ("inv", {}), ("numrank", {}), ("step", {"c": 1.0})]
NCOL = 4


def oracle_values(A, Xc, deg, orth):
	_, nodes, weights, steps = oracle.quad_batch(A, Xc, deg, orth, fun="identity", fresh_q=True, prefer="csr", return_rule=True, nthreads=8)
	vn2 = np.sum(Xc * Xc, axis=0)
	out = {}
	for f, kw in FUNS:
		out[f] = np.array([np.sum(oracle.apply_fun(f, nodes[i], **kw) * weights[i]) * vn2[i] for i in range(Xc.shape[1])])
	return out, steps


worst = {}
rng = np.random.default_rng(2024)
ops = [("lap2d_100", laplacian_2d(100)), ("lap3d_22", laplacian_3d(22))]
for name, A in ops:
	n = A.shape[0]
	op = eng.DeviceOperator(A)
	for deg in ((100,) if QUICK else (100, 300)):
		for orth in (1, 3, 8):
			for P in ((64,) if QUICK else (20, 64, 130)):
				X = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1)
				cols = sorted(set([0, 1, P // 2, P - 1]))[:NCOL]
				Xc = np.asfortranarray(X[:, cols])
				t0 = time.time()
				ref, steps = oracle_values(A, Xc, deg, orth)
				Xp = np.asfortranarray(Xc * (1 + np.finfo(np.float64).eps * np.sign(rng.standard_normal(Xc.shape))))
				refp, _ = oracle_values(A, Xp, deg, orth)
				t_or = time.time() - t0
				row = {}
				for gram in ("1", "0"):
					os.environ["SLQ_GRAM"] = gram
					plan = eng.LanczosPlan(op, P, deg, orth)
					info = plan.describe()
					plan.set_probes(X)
					plan.run()
					a, b, st = plan.tridiag()
					for f, kw in FUNS:
						got = plan.quadrature(f, **kw)[cols]
						row[(gram, f)] = np.max(np.abs(got - ref[f]) / np.maximum(np.abs(ref[f]), 1e-300))
					row[(gram, "seq")] = info["sequence"]
					row[(gram, "tiles")] = info["tiles"]
					row[(gram, "steps")] = int(st.min())
					plan.close()
				del os.environ["SLQ_GRAM"]
				line = f"{name} k={deg} orth={orth} P={P} [{row[('1','seq')]}/{row[('0','seq')]} tiles={row[('1','tiles')]} steps dev {row[('1','steps')]} oracle {int(steps.min())}] oracle {t_or:.1f}s"
				for f, _ in FUNS:
					sens = np.max(np.abs(refp[f] - ref[f]) / np.maximum(np.abs(ref[f]), 1e-300))
					line += f" | {f}: gram {row[('1', f)]:.1e} merged {row[('0', f)]:.1e} sens {sens:.1e}"
					worst[f] = max(worst.get(f, 0.0), row[("1", f)])
				print(line, flush=True)
	op.close()
## (b) an ill-conditioned operator with a clustered spectrum (advisor finding r03): D^1/2 L D^1/2 with D spanning 1e-4 .. 1e4 on the
## same 5-point pattern (cond ~ 1e8 x cond(L)), fp64 and fp32, every window 1 .. 8: alpha / beta and the quadrature, Gram vs merged
import scipy.sparse as sp
L2 = laplacian_2d(100)
n = L2.shape[0]
dsc = 10.0 ** rng.uniform(-2.0, 2.0, n)
Ab = (sp.diags(dsc) @ L2 @ sp.diags(dsc)).tocsr()
Ab.sort_indices()
for dt in (np.float64, np.float32):
	A = Ab.astype(dt)
	op = eng.DeviceOperator(A)
	P, deg = 64, 60
	X = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1).astype(dt)
	cols = [0, 1, P // 2, P - 1]
	Xc = np.asfortranarray(X[:, cols])
	for orth in range(1, 9):
		ref64 = oracle.quad_batch(Ab, Xc.astype(np.float64), deg, orth, fun="log", fresh_q=True, prefer="csr")
		ref = oracle.quad_batch(A, Xc, deg, orth, fun="log", fresh_q=True, prefer="csr")
		noise = np.max(np.abs(ref - ref64) / np.abs(ref64))
		al, be, Qr = np.zeros(deg + 1, dt), np.zeros(deg + 1, dt), np.zeros((n, max(orth, 2)), dt, order="F")
		oracle.lanczos(A, Xc[:, 0].copy(), deg, 1e-8, orth, al, be, Qr)
		line = f"illcond {np.dtype(dt).name} k={deg} orth={orth} P={P} oracle-vs-fp64 {noise:.1e}"
		for gram in ("1", "0"):
			os.environ["SLQ_GRAM"] = gram
			plan = eng.LanczosPlan(op, P, deg, orth)
			seq = plan.describe()["sequence"]
			plan.set_probes(X)
			plan.run()
			a, b, st = plan.tridiag()
			got = plan.quadrature("log")[cols]
			plan.close()
			ea = np.max(np.abs(a[0][:deg] - al[:deg])) / np.max(np.abs(al[:deg]))
			eb = np.max(np.abs(b[0][1:deg] - be[1:deg])) / np.max(np.abs(be[1:deg]))
			line += f" | {seq}: quad vs fp64 oracle {np.max(np.abs(got - ref64) / np.abs(ref64)):.1e} alpha {ea:.1e} beta {eb:.1e}"
		del os.environ["SLQ_GRAM"]
		print(line, flush=True)
	op.close()
print("worst gram-sequence error per f:", {k: f"{v:.2e}" for k, v in worst.items()})
