"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON in the dev container.

Run (dev container only; /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

How the reference is made importable: `import primate` fails with an ordinary
PackageNotFoundError because src/primate/__init__.py:3 reads dist metadata; registering an empty
module object named `primate` whose __path__ is /root/reference/src/primate lets every submodule
import as plain Python. Four of them (lanczos, operators, trace, diagonal) do
`from . import _lanczos` — the compiled extension, which is unbuildable here (Eigen is an empty
submodule). Two kinds of vectors are therefore captured, and each array's provenance is recorded
in the fixture's `provenance` entry:

  "pure"      produced by reference code alone: the NumPy twins `_lanczos_recurrence` /
              `_orth_vector` (src/primate/lanczos.py:196-238), `quadrature`, `eigh_tridiag`,
              `fttr`, `isotropic`, `param_callable`, `Covariance`/`MeanEstimator`, and `hutch` on a
              plain ndarray (no Lanczos involved). These PIN the oracle.
  "injected"  produced by the reference's drivers (`lanczos()`, `MatrixFunction.quad/_matvec`,
              `hutch(MatrixFunction)`) running on top of oracle/oracle.py:lanczos standing in for
              `primate._lanczos`. These pin the host-side glue (clamps, buffer reuse, estimator
              order), given an oracle already pinned by the "pure" vectors.

Only data is written: inputs (or the seeds/recipes that regenerate them) and expected outputs.
"""

import os
import sys
import types
from pathlib import Path

import numpy as np
import scipy.sparse as sp

REF = Path("/root/reference/src/primate")
HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.dont_write_bytecode = True
sys.path.insert(0, str(ROOT))


def import_reference():
	from oracle import oracle

	pkg = types.ModuleType("primate")
	pkg.__path__ = [str(REF)]
	sys.modules["primate"] = pkg
	shim = types.ModuleType("primate._lanczos")
	shim.lanczos = oracle.lanczos
	sys.modules["primate._lanczos"] = shim
	pkg._lanczos = shim
	import primate.diagonal  # noqa: F401
	import primate.estimators  # noqa: F401
	import primate.fttr  # noqa: F401
	import primate.integrate  # noqa: F401
	import primate.lanczos  # noqa: F401
	import primate.operators  # noqa: F401
	import primate.random  # noqa: F401
	import primate.special  # noqa: F401
	import primate.stats  # noqa: F401
	import primate.trace  # noqa: F401
	import primate.tridiag  # noqa: F401

	return sys.modules["primate"]


def laplacian_2d(m: int, dtype=np.float64):
	"""2D 5-point Dirichlet Laplacian on an m x m grid (BASELINE.json configs[1] recipe at size m)."""
	T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))
	A = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsr().astype(dtype)
	A.sort_indices()
	return A


def main():
	P = import_reference()
	from primate.estimators import MeanEstimator
	from primate.fttr import fttr
	from primate.integrate import quadrature
	from primate.lanczos import _lanczos_recurrence, lanczos
	from primate.operators import MatrixFunction
	from primate.random import isotropic, symmetric
	from primate.special import param_callable
	from primate.stats import Covariance
	from primate.trace import hutch
	from primate.tridiag import eigh_tridiag

	## ---- 1. Lanczos known-answer vectors (pure: NumPy twin lanczos.py:211-238) -----------------
	out = {}
	prov = {}
	rng = np.random.default_rng(seed=1234)  # tests/test_lanczos.py:11-20 inputs
	d = 50
	A = rng.uniform(size=(d, d))
	A @= A.T
	v0 = rng.uniform(size=d)
	out["kat_A"], out["kat_v0"] = A, v0
	for orth, ncv in [(50, 50), (0, 2), (3, 3), (3, 50), (10, 20)]:
		a, b, Q = _lanczos_recurrence(A, v0.copy(), d, 1e-8, orth, None, ncv)
		out[f"kat_alpha_o{orth}_c{ncv}"], out[f"kat_beta_o{orth}_c{ncv}"] = a, b
		prov[f"kat_*_o{orth}_c{ncv}"] = "pure"
		if (orth, ncv) == (50, 50):
			out["kat_Q_o50_c50"] = Q
	out["kat_eigvalsh"] = np.linalg.eigvalsh(A)

	## early stop: A with a 5-dimensional invariant subspace reachable from v (lanczos.h:140-142)
	rng = np.random.default_rng(77)
	U, _ = np.linalg.qr(rng.standard_normal((40, 40)))
	ew = np.repeat([1.0, 2.0, 3.5, 5.0, 8.0], 8)
	Ainv = (U * ew) @ U.T
	Ainv = (Ainv + Ainv.T) / 2
	vinv = rng.standard_normal(40)
	a, b, Q = _lanczos_recurrence(Ainv, vinv.copy(), 20, 1e-8, 20, None, 20)
	out["stop_A"], out["stop_v"], out["stop_alpha"], out["stop_beta"] = Ainv, vinv, a, b
	prov["stop_*"] = "pure"

	## ---- 2. sparse Laplacian, pure twin + pure quadrature, per-probe SLQ values ---------------
	m = 24
	L = laplacian_2d(m)
	n = L.shape[0]
	pdf = isotropic(pdf="rademacher", seed=1234)
	V = pdf(size=(n, 12))  # F-ordered float64, src/primate/random.py:74-78
	out["lap_m"], out["lap_probes"] = np.int64(m), V
	prov["lap_probes"] = "pure"
	deg = 20
	funs = {
		"identity": {}, "log": {}, "exp": {}, "sqrt": {}, "inv": {}, "abs": {},
		"smoothstep": {"a": 0.5, "b": 6.0}, "numrank": {}, "exp_t": {"t": -0.1},
	}  # fmt: skip
	for orth in [0, 3, deg]:
		al = np.zeros((V.shape[1], deg))
		be = np.zeros((V.shape[1], deg))
		nodes = np.zeros((V.shape[1], deg))
		weights = np.zeros((V.shape[1], deg))
		for j in range(V.shape[1]):
			a, b, _ = _lanczos_recurrence(L, V[:, j].copy(), deg, 1e-8, orth, None, deg)
			al[j], be[j] = a, b  # b includes beta[0] = 0 (len deg)
			nodes[j], weights[j] = quadrature(a, b, deg=deg, quad="gw")
		out[f"lap_alpha_o{orth}"], out[f"lap_beta_o{orth}"] = al, be
		out[f"lap_nodes_o{orth}"], out[f"lap_weights_o{orth}"] = nodes, weights
		for name, kw in funs.items():
			f = param_callable("exp" if name == "exp_t" else name, **dict(kw))
			nrm2 = np.linalg.norm(V, axis=0) ** 2
			out[f"lap_quad_{name}_o{orth}"] = np.array([np.sum(f(nodes[j]) * weights[j]) for j in range(V.shape[1])]) * nrm2
		prov[f"lap_*_o{orth}"] = "pure"
	out["lap_fun_params"] = np.array([0.5, 6.0, -0.1])  # smoothstep a, b; exp_t t

	## ---- 3. the same through the reference's MatrixFunction / hutch drivers (injected) --------
	for orth in [0, 3, deg]:
		M = MatrixFunction(L, fun="log", deg=deg, orth=orth)
		out[f"mf_quad_log_o{orth}"] = M.quad(V.copy())  # stale-Q semantics (operators.py:138-148)
		prov[f"mf_quad_log_o{orth}"] = "injected"
	for fun in ["log", "exp"]:
		M = MatrixFunction(L, fun=fun, deg=deg, orth=3)
		out[f"hutch_mf_{fun}_c40"] = np.float64(hutch(M, converge="count", count=40, seed=1234))
		est, info = hutch(M, converge="count", count=40, seed=1234, full=True, batch=8, record=True)
		out[f"hutch_mf_{fun}_c40_full"] = np.float64(est)
		out[f"hutch_mf_{fun}_c40_samples"] = np.ravel(info.estimator.values)
		prov[f"hutch_mf_{fun}_*"] = "injected"
	M = MatrixFunction(L, fun="exp", deg=deg, orth=deg, t=-0.1)
	out["mf_matvec_exp_t"] = np.column_stack([M._matvec(V[:, j].copy()).ravel() for j in range(4)])
	prov["mf_matvec_exp_t"] = "injected"
	## lanczos() public API clamps (lanczos.py:78-90): orth -> ncv
	for orth, rb in [(0, False), (5, False), (-1, False), (3, True)]:
		res = lanczos(L, v0=V[:, 0].copy(), deg=deg, orth=orth, return_basis=rb)
		(a, b) = res[0] if rb else res
		out[f"api_alpha_o{orth}_rb{int(rb)}"], out[f"api_beta_o{orth}_rb{int(rb)}"] = a, b
		if rb:
			out[f"api_Q_o{orth}_rb1"] = res[1]
		prov[f"api_*_o{orth}_rb{int(rb)}"] = "injected"

	## ---- 4. plumbing config (BASELINE.json configs[0]) at reduced size: pure ------------------
	rng = np.random.default_rng(1234)
	B = rng.standard_normal((200, 200))
	Ad = B @ B.T / 200 + np.eye(200)
	out["dense_A"] = Ad
	out["dense_hutch_c64"] = np.float64(hutch(Ad, converge="count", count=64, seed=1234))
	prov["dense_hutch_c64"] = "pure"
	## tests/test_trace.py:48-57 identity: hutch(A) == hutch(MatrixFunction(A, deg=n, orth=n))
	rng = np.random.default_rng(1234)
	nn = 50
	ew = rng.uniform(size=nn, low=1 / nn, high=1.0)
	As = symmetric(nn, pd=True, ew=ew, seed=rng)
	out["sym_A"] = As
	out["sym_hutch_c150"] = np.float64(hutch(As, converge="count", count=150, seed=1234))
	prov["sym_hutch_c150"] = "pure"
	out["sym_hutch_mf_c150"] = np.float64(hutch(MatrixFunction(As, deg=nn, orth=nn), converge="count", count=150, seed=1234))
	prov["sym_hutch_mf_c150"] = "injected"

	## ---- 5. probes: stream-order contract (tests/test_random.py:6-39), pure --------------------
	for pdfname in ["rademacher", "normal", "sphere"]:
		out[f"iso_{pdfname}_37x5_s1234"] = isotropic(size=(37, 5), pdf=pdfname, seed=1234)
		g = isotropic(pdf=pdfname, seed=99)
		out[f"iso_{pdfname}_seq_s99"] = np.column_stack([g(size=(11, 1)), g(size=(11, 2)), g(size=(11, 1))])
		prov[f"iso_{pdfname}_*"] = "pure"

	## ---- 6. quadrature / tridiagonal known answers, pure ---------------------------------------
	rng = np.random.default_rng(4756)
	dd = rng.uniform(size=30, low=0.5, high=4.0)
	ee = np.append([0.0], rng.uniform(size=29, low=0.1, high=2.0))
	out["tri_d"], out["tri_e"] = dd, ee
	th, tau = quadrature(dd, ee, deg=30, quad="gw")
	out["tri_nodes"], out["tri_weights"] = th, tau
	rw, Y = eigh_tridiag(dd, ee)
	out["tri_Y_abs"] = np.abs(Y)
	w_fttr = np.zeros(30)
	fttr(th, dd, ee, 30, w_fttr)
	out["tri_fttr_raw"] = w_fttr  # as the reference's fttr() returns them
	prov["tri_*"] = "pure"

	## ---- 7. spectral functions (special.py:78-107), pure ---------------------------------------
	xs = np.concatenate([np.linspace(-2, 8, 41), [0.0, 1e-20, 1e-7, 2.5e-6, np.finfo(np.float64).eps]])
	out["fun_x"] = xs
	with np.errstate(all="ignore"):
		for name, kw in funs.items():
			f = param_callable("exp" if name == "exp_t" else name, **dict(kw))
			out[f"fun_{name}"] = np.asarray(f(xs), dtype=np.float64)
		## "softsign" is dispatched by param_callable (special.py:99-101) but is missing from
		## _BUILTIN_MATRIX_FUNCTIONS (:7), so the string form asserts; call the factory directly.
		from primate.special import softsign

		out["fun_softsign"] = np.asarray(softsign(q=10)(xs), dtype=np.float64)
	prov["fun_*"] = "pure"

	## ---- 8. streaming estimator merge semantics (stats.py:66-86), pure -------------------------
	rng = np.random.default_rng(5)
	xs = rng.standard_normal(57) * 3 + 10
	cov = Covariance(dim=1)
	est = MeanEstimator(covariance=True)
	trail = []
	for lo, hi in [(0, 1), (1, 9), (9, 41), (41, 57)]:
		cov.update(xs[lo:hi])
		est.update(xs[lo:hi])
		trail.append([cov.n, cov.mu.item(), cov.S.item(), est.estimate, np.ravel(est.delta)[0]])
	out["est_samples"], out["est_trail"] = xs, np.array(trail)
	prov["est_*"] = "pure"

	## ---- 9. BASELINE configs[1] single-probe anchor (pure twin, ~12 s) ------------
	if os.environ.get("GOLDEN_SKIP_C2", "0") != "1":
		L2 = laplacian_2d(1000)
		v = isotropic(pdf="rademacher", seed=1234)(size=(L2.shape[0], 1))
		f = param_callable("log")
		vals = []
		for orth in [0, 3]:
			a, b, _ = _lanczos_recurrence(L2, v[:, 0].copy(), 30, 1e-8, orth, None, 30)
			nd, wt = quadrature(a, b, deg=30, quad="gw")
			vals.append(np.sum(f(nd) * wt) * np.linalg.norm(v) ** 2)
		out["c2_quad_log_seed1234_o0_o3"] = np.array(vals)
		out["c2_nnz_n"] = np.array([L2.nnz, L2.shape[0]])
		out["c2_rowdeg_hist"] = np.bincount(np.diff(L2.indptr))
		prov["c2_*"] = "pure"

	out["provenance"] = np.array([f"{k}={v}" for k, v in sorted(prov.items())])
	np.savez_compressed(HERE / "slq_golden.npz", **out)
	sz = (HERE / "slq_golden.npz").stat().st_size
	print(f"wrote {HERE / 'slq_golden.npz'} ({sz/1024:.1f} KiB, {len(out)} arrays)")
	for k in ["c2_quad_log_seed1234_o0_o3", "hutch_mf_log_c40", "dense_hutch_c64", "sym_hutch_c150", "sym_hutch_mf_c150"]:
		if k in out:
			print(k, out[k])
	## stale-Q effect, for DESIGN.md
	for orth in [3, deg]:
		d_ = np.max(np.abs(out[f"mf_quad_log_o{orth}"] / out[f"lap_quad_log_o{orth}"] - 1))
		print(f"stale-Q vs fresh relative difference (n={n}, orth={orth}): {d_:.3e}")


if __name__ == "__main__":
	main()
