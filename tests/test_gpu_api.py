"""The host-side mirror of the reference API (MatrixFunction / hutch / lanczos / quadrature) on
the GPU: these read like the reference's own tests (tests/test_operator.py, test_trace.py,
test_lanczos.py, test_quadrature.py in the reference tree) and also check golden driver outputs.
"""

from pathlib import Path

import numpy as np
import pytest

from conftest import laplacian_2d

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def spd(n, seed=1234, lo=None):
	rng = np.random.default_rng(seed)
	U, _ = np.linalg.qr(rng.standard_normal((n, n)))
	ew = rng.uniform(size=n, low=(1 / n if lo is None else lo), high=1.0)
	A = (U * ew) @ U.T
	return (A + A.T) / 2, ew


def test_quad_form_matches_exact_quadratic_form():
	## reference tests/test_operator.py:35-46
	from primate_amd.operators import MatrixFunction

	rng = np.random.default_rng(1234)
	n = 100
	A, _ = spd(n)
	M = MatrixFunction(A, deg=n, orth=n, dtype=np.float64)
	v = rng.uniform(size=n)
	assert len(M.quad(v)) == 1
	V = rng.uniform(size=(n, 10))
	y1 = M.quad(V)
	assert len(y1) == V.shape[1]
	assert np.allclose(y1, np.diag(V.T @ A @ V))


def test_operator_interface_and_builtin_functions():
	## reference tests/test_operator.py:49-83
	from scipy.sparse.linalg import LinearOperator, aslinearoperator

	from primate_amd.operators import MatrixFunction, is_linear_op, matrix_function
	from primate_amd.special import _BUILTIN_MATRIX_FUNCTIONS, param_callable

	rng = np.random.default_rng(1234)
	n = 100
	A, _ = spd(n, lo=0.05)
	v = rng.uniform(size=n, low=-1, high=1)
	M = MatrixFunction(A, deg=n, orth=n, dtype=np.float64)
	assert isinstance(M, LinearOperator) and is_linear_op(M) and M.degree == n
	assert np.allclose(A @ v, M @ v)
	assert np.allclose(A @ v, MatrixFunction(aslinearoperator(A), deg=n, orth=n) @ v)  # host-callback plugin
	ew, ev = np.linalg.eigh(A)
	for fun in _BUILTIN_MATRIX_FUNCTIONS:
		f = param_callable(fun)
		for F in (fun, f, lambda x, f=f: f(x)):  # name, tagged callable (device), opaque callable (host)
			M = MatrixFunction(A, fun=F, deg=n)
			assert np.allclose(ev @ np.diag(f(ew)) @ ev.T @ v, M @ v), fun
	assert np.allclose(matrix_function(A, fun="exp", v=v, deg=n).ravel(), ev @ np.diag(np.exp(ew)) @ ev.T @ v)


def test_hutch_matrix_function_identity():
	## reference tests/test_trace.py:48-57: same seed, same count -> same estimate to 1e-6
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import hutch

	n = 50
	A, _ = spd(n)
	M = MatrixFunction(A, deg=n, orth=n)
	est1 = hutch(A, converge="count", count=150, seed=1234)
	est2 = hutch(M, converge="count", count=150, seed=1234)
	assert np.isclose(est1, est2, atol=1e-6)


def test_hutch_against_reference_driver_outputs(golden):
	"""Golden: the reference's own hutch() on the same inputs ("pure" for plain arrays; "injected"
	for MatrixFunction, with orth = 0 so the reference's stale-ring quirk does not enter)."""
	from primate_amd.estimators import EstimatorResult
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import hutch

	assert hutch(golden["dense_A"], converge="count", count=64, seed=1234) == pytest.approx(float(golden["dense_hutch_c64"]), rel=1e-12)
	assert hutch(golden["sym_A"], converge="count", count=150, seed=1234) == pytest.approx(float(golden["sym_hutch_c150"]), rel=1e-12)
	M = MatrixFunction(golden["sym_A"], deg=50, orth=50)
	assert hutch(M, converge="count", count=150, seed=1234) == pytest.approx(float(golden["sym_hutch_mf_c150"]), rel=1e-9)
	## default stopping rule, full/callback surface (reference tests/test_trace.py:8-33)
	A = golden["sym_A"]
	est = hutch(A, seed=1234)
	assert abs(A.trace() - est) <= 10 / np.sqrt(50)
	est, info = hutch(A, seed=1234, full=True)
	assert isinstance(info, EstimatorResult) and info.nit >= 3
	calls = []
	hutch(A, callback=lambda r: calls.append(r.nit), seed=5)
	assert len(calls) > 0


def test_hutch_slq_on_sparse_laplacian_matches_oracle_stream(oracle, golden):
	"""hutch(MatrixFunction(L, "log")) with the reference's probe stream: per-sample values equal
	the oracle's fresh-ring values on the very same probes, and the final estimate follows."""
	from primate_amd.operators import MatrixFunction
	from primate_amd.random import isotropic
	from primate_amd.trace import hutch

	L = laplacian_2d(int(golden["lap_m"]))
	for orth in (0, 3):
		M = MatrixFunction(L, fun="log", deg=20, orth=orth)
		est, info = hutch(M, converge="count", count=40, seed=1234, full=True, batch=8, record=True)
		V = isotropic(size=(L.shape[0], 40), pdf="rademacher", seed=1234)
		ref = oracle.quad_batch(L, V, 20, orth, fun="log", fresh_q=True)
		np.testing.assert_allclose(np.ravel(info.estimator.values), ref, rtol=1e-10)
		assert est == pytest.approx(ref.mean(), rel=1e-12)
		assert hutch(M, converge="count", count=40, seed=1234) == pytest.approx(ref.mean(), rel=1e-12)
	## orth = 0: the reference driver's own number (injected golden; no stale-ring effect at orth 0)
	M0 = MatrixFunction(L, fun="log", deg=20, orth=0)
	ref0 = oracle.quad_batch(L, isotropic(size=(L.shape[0], 40), pdf="rademacher", seed=1234), 20, 0, fun="log", fresh_q=False)
	assert hutch(M0, converge="count", count=40, seed=1234) == pytest.approx(ref0.mean(), rel=1e-12)


def test_lanczos_and_rayleigh_ritz():
	## reference tests/test_lanczos.py
	from scipy.linalg import eigvalsh_tridiagonal

	from primate_amd.lanczos import lanczos, rayleigh_ritz

	rng = np.random.default_rng(seed=1234)
	d = 50
	A = rng.uniform(size=(d, d))
	A @= A.T
	v0 = rng.uniform(size=A.shape[1])
	a, b = lanczos(A, v0=v0, deg=d, orth=d)
	assert np.allclose(eigvalsh_tridiagonal(a, b), np.linalg.eigvalsh(A)), "Eigenvalues not similar"
	A, ew = spd(d, lo=0.0)
	v0 = rng.uniform(size=d)
	rw = rayleigh_ritz(A, 20, v0=v0)
	assert np.isclose(np.max(rw), np.max(ew), atol=1e-2) and np.isclose(np.min(rw), np.min(ew), atol=1e-2)
	rw, rv = rayleigh_ritz(A, 20, v0=v0, return_eigenvectors=True)
	assert np.allclose(rv.T @ rv, np.eye(len(rw)))
	## lanczos-based f(A)v identity of tests/test_operator.py:10-32, deg = n and deg = 5 (A v in K_2)
	A, _ = spd(100)
	v = rng.uniform(size=100, low=-1, high=1)
	for deg in (100, 5):
		(a, b), Q = lanczos(A, v0=v, deg=deg, return_basis=True)
		from scipy.linalg import eigh_tridiagonal

		rw, Y = eigh_tridiagonal(a, b)
		z = np.linalg.norm(v) * Q @ (Y @ (rw * Y[0, :]))
		assert np.isclose(np.linalg.norm(z - A @ v), 0.0, atol=1e-8)


def test_quadrature_api():
	## reference tests/test_quadrature.py:7-20
	from primate_amd.integrate import quadrature
	from primate_amd.lanczos import lanczos

	rng = np.random.default_rng(seed=1234)
	A, _ = spd(50, lo=0.0)
	ests = []
	for _ in range(40):
		v = rng.uniform(size=50, low=0, high=1)
		v /= np.linalg.norm(v)
		a, b = lanczos(A, deg=50, v0=v)
		nodes, weights = quadrature(a, b, deg=30, quad="gw")
		assert nodes.shape == (30,) and np.all(np.diff(nodes) >= 0) and abs(weights.sum() - 1) < 1e-10
		## the rule of the leading 30 x 30 block integrates degree-1 polynomials exactly:
		## sum theta*tau = (T_30)_11 = alpha_0 = v^T A v for the unit vector v
		assert np.sum(nodes * weights) == pytest.approx(v @ A @ v, rel=1e-10)
		ests.append(np.sum(nodes * weights))
	out_n, out_w = np.zeros(30), np.zeros(30)
	quadrature(a, np.append([0], b), deg=30, nodes=out_n, weights=out_w)
	assert np.allclose(out_n, nodes) and np.allclose(out_w, weights)
	with pytest.raises(ValueError):
		quadrature(a, b, quad="nope")


def test_drivers_over_matrix_function(golden):
	"""diag / hutchpp / xtrace / xdiag over a MatrixFunction: the reference's drivers ran over the oracle
	("injected" golden, full reorth so the stale-ring quirk is absent: _matvec clears Q)."""
	from pathlib import Path

	from primate_amd.diagonal import diag
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import hutchpp, xtrace

	gd = np.load(Path(__file__).resolve().parent / "golden" / "slq_golden_drivers.npz")
	L = laplacian_2d(int(gd["lap_m"]))
	M = MatrixFunction(L, fun="exp", deg=20, orth=20, t=-0.1)
	## device accumulation path (count criterion) and the generic host loop give the reference's numbers
	d_dev = diag(M, converge="count", count=30, seed=1234, batch=8)
	np.testing.assert_allclose(d_dev, gd["mf_diag_c30"], rtol=1e-8)
	d_host = diag(M, converge="count", count=30, seed=1234, record=True)
	np.testing.assert_allclose(d_host, gd["mf_diag_c30"], rtol=1e-8)
	d, info = diag(M, converge="count", count=30, seed=1234, full=True, batch=30)
	np.testing.assert_allclose(info.info["numer"] / info.info["denom"], gd["mf_exact_diag"], atol=0.25)
	assert hutchpp(M, m=24, seed=1234, mode="full") == pytest.approx(float(gd["mf_hutchpp_m24_full"]), rel=1e-8)
	M0 = MatrixFunction(L, fun="exp", deg=20, orth=0, t=-0.1)
	assert hutchpp(M0, m=24, seed=1234) == pytest.approx(float(gd["mf0_hutchpp_m24"]), rel=1e-8)
	assert xtrace(M, batch=12, seed=1234) == pytest.approx(float(gd["mf_xtrace_b12"]), rel=1e-7)
	assert xtrace(M, batch=16, seed=5, count=48) == pytest.approx(float(gd["mf_exact_trace"]), rel=3e-2)
	## xdiag over the MatrixFunction (two lock-step Lanczos batches of m / 2 columns) = xdiag over the dense f(A) with the same probes,
	## to the accuracy of the degree-20 Lanczos approximation of exp(-0.1 A) x
	from scipy.linalg import expm

	from primate_amd.diagonal import xdiag

	F = expm(-0.1 * L.toarray())
	np.testing.assert_allclose(xdiag(M, m=40, seed=11), xdiag(F, m=40, seed=11), rtol=1e-7, atol=1e-9)
	assert np.linalg.norm(xdiag(M, m=60, seed=11) - F.diagonal()) < 0.3 * np.linalg.norm(F.diagonal())  # (30 probes on a flat spectrum: measured 0.20)


def test_fttr_quadrature_on_device():
	from pathlib import Path

	from primate_amd.integrate import quadrature

	gd = np.load(Path(__file__).resolve().parent / "golden" / "slq_golden_drivers.npz")
	a, b = gd["fttr_alpha"], gd["fttr_beta"]
	th, w = quadrature(a, b, deg=15, quad="fttr")
	np.testing.assert_allclose(th, gd["fttr_nodes"], atol=1e-13)
	np.testing.assert_allclose(w, gd["fttr_weights"], rtol=1e-9)


def test_stale_ring_compat_reproduces_reference_quad(oracle, golden):
	"""The reference's MatrixFunction.quad carries probe j-1's Lanczos vectors into probe j's
	reorthogonalisation. stale_ring=True reproduces it through the single-vector
	drop-in entry. The chain is numerically unstable for partial reorthogonalisation: rounding
	differences between two correct implementations grow ~50x per probe (measured: 1e-14, 2e-13,
	1e-11 in alpha for probes 0, 1, 2), so exact agreement with the reference's own sequence
	("injected" golden) is only meaningful for the first probes; later ones agree to ~1e-3."""
	from primate_amd.lanczos import _native_lanczos
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import hutch

	L, V = laplacian_2d(int(golden["lap_m"])), golden["lap_probes"]
	M = MatrixFunction(L, fun="log", deg=20, orth=0, stale_ring=True)
	np.testing.assert_allclose(M.quad(V.copy()), golden["mf_quad_log_o0"], rtol=1e-10)  # orth = 0: no ring effect
	for orth in (3, 20):
		M = MatrixFunction(L, fun="log", deg=20, orth=orth, stale_ring=True)
		q, ref, clean = M.quad(V.copy()), golden[f"mf_quad_log_o{orth}"], golden[f"lap_quad_log_o{orth}"]
		np.testing.assert_allclose(q[:3], ref[:3], rtol=1e-8)
		np.testing.assert_allclose(q, ref, rtol=2e-3)
		## it really is the stale behaviour that is reproduced, not the clean one
		assert np.max(np.abs(q[1:3] / clean[1:3] - 1)) > 1e-4 and np.max(np.abs(q[1:3] / ref[1:3] - 1)) < 1e-8
	## the batched default equals the cleared-ring values (pure golden)
	M = MatrixFunction(L, fun="log", deg=20, orth=3)
	np.testing.assert_allclose(M.quad(V.copy()), golden["lap_quad_log_o3"], rtol=1e-10)
	## hutch over the compat operator vs the reference's hutch(MatrixFunction, orth = 3), same call
	## sequence as tests/golden/make_golden.py (two runs on ONE operator: the second starts from the
	## ring the first one left). 40- and 80-probe chains: agreement to the chain's stability.
	for fun in ("log", "exp"):
		M = MatrixFunction(L, fun=fun, deg=20, orth=3, stale_ring=True)
		assert hutch(M, converge="count", count=40, seed=1234) == pytest.approx(float(golden[f"hutch_mf_{fun}_c40"]), rel=1e-3)
		est, info = hutch(M, converge="count", count=40, seed=1234, full=True, batch=8, record=True)
		assert est == pytest.approx(float(golden[f"hutch_mf_{fun}_c40_full"]), rel=1e-3)
		np.testing.assert_allclose(np.ravel(info.estimator.values), golden[f"hutch_mf_{fun}_c40_samples"], rtol=2e-2)
	## the native entry against the oracle with the SAME dirty ring on both sides (one call: stable).
	## With stale columns present the device loop switches to exact MGS order (slq.hip: mgs).
	rng = np.random.default_rng(3)
	for orth, ncv, tol in [(3, 20, 1e-10), (2, 2, 1e-10), (20, 20, 1e-9), (5, 7, 1e-9)]:
		Q0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((L.shape[0], ncv)))[0])
		al, be, Q = np.zeros(21), np.zeros(21), Q0.copy(order="F")
		al2, be2, Q2 = np.zeros(21), np.zeros(21), Q0.copy(order="F")
		s1 = _native_lanczos(L, V[:, 1], 20, 1e-8, orth, al, be, Q)
		s2 = oracle.lanczos(L, V[:, 1], 20, 1e-8, orth, al2, be2, Q2)
		assert s1 == s2
		np.testing.assert_allclose(al, al2, rtol=tol, atol=tol)
		np.testing.assert_allclose(be, be2, rtol=tol, atol=tol)


def test_sharded_drivers_single_rank(golden):
	"""The probe-sharded drivers on one rank (world = 1): same per-probe streams as one big batch."""
	from primate_amd.distributed import sharded_diag_device, sharded_hutch_device
	from primate_amd.engine import DeviceOperator, LanczosPlan

	L = laplacian_2d(30)
	op = DeviceOperator(L)
	cnt, mean, var = sharded_hutch_device(op, 96, 20, 3, fun="log", seed=5)
	plan = LanczosPlan(op, 96, 20, 3)
	plan.generate_probes("rademacher", seed=5)
	plan.run()
	q = plan.quadrature("log")
	assert cnt == 96 and mean == pytest.approx(q.mean(), rel=1e-13) and var == pytest.approx(q.var(ddof=1), rel=1e-10)
	est, numer, denom, c = sharded_diag_device(op, 40, 20, 20, fun="exp", t=-0.1, seed=5, batch=16)
	ew, ev = np.linalg.eigh(L.toarray())
	exact = np.einsum("ij,j,ij->i", ev, np.exp(-0.1 * ew), ev)
	assert c == 40 and np.all(denom == 40) and np.linalg.norm(est - exact) / np.linalg.norm(exact) < 0.15


def test_xtrace_device_rng_and_probe_export():
	"""Device-drawn xtrace probes: the exported sample matrix equals what the plan generator produces
	(sphere draws scaled to norm sqrt(n)), and the estimate agrees with the host-drawn one statistically."""
	from primate_amd.engine import DeviceMatrix, DeviceOperator, LanczosPlan
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import xtrace

	L = laplacian_2d(24)
	n = L.shape[0]
	op = DeviceOperator(L)
	plan = LanczosPlan(op, 12, 20, 3, keep_basis=True)
	W = DeviceMatrix(n, 24, ctx=op.ctx)
	for pdf in ("rademacher", "normal", "sphere"):
		plan.generate_probes(pdf, seed=9, probe_offset=4)
		X = plan.get_probes()
		plan.get_probes_into(W, 12)
		G = W.get(12, 12)
		if pdf == "sphere":
			np.testing.assert_allclose(np.linalg.norm(G, axis=0), np.sqrt(n), rtol=1e-13)
			np.testing.assert_allclose(G, X * (np.sqrt(n) / np.linalg.norm(X, axis=0)), rtol=1e-13)
		else:
			np.testing.assert_array_equal(G, X)
	M = MatrixFunction(L, fun="exp", deg=20, orth=3, t=-0.5)
	exact = np.sum(np.exp(-0.5 * np.linalg.eigvalsh(L.toarray())))
	a = xtrace(M, batch=16, seed=3, count=64, device_rng=True)
	b = xtrace(M, batch=16, seed=3, count=64)
	assert abs(a - exact) / exact < 2e-2 and abs(b - exact) / exact < 2e-2
	## same seed, same stream: reproducible
	assert a == xtrace(M, batch=16, seed=3, count=64, device_rng=True)


def test_hutch_device_drawn_probes():
	"""hutch(pdf="device:rademacher"): same values as evaluating the generated probes explicitly, and the
	stream does not depend on the batch size."""
	from primate_amd.engine import DeviceOperator, LanczosPlan
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import hutch

	L = laplacian_2d(30)
	M = MatrixFunction(L, fun="log", deg=20, orth=3)
	a = hutch(M, pdf="device:rademacher", converge="count", count=96, seed=11, batch=32)
	b, info = hutch(M, pdf="device:rademacher", converge="count", count=96, seed=11, batch=48, full=True)
	op = DeviceOperator(L)
	plan = LanczosPlan(op, 96, 20, 3)
	plan.generate_probes("rademacher", seed=11)
	X = plan.get_probes()
	q = M.quad(X)
	assert a == pytest.approx(q.mean(), rel=1e-12) and b == pytest.approx(q.mean(), rel=1e-12)
	assert info.nit == 96
	exact = np.sum(np.log(np.linalg.eigvalsh(L.toarray())))
	assert abs(a - exact) / abs(exact) < 0.05


def test_hutchpp_device_path_matches_host_algebra(monkeypatch):
	"""hutchpp over a device MatrixFunction keeps its sketches in HBM; same probes (NumPy stream), so it
	equals the reference's host algebra (QR on the host, A @ G through _matmat) to rounding."""
	import primate_amd.trace as T
	from primate_amd.operators import MatrixFunction

	L = laplacian_2d(26)
	M = MatrixFunction(L, fun="exp", deg=20, orth=5, t=-0.3)
	exact = np.sum(np.exp(-0.3 * np.linalg.eigvalsh(L.toarray())))
	for mode in ("reduced", "full"):
		dev, info = T.hutchpp(M, m=45, mode=mode, seed=5, full=True)
		with monkeypatch.context() as mp:
			mp.setattr(T, "_hutchpp_device", None)  # the host path must not need it
			mp.setattr(M, "_builtin", None)  # host algebra: QR and deflation in NumPy
			host = T.hutchpp(M, m=45, mode=mode, seed=5)
		assert info.nit == 90 and info.samples.shape == (90,)
		assert dev == pytest.approx(host, rel=1e-9), mode
		assert abs(dev - exact) / exact < 2e-2


def test_tridiag_fttr_and_isotropic_modules_on_device():
	"""`primate.tridiag` / `primate.fttr` names and the device mode of `Isotropic`, against vectors captured from
	the reference (LAPACK MRRR behind its eigh_tridiag) and against the plan generator."""
	from primate_amd.engine import DeviceOperator, LanczosPlan
	from primate_amd.fttr import fttr, ortho_poly
	from primate_amd.random import Isotropic
	from primate_amd.tridiag import eigh_tridiag, eigvalsh_tridiag

	gd = np.load(ROOT / "tests" / "golden" / "slq_golden_drivers.npz")
	d, e = gd["tri_d"], gd["tri_e"]
	w, Z = eigh_tridiag(d, e)
	np.testing.assert_allclose(w, gd["tri_w"], rtol=0, atol=2e-14)
	np.testing.assert_allclose(np.abs(Z), gd["tri_absZ"], rtol=0, atol=1e-12)  # eigenvectors up to sign
	T = np.diag(d) + np.diag(e[1:], 1) + np.diag(e[1:], -1)
	np.testing.assert_allclose(T @ Z, Z * w, atol=1e-13)
	np.testing.assert_allclose(Z.T @ Z, np.eye(len(d)), atol=1e-13)
	np.testing.assert_allclose(eigvalsh_tridiag(d, e[1:]), gd["tri_w_only"], rtol=0, atol=2e-14)
	with pytest.raises(AssertionError):
		eigh_tridiag(d, e[:5])
	## the reference's own test (tests/test_tridiagonal.py:11-44): d = 150 needs the global-scratch variant
	from primate_amd.lanczos import lanczos
	from primate_amd.random import symmetric
	from primate_amd.tqli import tqli

	for seed in [1234, 43]:
		rng = np.random.default_rng(seed)
		k = 150
		ew = np.sort(rng.uniform(size=k, low=1 / k, high=1))
		A = symmetric(k, seed=rng, pd=True, ew=ew)
		a, b = lanczos(A, v0=rng.uniform(size=k), deg=k, orth=k)
		for method in ["tqli", "mrrr"]:
			assert np.max(np.abs(np.sort(eigvalsh_tridiag(a, b, method=method)) - ew)) <= 1e-13
			ew_t, ev_t = eigh_tridiag(a, b, method=method)
			np.testing.assert_allclose(ev_t.T @ ev_t, np.eye(k), atol=1e-12)
			np.testing.assert_allclose(np.sort(ew_t), ew, atol=1e-13)
		dd, ee, Zt = a.copy(), np.append([0], b), np.eye(k)
		tqli(dd, ee, Zt, 30)
		assert np.allclose(np.sort(dd), ew) and np.allclose(ee, 0.0)
		Tm = np.diag(a) + np.diag(b, 1) + np.diag(b, -1)
		np.testing.assert_allclose(Tm @ Zt, Zt * dd, atol=1e-12)
	## fttr with the reference's in-place signature, on the golden rule
	wts = np.zeros(len(gd["fttr_nodes"]))
	fttr(gd["fttr_nodes"], gd["fttr_alpha"], gd["fttr_beta"], len(wts), wts)
	np.testing.assert_allclose(wts, gd["fttr_weights"], rtol=1e-10)
	z = np.zeros(len(gd["fttr_alpha"]))
	mu0 = np.sum(np.abs(gd["fttr_nodes"]))
	ortho_poly(gd["fttr_nodes"][0], 1 / np.sqrt(mu0), gd["fttr_alpha"], gd["fttr_beta"], z, len(z))
	assert 1.0 / (mu0 * np.sum(z * z)) == pytest.approx(gd["fttr_weights"][0], rel=1e-10)
	## Isotropic(device=True): same stream as the plan generator, ids advancing with every fill
	L = laplacian_2d(12)
	n = L.shape[0]
	plan = LanczosPlan(DeviceOperator(L), 6, 5, 0)
	for pdf in ("signs", "normal", "sphere"):
		iso = Isotropic((n, 6), pdf=pdf, seed=21, device=True)
		for rep in range(2):
			iso.fill()
			plan.generate_probes({"signs": "rademacher"}.get(pdf, pdf), seed=21, probe_offset=6 * rep)
			X = plan.get_probes()
			if pdf == "sphere":
				X = X * (np.sqrt(n) / np.linalg.norm(X, axis=0))
			np.testing.assert_allclose(iso.values, X, rtol=1e-13)


def test_toeplitz_plugin_through_the_callback_operator():
	"""A matrix-free host plugin (FFT-applied Toeplitz) driven by the device Lanczos through the callback operator:
	same tridiagonal and quadrature as the dense matrix it represents."""
	from scipy.linalg import toeplitz

	from primate_amd.lanczos import lanczos
	from primate_amd.operators import MatrixFunction, Toeplitz

	n = 64
	c = 0.5 ** np.arange(n)  # Kac-Murdock-Szego matrix: SPD
	T, Td = Toeplitz(c), toeplitz(c)
	v = np.random.default_rng(0).standard_normal(n)
	(a1, b1), (a2, b2) = lanczos(T, v0=v, deg=20, orth=5), lanczos(Td, v0=v, deg=20, orth=5)
	np.testing.assert_allclose(a1, a2, rtol=1e-9, atol=1e-11)
	np.testing.assert_allclose(b1, b2, rtol=1e-9, atol=1e-11)
	M1, M2 = MatrixFunction(T, fun="log", deg=20), MatrixFunction(Td, fun="log", deg=20)
	X = np.random.default_rng(1).standard_normal((n, 6))
	np.testing.assert_allclose(M1.quad(X), M2.quad(X), rtol=1e-9)


def test_gram_and_affine_sparse_operators(oracle):
	"""The two native operators of the reference's plugin header that its Python module never binds
	(eigen_operators.h:57-72 gram = true, :106-137 affine): x -> A^T (A x) for a rectangular sparse A, and A + t B with a
	parameter that changes after creation. Oracle: the reference recurrence over a Python callback that applies the same
	product (pylinop.h:32-40), same probes."""
	import scipy.sparse as sp

	from primate_amd.engine import DeviceOperator, quad_batch
	from primate_amd.operators import AffineOperator, GramOperator, MatrixFunction

	rng = np.random.default_rng(17)
	B = sp.random(700, 400, density=0.02, random_state=3, format="csr", dtype=np.float64)
	G = GramOperator(B)
	assert G.shape == (400, 400)
	X = np.asfortranarray(rng.standard_normal((400, 70)))
	dense = (B.T @ B).toarray()

	class GramPy:  # what the reference's PyLinearOperator would be handed
		shape, dtype = (400, 400), np.dtype(np.float64)

		def matvec(self, x):
			return B.T @ (B @ x)

	op = DeviceOperator(G)
	np.testing.assert_allclose(op.matmat(X), dense @ X, rtol=1e-12, atol=1e-12)
	for orth in (0, 3, 25):
		ref = oracle.quad_batch(GramPy(), np.asfortranarray(X[:, :6]), 25, orth, fun="sqrt", fresh_q=True)
		## (A^T A of a random sparse A has a small-eigenvalue tail that sqrt amplifies: measured 1.5e-9; the bar is 1e-6)
		np.testing.assert_allclose(quad_batch(op, X, 25, orth, fun="sqrt")[:6], ref, rtol=1e-7, err_msg=f"orth={orth}")
	## numerical rank of a rectangular matrix of known rank through numrank (special.py:103-105) of its Gram operator:
	## every probe's Gauss rule counts the same number of eigenvalues above the threshold once k exceeds the rank
	U, V = rng.standard_normal((300, 12)), rng.standard_normal((12, 200))
	R = sp.csr_matrix(U @ V)
	M = MatrixFunction(GramOperator(R), fun="numrank", deg=40, orth=40)
	q = M.quad(np.asfortranarray(rng.standard_normal((200, 16))))
	np.testing.assert_allclose(q.mean(), 12.0, rtol=0.35)  # tr step(A^T A) = rank; 16 Gaussian probes: ~25 % standard error
	## affine: A + t B on the union pattern, t changed in place
	from conftest import laplacian_2d

	A0 = laplacian_2d(24)
	n = A0.shape[0]
	Bm = sp.diags([rng.uniform(0.5, 1.5, n), rng.uniform(-0.2, 0.2, n - 7), rng.uniform(-0.2, 0.2, n - 7)], [0, 7, -7]).tocsr()
	Bm = ((Bm + Bm.T) * 0.5).tocsr()
	Aff = AffineOperator(A0, Bm)
	Mf = MatrixFunction(Aff, fun="log", deg=20, orth=3)
	Xa = np.asfortranarray(np.floor(rng.random((n, 9)) * 2) * 2 - 1)
	for t in (0.0, 0.75, -0.25, 0.0):
		Aff.set_parameter(t)  # reaches the device operator inside Mf
		ref = oracle.quad_batch((A0 + t * Bm).tocsr(), Xa, 20, 3, fun="log", fresh_q=True)
		np.testing.assert_allclose(Mf.quad(Xa), ref, rtol=1e-10, err_msg=f"t={t}")
		np.testing.assert_allclose(Mf._op.matmat(Xa), (A0 + t * Bm) @ Xa, rtol=1e-12, atol=1e-12)


def test_torch_plugin_operator_stays_on_the_device():
	"""A LinearOperator plugin written in torch (GPU tensors in, GPU tensors out) inside the device Lanczos loop:
	same tridiagonal, quadrature and f(A)v as the dense matrix it wraps; errors raised in the plugin surface."""
	import torch

	from primate_amd.engine import DeviceOperator, quad_batch
	from primate_amd.lanczos import lanczos
	from primate_amd.operators import MatrixFunction, TorchOperator

	A, _ = spd(300, seed=5)
	At = torch.tensor(A, device="cuda")
	T = TorchOperator(lambda X: At @ X, 300)
	v = np.random.default_rng(0).standard_normal(300)
	(a1, b1), (a2, b2) = lanczos(T, v0=v, deg=25, orth=25), lanczos(A, v0=v, deg=25, orth=25)
	np.testing.assert_allclose(a1, a2, rtol=1e-9, atol=1e-12)
	np.testing.assert_allclose(b1, b2, rtol=1e-9, atol=1e-12)
	X = np.asfortranarray(np.random.default_rng(1).standard_normal((300, 70)))
	op_t, op_d = DeviceOperator(T), DeviceOperator(A)
	assert op_t.kind == "device_callback"
	np.testing.assert_allclose(quad_batch(op_t, X, 20, 3, fun="log"), quad_batch(op_d, X, 20, 3, fun="log"), rtol=1e-10)
	np.testing.assert_allclose(op_t.matmat(X), A @ X, rtol=1e-12, atol=1e-12)
	M = MatrixFunction(T, fun="exp", deg=20, t=-1.0)
	w, U = np.linalg.eigh(A)
	np.testing.assert_allclose(M @ X[:, :3], (U * np.exp(-w)) @ (U.T @ X[:, :3]), rtol=1e-8, atol=1e-10)

	def boom(Xt):
		raise RuntimeError("plugin failure")

	with pytest.raises(RuntimeError, match="plugin failure"):
		quad_batch(DeviceOperator(TorchOperator(boom, 300)), X, 5, 0)


def test_device_resident_csr_operator_is_built_like_a_host_one():
	"""slq_csr_create_device (a torch sparse-CSR tensor on the GPU): the operator goes through the same analysis as a scipy
	matrix - reordering, upper triangle, LDS tiles - so its results are the host-created operator's bit for bit, and bad
	indices are refused."""
	from primate_amd import engine as eng

	torch = pytest.importorskip("torch")
	A = laplacian_2d(300)  # 90,000 rows: takes the tiles by default
	T = torch.sparse_csr_tensor(torch.from_numpy(A.indptr.astype(np.int64)), torch.from_numpy(A.indices.astype(np.int64)), torch.from_numpy(A.data), size=A.shape).cuda()
	op_d, op_h = eng.DeviceOperator(T), eng.DeviceOperator(A)
	assert op_d.kind == "csr" and op_d.nnz == A.nnz and op_d.dtype == np.float64
	pd, ph = eng.LanczosPlan(op_d, 130, 10, 3), eng.LanczosPlan(op_h, 130, 10, 3)
	assert pd.describe() == ph.describe() and pd.describe()["tiles"] == 2
	for p in (pd, ph):
		p.generate_probes("rademacher", seed=5)
		p.run()
	assert np.array_equal(pd.quadrature("log"), ph.quadrature("log"))
	pd.close(), ph.close(), op_d.close(), op_h.close()
	## r04: with ring tiles the stored arrays are built on the device straight from the tensor's arrays (the values never visit the host);
	## an operator without tiles - a small random graph - is still built on the host, from values fetched then
	import scipy.sparse as sp

	G = sp.random(5000, 5000, density=0.002, random_state=3, format="csr")
	G = (G + G.T + sp.identity(5000) * 8.0).tocsr()
	G.sort_indices()
	Tg = torch.sparse_csr_tensor(torch.from_numpy(G.indptr.astype(np.int64)), torch.from_numpy(G.indices.astype(np.int64)), torch.from_numpy(G.data), size=G.shape).cuda()
	og, oh = eng.DeviceOperator(Tg), eng.DeviceOperator(G)
	pg, ph = eng.LanczosPlan(og, 12, 10, 3), eng.LanczosPlan(oh, 12, 10, 3)
	assert pg.describe() == ph.describe() and pg.describe()["tiles"] == 0
	for p in (pg, ph):
		p.generate_probes("rademacher", seed=5)
		p.run()
	assert np.array_equal(pg.quadrature("log"), ph.quadrature("log"))
	pg.close(), ph.close(), og.close(), oh.close()
	bad = torch.sparse_csr_tensor(torch.tensor([0, 1, 2]), torch.tensor([0, 5]), torch.tensor([1.0, 1.0]), size=(2, 2), check_invariants=False).cuda()
	with pytest.raises(ValueError):
		eng.DeviceOperator(bad)


def test_ring_bail_out_flag_reaches_every_accessor():
	"""A workgroup of the ring-fed tile pass whose bounded wait runs out raises the plan's device word; every accessor that
	hands results to the host must then report SLQ_EHIP, not numbers. The word is poked directly (no kernel is stalled on
	the box to get there); a plan whose word is clear returns results as usual, and a poked plan stays dead."""
	from primate_amd import _capi
	from primate_amd.engine import DeviceMatrix, DeviceOperator, DiagAccumulator, LanczosPlan

	A = laplacian_2d(40)
	n = A.shape[0]
	rng = np.random.default_rng(5)
	X = np.asfortranarray(np.floor(rng.random((n, 6)) * 2) * 2 - 1)
	op = DeviceOperator(A)
	plan = LanczosPlan(op, 6, 10, 3, keep_basis=True)
	plan.set_probes(X)
	plan.run()
	good = plan.quadrature("log")
	assert np.all(np.isfinite(good)) and plan.tridiag()[2].tolist() == [10] * 6
	_capi.check(_capi.lib().slq_debug_plan_poke_ring_flag(plan._h, 1))
	out, acc = DeviceMatrix(n, 6), DiagAccumulator(n)
	calls = {
		"tridiag": plan.tridiag, "quadrature": lambda: plan.quadrature("log"), "basis": lambda: plan.basis(0),
		"fun_action": lambda: plan.fun_action("exp"), "fun_action_into": lambda: plan.fun_action_into(out, 0, "exp"),
		"diag_update": lambda: acc.update(plan, "exp"),
	}  # fmt: skip
	for name, call in calls.items():
		with pytest.raises(_capi.SlqError) as ei:
			call()
		assert ei.value.code == _capi.SLQ_EHIP and "ring-fed tile pass" in str(ei.value), name
	## a second run on the same plan does not clear it: the plan is dead
	plan.set_probes(X)
	plan.run()
	with pytest.raises(_capi.SlqError):
		plan.quadrature("log")
	out.close(), acc.close(), plan.close()
	## a fresh plan on the same operator is unaffected
	np.testing.assert_array_equal(quad_fresh(op, X), good)
	op.close()


def quad_fresh(op, X):
	from primate_amd.engine import LanczosPlan

	plan = LanczosPlan(op, X.shape[1], 10, 3, keep_basis=True)
	plan.set_probes(X)
	plan.run()
	q = plan.quadrature("log")
	plan.close()
	return q
