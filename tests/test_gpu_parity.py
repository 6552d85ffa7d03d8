"""Parity of the HIP path with the oracle and with the golden vectors, through the C-ABI.

Every test here needs a real MI355X (`-m gpu`). Nothing reads /root/reference. Tolerances: fp64
per-probe quadratic forms 1e-10 relative against the oracle on identical probes (the north_star
bar is 1e-6); fp32 2e-4; integer structure (steps, shapes, nnz, Rademacher support) bit-exact.
"""

import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import laplacian_2d, laplacian_3d

pytestmark = pytest.mark.gpu

FUNS = {
	"identity": {}, "log": {}, "exp": {}, "sqrt": {}, "inv": {}, "abs": {},
	"smoothstep": {"a": 0.5, "b": 6.0}, "numrank": {},
}  # fmt: skip


@pytest.fixture(scope="module")
def eng():
	from primate_amd import engine

	return engine


@pytest.fixture(scope="module")
def lap(golden, eng):
	L = laplacian_2d(int(golden["lap_m"]))
	return L, eng.DeviceOperator(L), golden["lap_probes"]


def random_spd_graph(n, avg_deg, seed, dtype=np.float64):
	"""Symmetric graph Laplacian + I of a G(n, p) graph: SPD, irregular row degrees, some empty rows."""
	rng = np.random.default_rng(seed)
	m = int(n * avg_deg / 2)
	i, j = rng.integers(0, n, m), rng.integers(0, n, m)
	keep = i != j
	W = sp.coo_matrix((rng.uniform(0.5, 2.0, keep.sum()), (i[keep], j[keep])), shape=(n, n))
	W = (W + W.T).tocsr()
	W.sum_duplicates()
	Lg = sp.diags(np.asarray(W.sum(axis=1)).ravel() + 1.0) - W
	A = Lg.tocsr().astype(dtype)
	A.sort_indices()
	return A


@pytest.mark.parametrize("orth", [0, 3, 20])
def test_golden_laplacian_all_functions(lap, golden, eng, orth):
	L, op, V = lap
	for fun, kw in FUNS.items():
		q = eng.quad_batch(op, V, 20, orth, fun=fun, **kw)
		np.testing.assert_allclose(q, golden[f"lap_quad_{fun}_o{orth}"], rtol=1e-11, err_msg=fun)
	q = eng.quad_batch(op, V, 20, orth, fun="exp", t=-0.1)
	np.testing.assert_allclose(q, golden["lap_quad_exp_t_o%d" % orth], rtol=1e-11)


@pytest.mark.parametrize("orth", [0, 3, 20])
def test_golden_laplacian_tridiag_and_rule(lap, golden, eng, orth):
	L, op, V = lap
	plan = eng.LanczosPlan(op, V.shape[1], 20, orth)
	plan.set_probes(V)
	plan.run()
	a, b, steps = plan.tridiag()
	assert np.array_equal(steps, np.full(V.shape[1], 20))  # integer: bit-exact
	np.testing.assert_allclose(a[:, :20], golden[f"lap_alpha_o{orth}"], rtol=0, atol=1e-12)
	np.testing.assert_allclose(b[:, :20], golden[f"lap_beta_o{orth}"], rtol=0, atol=1e-12)
	assert np.all(b[:, 0] == 0)
	q, nodes, weights = plan.quadrature("log", return_rule=True)
	np.testing.assert_allclose(nodes, golden[f"lap_nodes_o{orth}"], rtol=0, atol=1e-12)
	np.testing.assert_allclose(weights, golden[f"lap_weights_o{orth}"], rtol=0, atol=1e-12)
	np.testing.assert_allclose(weights.sum(axis=1), 1.0, atol=1e-13)


def test_golden_kat_dense_full_reorth(golden, eng):
	## inputs of the reference's tests/test_lanczos.py:11-20; dense operator plugin
	A, v0 = golden["kat_A"], golden["kat_v0"]
	op = eng.DeviceOperator(A)
	plan = eng.LanczosPlan(op, 1, 50, 50, keep_basis=True)
	plan.set_probes(v0)
	plan.run()
	a, b, steps = plan.tridiag()
	assert steps[0] == 50
	np.testing.assert_allclose(a[0, :50], golden["kat_alpha_o50_c50"], rtol=0, atol=1e-11 * np.abs(a).max())
	np.testing.assert_allclose(b[0, :50], golden["kat_beta_o50_c50"], rtol=0, atol=1e-11 * np.abs(a).max())
	Q = plan.basis(0)
	np.testing.assert_allclose(np.abs(Q.T @ golden["kat_Q_o50_c50"]), np.eye(50), atol=1e-8)
	np.testing.assert_allclose(Q.T @ Q, np.eye(50), atol=1e-10)
	from scipy.linalg import eigvalsh_tridiagonal

	assert np.allclose(eigvalsh_tridiagonal(a[0, :50], b[0, 1:50]), golden["kat_eigvalsh"])


def test_golden_early_stop(golden, eng):
	op = eng.DeviceOperator(golden["stop_A"])
	plan = eng.LanczosPlan(op, 1, 20, 20)
	plan.set_probes(golden["stop_v"])
	plan.run()
	a, b, steps = plan.tridiag()
	assert steps[0] == 5  # lanczos.h:140-142, bit-exact step count
	np.testing.assert_allclose(b[0, :5], golden["stop_beta"][:5], rtol=1e-10)
	np.testing.assert_allclose(a[0, :5], golden["stop_alpha"][:5], rtol=1e-10)
	assert b[0, 5] < np.sqrt(40) * 1e-8 and np.all(b[0, 6:] == 0) and np.all(a[0, 5:] == 0)
	## the quadrature of the zero-tailed T equals v^T A v (f = identity)
	q = plan.quadrature("identity")
	v = golden["stop_v"]
	np.testing.assert_allclose(q[0], v @ golden["stop_A"] @ v, rtol=1e-10)


def test_c2_anchor_full_size(golden, eng):
	"""BASELINE.json configs[1] at full size, one probe: the reference twin's value for the
	seed-1234 Rademacher probe, plus the bit-exact integer structure."""
	from primate_amd.random import isotropic

	L2 = laplacian_2d(1000)
	assert [L2.nnz, L2.shape[0]] == list(golden["c2_nnz_n"])
	assert list(np.bincount(np.diff(L2.indptr))) == list(golden["c2_rowdeg_hist"])
	op = eng.DeviceOperator(L2)
	assert (op.shape, op.nnz) == ((1000000, 1000000), 4996000)
	v = isotropic(pdf="rademacher", seed=1234)(size=(L2.shape[0], 1))
	for k, orth in enumerate([0, 3]):
		q = eng.quad_batch(op, v, 30, orth, fun="log")
		assert abs(q[0] / golden["c2_quad_log_seed1234_o0_o3"][k] - 1) < 1e-10


def test_step_function_with_a_cut_inside_the_spectrum(oracle, eng, lap):
	"""f = step(c) (src/primate/special.py:69-74; numrank is step(1e-6, nonnegative=True), :103-105) with the cut INSIDE the
	spectrum, where it actually removes Ritz values: per-probe sums against the oracle, both senses of `nonnegative`, and
	the counting identity sum f(theta) tau = sum of the weights of the nodes at or above the cut."""
	L, op, V = lap
	for orth in (0, 3, 20):
		for kw in ({"c": 3.9173}, {"c": 1.2345, "nonnegative": True}, {"c": 7.5}):
			got = eng.quad_batch(op, V, 20, orth, fun="step", **kw)
			ref = oracle.quad_batch(L, V, 20, orth, fun="step", fresh_q=True, **kw)
			np.testing.assert_allclose(got, ref, rtol=1e-10, err_msg=f"orth={orth} {kw}")
		plan = eng.LanczosPlan(op, V.shape[1], 20, orth)
		plan.set_probes(V)
		plan.run()
		q, nodes, weights = plan.quadrature("step", return_rule=True, c=3.9173)
		np.testing.assert_allclose(q, (weights * (nodes >= 3.9173)).sum(axis=1) * (V * V).sum(axis=0), rtol=1e-12)
		assert np.all(q > 0) and np.all(q < (V * V).sum(axis=0))  # the cut bites: neither nothing nor everything is counted
		plan.close()


@pytest.mark.parametrize("dtype,rtol", [(np.float64, 1e-10), (np.float32, 3e-4)])
@pytest.mark.parametrize("orth", [0, 3, 25])
def test_oracle_random_graph_ragged(oracle, eng, dtype, rtol, orth):
	## irregular degrees, empty-ish rows, n not a multiple of anything, ragged probe count
	A = random_spd_graph(2003, 6.0, seed=11, dtype=dtype)
	rng = np.random.default_rng(5)
	X = np.asfortranarray(rng.standard_normal((A.shape[0], 37)).astype(dtype))
	op = eng.DeviceOperator(A)
	for fun, kw in [("log", {}), ("exp", {"t": -0.05}), ("sqrt", {})]:
		got = eng.quad_batch(op, X, 25, orth, fun=fun, **kw)
		ref = oracle.quad_batch(A, X, 25, orth, fun=fun, fresh_q=True, prefer="csr", **kw)
		np.testing.assert_allclose(got, ref, rtol=rtol, err_msg=f"{fun} orth={orth}")


@pytest.mark.parametrize("orth", [1, 2, 4, 5, 6, 7, 8, 9, 11])
def test_every_ring_column_count(oracle, eng, orth):
	"""The fused passes are specialised on the number of ring columns (1..8) and hand over to the
	store-and-revisit sweeps from r = 9 on, in the middle of a run: every variant and the switch-over."""
	A = random_spd_graph(1234, 7.0, seed=orth)
	rng = np.random.default_rng(orth)
	for dtype, rtol in [(np.float64, 1e-10), (np.float32, 3e-4)]:
		Ad = A.astype(dtype)
		X = np.asfortranarray(rng.standard_normal((A.shape[0], 70)).astype(dtype))
		op = eng.DeviceOperator(Ad)
		got = eng.quad_batch(op, X, 16, orth, fun="exp", t=-0.1)
		ref = oracle.quad_batch(Ad, X, 16, orth, fun="exp", t=-0.1, fresh_q=True, prefer="csr")
		np.testing.assert_allclose(got, ref, rtol=rtol, err_msg=f"{dtype.__name__} orth={orth}")
		if dtype == np.float64:
			plan = eng.LanczosPlan(op, 70, 16, orth)
			plan.set_probes(X)
			plan.run()
			a, b, steps = plan.tridiag()
			ar, br, Qr = np.zeros(17), np.zeros(17), np.zeros((A.shape[0], max(orth, 2)), order="F")
			oracle.lanczos(A, X[:, 33].copy(), 16, 1e-8, orth, ar, br, Qr)
			np.testing.assert_allclose(a[33][:16], ar[:16], rtol=1e-9, atol=1e-11)
			np.testing.assert_allclose(b[33][1:16], br[1:16], rtol=1e-9, atol=1e-11)
		op.close()


@pytest.mark.parametrize("nprobes", [1, 2, 15, 16, 17, 64, 129, 300])
def test_probe_count_edges(oracle, eng, nprobes):
	## every panel geometry (PW = 16..128, one or several panels, ragged last panel)
	A = laplacian_2d(17)
	rng = np.random.default_rng(nprobes)
	X = np.asfortranarray(np.floor(rng.random((A.shape[0], nprobes)) * 2) * 2 - 1)
	op = eng.DeviceOperator(A)
	## (no tiles at this size: the generic passes, on the Gram sequence since r04 - k_csr_pass<PASS_UPDATEG>)
	plan = eng.LanczosPlan(op, nprobes, 12, 3)
	assert plan.describe()["sequence"] == "fused_gram" and plan.describe()["tiles"] == 0, plan.describe()
	plan.close()
	got = eng.quad_batch(op, X, 12, 3, fun="log")
	ref = oracle.quad_batch(A, X, 12, 3, fun="log", fresh_q=True)
	np.testing.assert_allclose(got, ref, rtol=1e-10)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-10), (np.float32, 3e-4)])
def test_gram_sequence_on_the_generic_passes(oracle, eng, monkeypatch, dtype, tol):
	"""The Gram sequence where the plan's fused passes are the generic ones (r04: k_csr_pass<PASS_UPDATEG>; operators below the tiles' size,
	panels of 8 lanes per row, the single-vector lanczos() - src/primate/lanczos.py:109; hutch draws one probe per iteration,
	src/primate/trace.py:104-116): every window 1..8 and a window that hands over to the sweeps (orth 12), panels of 1 / 7 / 16 / 40 / 130
	probes (all four panel widths, several rows per wave and one), a 2-D and a 3-D grid and an irregular graph with empty rows whose rows
	end in 1, 2 and 3 leftover entries (the masked last batch of the narrow panels' row loop), against the oracle; SLQ_GRAM_CSR=0 (the
	merged alpha+dots sequence) on the same inputs. lanczos.h:43-66,127-136."""
	rng = np.random.default_rng(8)
	ops = [laplacian_2d(60), laplacian_3d(12), random_spd_graph(3000, 5.0, seed=17)]
	for A in ops:
		A = A.astype(dtype)
		n = A.shape[0]
		op = eng.DeviceOperator(A)
		for P in (1, 7, 16, 40, 130):
			X = np.asfortranarray(rng.standard_normal((n, P))).astype(dtype)
			cols = sorted({0, P // 2, P - 1})
			for orth in (1, 2, 3, 5, 8, 12):
				ref = oracle.quad_batch(A, np.asfortranarray(X[:, cols]), 16, orth, fun="exp", t=-0.05, fresh_q=True, prefer="csr")
				for gram in ("1", "0"):
					monkeypatch.setenv("SLQ_GRAM_CSR", gram)
					plan = eng.LanczosPlan(op, P, 16, orth)
					assert plan.describe()["sequence"] == ("fused_gram" if gram == "1" else "fused") and plan.describe()["tiles"] == 0
					plan.set_probes(X)
					plan.run()
					np.testing.assert_allclose(plan.quadrature("exp", t=-0.05)[cols], ref, rtol=tol, err_msg=f"n={n} P={P} orth={orth} gram={gram}")
					plan.close()
				monkeypatch.delenv("SLQ_GRAM_CSR")
		op.close()


def test_tiny_and_degenerate_operators(oracle, eng):
	## n smaller than deg (deg clamps to n, lanczos.py:79), n = 2, diagonal matrix with a zero row
	for n in [2, 3, 7]:
		A = sp.diags(np.arange(1.0, n + 1)).tocsr()
		X = np.asfortranarray(np.ones((n, 3)) + np.arange(3))
		got = eng.quad_batch(eng.DeviceOperator(A), X, 20, 0, fun="identity")
		np.testing.assert_allclose(got, np.einsum("ij,ij->j", X, A @ X), rtol=1e-10)
	A = sp.csr_matrix(np.diag([2.0, 0.0, 3.0, 1.0]))  # row 1 is empty
	X = np.asfortranarray(np.array([[1.0, 2.0, -1.0, 0.5]]).T)
	got = eng.quad_batch(eng.DeviceOperator(A), X, 4, 4, fun="identity")
	np.testing.assert_allclose(got, X[:, 0] @ (A @ X[:, 0]), rtol=1e-10)
	## an all-zero probe is 0/0 in the reference (lanczos.h:120): NaN, and its neighbours are untouched
	A = laplacian_2d(5)
	X = np.asfortranarray(np.ones((25, 3)))
	X[:, 1] = 0
	got = eng.quad_batch(eng.DeviceOperator(A), X, 10, 3, fun="identity")
	assert np.isnan(got[1]) and np.allclose(got[[0, 2]], np.ones(25) @ (A @ np.ones(25)))


def test_operator_products(eng):
	rng = np.random.default_rng(8)
	A = random_spd_graph(501, 5.0, seed=3)
	X = np.asfortranarray(rng.standard_normal((501, 21)))
	ref = A @ X
	np.testing.assert_allclose(eng.DeviceOperator(A).matmat(X), ref, rtol=1e-12, atol=1e-12)
	np.testing.assert_allclose(eng.DeviceOperator(A.toarray()).matmat(X), ref, rtol=1e-11, atol=1e-11)

	class PyOp:  # the Python plugin surface (pylinop.h:22-29): .matvec + .shape (+ dtype)
		shape, dtype = A.shape, np.dtype(np.float64)
		calls = 0

		def matvec(self, v):
			PyOp.calls += 1
			return A @ v

	np.testing.assert_allclose(eng.DeviceOperator(PyOp()).matmat(X), ref, rtol=1e-12, atol=1e-12)
	assert PyOp.calls == 21
	A32 = A.astype(np.float32)
	np.testing.assert_allclose(eng.DeviceOperator(A32).matmat(X.astype(np.float32)), ref, rtol=2e-5, atol=2e-5)
	## a dense operator's product is A X for WHATEVER is given (eigen_operators.h:24-30), also a non-symmetric array, on
	## the fp64 matrix-core kernel and on the fp32 column-walking one
	G = rng.standard_normal((301, 301))
	Xg = np.asfortranarray(rng.standard_normal((301, 21)))
	np.testing.assert_allclose(eng.DeviceOperator(G).matmat(Xg), G @ Xg, rtol=1e-11, atol=1e-11)
	np.testing.assert_allclose(eng.DeviceOperator(G.astype(np.float32)).matmat(Xg.astype(np.float32)), G @ Xg, rtol=3e-4, atol=3e-4)


@pytest.mark.parametrize("n,P", [(516, 64), (517, 64), (700, 130), (257, 20), (300, 5), (1030, 300)])
def test_dense_fp32_operator_on_the_matrix_cores(oracle, eng, n, P):
	"""fp32 dense operator (eigen_operators.h:24-30 with F = float; _lanczos.cpp:104): k_dense_mfma32_lds, 256-row tiles over
	a K split, against the oracle on the same fp32 arrays - every panel width (64 / 128 / 256 columns, two panels at P = 300),
	aligned and unaligned leading dimensions (16-byte and element loads), ragged last row tile, a non-symmetric array."""
	rng = np.random.default_rng(n + P)
	B = rng.standard_normal((n, n))
	A = (B @ B.T / n + np.eye(n)).astype(np.float32)
	X = np.asfortranarray(rng.standard_normal((n, P)).astype(np.float32))
	op = eng.DeviceOperator(A)
	ref = A.astype(np.float64) @ X.astype(np.float64)
	np.testing.assert_allclose(op.matmat(X), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())  # an f32 fma chain over K = n: ~n eps relative to sum |a x|
	for orth in (0, 3):
		got = eng.quad_batch(op, X, 12, orth, fun="log")
		want = oracle.quad_batch(A, X, 12, orth, fun="log", fresh_q=True)
		np.testing.assert_allclose(got, want, rtol=3e-4, err_msg=f"orth={orth}")
	G = rng.standard_normal((n, n)).astype(np.float32)  # not symmetric: Y = G X, not G^T X
	refg = G.astype(np.float64) @ X.astype(np.float64)
	np.testing.assert_allclose(eng.DeviceOperator(G).matmat(X), refg, rtol=1e-5, atol=1e-5 * np.abs(refg).max())
	op.close()


def test_all_three_operator_kinds_agree_on_slq(oracle, eng):
	A = laplacian_2d(12)
	rng = np.random.default_rng(2)
	X = np.asfortranarray(rng.standard_normal((144, 5)))
	ref = oracle.quad_batch(A, X, 15, 3, fun="log", fresh_q=True)

	class PyOp:
		shape, dtype = A.shape, np.dtype(np.float64)

		def matvec(self, v):
			return A @ v

	for M in [A, A.toarray(), PyOp()]:
		got = eng.quad_batch(eng.DeviceOperator(M), X, 15, 3, fun="log")
		np.testing.assert_allclose(got, ref, rtol=1e-10)

	class Broken:
		shape, dtype = A.shape, np.dtype(np.float64)

		def matvec(self, v):
			raise KeyError("boom")

	with pytest.raises(KeyError):
		eng.quad_batch(eng.DeviceOperator(Broken()), X, 5, 0, fun="log")


def test_fun_action_against_dense_eigendecomposition(eng, golden):
	## reference tests/test_operator.py:72-83 style: M @ v == U f(L) U^T v for deg = n
	rng = np.random.default_rng(1234)
	n = 60
	B = rng.standard_normal((n, n))
	A = B @ B.T / n + 0.5 * np.eye(n)
	ew, ev = np.linalg.eigh(A)
	X = np.asfortranarray(rng.uniform(-1, 1, (n, 4)))
	op = eng.DeviceOperator(A)
	plan = eng.LanczosPlan(op, 4, n, n, keep_basis=True)
	for fun, kw, f in [("identity", {}, lambda x: x), ("log", {}, np.log), ("exp", {"t": -0.3}, lambda x: np.exp(-0.3 * x)), ("inv", {}, lambda x: 1 / x), ("sqrt", {}, np.sqrt)]:
		plan.set_probes(X)
		plan.run()
		Y = plan.fun_action(fun, **kw)
		np.testing.assert_allclose(Y, (ev * f(ew)) @ ev.T @ X, rtol=1e-8, atol=1e-9, err_msg=fun)
	## golden (injected): the reference's MatrixFunction._matvec on the Laplacian
	L = laplacian_2d(int(golden["lap_m"]))
	V = golden["lap_probes"][:, :4]
	plan = eng.LanczosPlan(eng.DeviceOperator(L), 4, 20, 20, keep_basis=True)
	plan.set_probes(V)
	plan.run()
	np.testing.assert_allclose(plan.fun_action("exp", t=-0.1), golden["mf_matvec_exp_t"], rtol=1e-9, atol=1e-10)


def test_one_call_fun_action_batch(eng, golden):
	"""slq_fAv_batch: f(A) X for a block of columns in one FFI crossing, against the dense eigendecomposition
	and against the plan-based route it wraps."""
	L = random_spd_graph(90, 5.0, seed=4)  # distinct eigenvalues: no Lanczos breakdown before deg = n
	n = L.shape[0]
	rng = np.random.default_rng(2)
	X = np.asfortranarray(rng.standard_normal((n, 11)))
	w, U = np.linalg.eigh(L.toarray())
	op = eng.DeviceOperator(L)
	for fun, kw, f in [("exp", {"t": -0.2}, lambda x: np.exp(-0.2 * x)), ("inv", {}, lambda x: 1 / x), ("identity", {}, lambda x: x)]:
		Y = eng.fun_action_batch(op, X, deg=n, orth=n, fun=fun, **kw)
		np.testing.assert_allclose(Y, (U * f(w)) @ (U.T @ X), rtol=1e-7, atol=1e-8)
	plan = eng.LanczosPlan(op, 11, 25, 5, keep_basis=True)
	plan.set_probes(X)
	plan.run()
	np.testing.assert_allclose(eng.fun_action_batch(op, X, deg=25, orth=5, fun="exp", t=-0.2), plan.fun_action("exp", t=-0.2), rtol=1e-12, atol=1e-13)
	with pytest.raises(ValueError):
		eng.fun_action_batch(op, X[:5], deg=10)
	## deg > 141: the eigenvector matrices no longer fit in LDS and live in a global scratch
	A2 = random_spd_graph(400, 6.0, seed=8)
	w2, U2 = np.linalg.eigh(A2.toarray())
	X2 = np.asfortranarray(rng.standard_normal((400, 5)))
	Y2 = eng.fun_action_batch(eng.DeviceOperator(A2), X2, deg=200, orth=200, fun="exp", t=-0.3)
	np.testing.assert_allclose(Y2, (U2 * np.exp(-0.3 * w2)) @ (U2.T @ X2), rtol=1e-8, atol=1e-9)


def test_standalone_quadrature_entry(eng, golden, oracle):
	d, e = golden["tri_d"], golden["tri_e"]
	nodes, weights = eng.quadrature_batch(d[None, :], e[None, :])
	np.testing.assert_allclose(nodes[0], golden["tri_nodes"], rtol=0, atol=1e-13)
	np.testing.assert_allclose(weights[0], golden["tri_weights"], rtol=0, atol=1e-13)
	## reference tests/test_tridiagonal.py:26-44: d = 150, eigenvalues within 1e-14 of LAPACK
	from scipy.linalg import eigvalsh_tridiagonal

	D, E = [], []
	for seed in [1234, 4756, 43, 102]:
		rng = np.random.default_rng(seed)
		D.append(rng.uniform(size=150))
		E.append(np.append([0.0], rng.uniform(size=149, low=0.0, high=0.5)))
	nodes, weights = eng.quadrature_batch(np.array(D), np.array(E))
	for i in range(4):
		assert np.max(np.abs(nodes[i] - eigvalsh_tridiagonal(D[i], E[i][1:]))) <= 2e-14
		np.testing.assert_allclose(weights[i].sum(), 1.0, atol=1e-13)


def test_single_vector_dropin_entry(oracle, eng, golden):
	"""slq_lanczos_f64: same in/out contract as primate._lanczos.lanczos (_lanczos.cpp:88-99)."""
	from primate_amd.lanczos import _native_lanczos, lanczos

	L = laplacian_2d(int(golden["lap_m"]))
	v = golden["lap_probes"][:, 0]
	for orth, ncv in [(0, 2), (3, 3), (3, 20), (20, 20), (5, 7)]:
		al, be, Q = np.zeros(21), np.zeros(21), np.zeros((L.shape[0], ncv), order="F")
		al2, be2, Q2 = np.zeros(21), np.zeros(21), np.zeros((L.shape[0], ncv), order="F")
		steps = _native_lanczos(L, v, 20, 1e-8, orth, al, be, Q)
		steps2 = oracle.lanczos(L, v, 20, 1e-8, orth, al2, be2, Q2)
		assert steps == steps2 == 20
		np.testing.assert_allclose(al, al2, rtol=1e-9, atol=1e-12)
		np.testing.assert_allclose(be, be2, rtol=1e-9, atol=1e-12)
		np.testing.assert_allclose(Q, Q2, rtol=0, atol=1e-9)  # same ring columns, same content
	## the public API, against vectors captured from the reference's lanczos() (injected)
	for orth, rb in [(0, False), (5, False), (-1, False), (3, True)]:
		res = lanczos(L, v0=v.copy(), deg=20, orth=orth, return_basis=rb)
		(a, b) = res[0] if rb else res
		np.testing.assert_allclose(a, golden[f"api_alpha_o{orth}_rb{int(rb)}"], rtol=1e-9)
		np.testing.assert_allclose(b, golden[f"api_beta_o{orth}_rb{int(rb)}"], rtol=1e-9)
		if rb:
			np.testing.assert_allclose(res[1], golden[f"api_Q_o{orth}_rb1"], atol=1e-9)


def test_device_probe_generator(eng):
	## reference tests/test_random.py:6-20 properties, on the Philox generator
	A = laplacian_2d(40)  # n = 1600
	op = eng.DeviceOperator(A)
	n = A.shape[0]
	plan = eng.LanczosPlan(op, 200, 4, 0)
	plan.generate_probes("rademacher", seed=7)
	R = plan.get_probes()
	assert set(np.unique(R)) == {-1.0, 1.0}  # exact support
	assert abs(R.mean()) < 5 / np.sqrt(R.size) and np.max(np.abs(R.T @ R / n - np.eye(200))) < 0.2
	plan.generate_probes("normal", seed=7)
	Gs = plan.get_probes()
	from scipy.stats import normaltest

	assert normaltest(Gs.ravel()).pvalue >= 0.001 and abs(Gs.std() - 1) < 0.01
	plan.generate_probes("sphere", seed=7)
	S = plan.get_probes()
	plan.run()
	q = plan.quadrature("identity")
	## sphere probes: direction g/||g||, norm^2 = n exactly (random.py:36-41)
	Sn = S / np.linalg.norm(S, axis=0) * np.sqrt(n)
	np.testing.assert_allclose(q, np.einsum("ij,ij->j", Sn, A @ Sn), rtol=1e-10)
	## same (seed, id) -> same probe, whatever the batch shape or offset
	p1 = eng.LanczosPlan(op, 200, 4, 0)
	p1.generate_probes("rademacher", seed=7)
	p2 = eng.LanczosPlan(op, 50, 4, 0)
	p2.generate_probes("rademacher", seed=7, probe_offset=100)
	assert np.array_equal(p1.get_probes()[:, 100:150], p2.get_probes())
	p2.generate_probes("rademacher", seed=8, probe_offset=100)
	assert not np.array_equal(p1.get_probes()[:, 100:150], p2.get_probes())


def test_full_size_properties_c2(eng, oracle):
	"""BASELINE configs[1] at full size, 256 device probes in the production geometry (2 panels of 128): the
	oracle on four of the columns (first and last of each panel), then size-independent checks."""
	L2 = laplacian_2d(1000)
	op = eng.DeviceOperator(L2)
	n = L2.shape[0]
	logdet = 1166809.9080624094  # closed form, BASELINE.md §2
	plan = eng.LanczosPlan(op, 256, 30, 3)
	assert plan.describe()["tiles"] == 2  # the default path of this operator: ring-fed LDS tiles
	plan.generate_probes("rademacher", seed=1234)
	plan.run()
	q = plan.quadrature("log")
	assert abs(q.mean() - logdet) < 6 * q.std(ddof=1) / np.sqrt(256)  # unbiased within 6 sigma
	assert abs(q.mean() / logdet - 1) < 2e-3
	## identity: sum f(theta) tau ||v||^2 with f = 1 is ||v||^2 = n exactly for Rademacher probes
	plan.generate_probes("rademacher", seed=1234)
	plan.run()
	qi, nodes, weights = plan.quadrature("identity", return_rule=True)
	np.testing.assert_allclose(weights.sum(axis=1), 1.0, atol=1e-12)
	assert nodes.min() > 0 and nodes.max() < 8  # spectrum of the Dirichlet Laplacian
	a, b, steps = plan.tridiag()
	assert np.all(steps == 30)
	## trace(A) estimate is exact for Rademacher probes on a matrix with constant diagonal 4:
	## v^T A v = 4n - 2 * (number of +1 neighbours minus -1 ...) varies; check first moment
	np.testing.assert_allclose(a[:, 0] * n, qi, rtol=1e-12)  # alpha_0 = v^T A v / ||v||^2
	## sharding independence: two half-batches with offsets reproduce the batch (the
	## probes are bitwise the same; the block partition of the reductions depends on the panel count,
	## so the values agree to rounding, not bitwise)
	plan.generate_probes("rademacher", seed=1234)
	V = plan.get_probes()  # the device-drawn probes, n x 256
	plan.run()
	q_all = plan.quadrature("log")
	cols = [0, 127, 128, 255]
	ref = oracle.quad_batch(L2, np.asfortranarray(V[:, cols]), 30, 3, fun="log", fresh_q=True)
	np.testing.assert_allclose(q_all[cols], ref, rtol=1e-10)
	del V
	half = eng.LanczosPlan(op, 128, 30, 3)
	parts = []
	for off in (0, 128):
		half.generate_probes("rademacher", seed=1234, probe_offset=off)
		half.run()
		parts.append(half.quadrature("log"))
	np.testing.assert_allclose(np.concatenate(parts), q_all, rtol=1e-13)
	half.generate_probes("rademacher", seed=1234, probe_offset=128)
	half.run()
	assert np.array_equal(half.quadrature("log"), parts[1])  # same configuration: bitwise reproducible
	## orth = 0 / 3 / 30 agree far below the 1e-6 bar on the same probes
	p0 = eng.LanczosPlan(op, 16, 30, 0)
	p0.generate_probes("rademacher", seed=1234)
	p0.run()
	np.testing.assert_allclose(p0.quadrature("log"), q_all[:16], rtol=1e-10)
	## deeper windows (4..6 ring columns per step: the 8-wave ring-fed form) and the narrow panels the reference's drivers
	## submit (hutch's batch of 32, an 8-GPU shard of 64; merged tiles), at full size: the oracle on two columns each
	V = None
	for P, orth in ((256, 6), (64, 3), (64, 6), (32, 3)):
		pl = eng.LanczosPlan(op, P, 30, orth)
		assert pl.describe()["tiles"] == 2, (P, orth)
		pl.generate_probes("rademacher", seed=1234)
		if V is None:
			V = pl.get_probes()
		pl.run()
		qd = pl.quadrature("log")
		pl.close()
		cc = [0, P - 1]
		ref = oracle.quad_batch(L2, np.asfortranarray(V[:, cc]), 30, orth, fun="log", fresh_q=True)
		np.testing.assert_allclose(qd[cc], ref, rtol=1e-10, err_msg=f"P={P} orth={orth}")


def test_full_size_north_star_operator_on_the_default_path(eng, oracle):
	"""The north_star's operator at full size (3-D 7-point Laplacian 100^3, nnz = 6.94 M, 256 probes, k = 30) as a user runs
	it - tiles, second-level row order and all: the oracle on the first and last column of each panel at orth 3 and 0, the
	generic passes (SLQ_TILES=0 is a different stored order and different kernels) to rounding, run-to-run bitwise."""
	A = laplacian_3d(100)
	op = eng.DeviceOperator(A)
	cols = [0, 127, 128, 255]
	res = {}
	for orth in (3, 0):
		plan = eng.LanczosPlan(op, 256, 30, orth)
		assert plan.describe()["tiles"] == 2
		plan.generate_probes("rademacher", seed=77)
		V = plan.get_probes()[:, cols]
		plan.run()
		q = plan.quadrature("log")
		ref = oracle.quad_batch(A, np.asfortranarray(V), 30, orth, fun="log", fresh_q=True)
		np.testing.assert_allclose(q[cols], ref, rtol=1e-10)
		plan.generate_probes("rademacher", seed=77)
		plan.run()
		assert np.array_equal(plan.quadrature("log"), q)
		res[orth] = q
		plan.close()
	np.testing.assert_allclose(res[0], res[3], rtol=1e-8)  # reorthogonalisation moves a 30-step rule of this operator far below 1e-6
	## the narrow panels the reference's drivers submit (an 8-GPU shard of 512 / 256 probes: 64 / 32 per GPU; hutch's batch of 32,
	## src/primate/trace.py:36) and a six-column window, at full size on this operator: merged tiles, the 8-wave form; the oracle
	## on the first and last column of each
	V = None
	for P, orth in ((64, 3), (64, 6), (32, 3)):
		pl = eng.LanczosPlan(op, P, 30, orth)
		info = pl.describe()
		assert info["tiles"] == 2 and info["sequence"] == "fused_gram", (P, orth, info)
		pl.generate_probes("rademacher", seed=77)
		if V is None:
			V = pl.get_probes()
		pl.run()
		qd = pl.quadrature("log")
		pl.close()
		cc = [0, P - 1]
		ref = oracle.quad_batch(A, np.asfortranarray(V[:, cc]), 30, orth, fun="log", fresh_q=True)
		np.testing.assert_allclose(qd[cc], ref, rtol=1e-10, err_msg=f"P={P} orth={orth}")
	## ... and the north_star's own wording of the inner loop, "with full reorthogonalization" (orth = k: lanczos.h:133-136), at full
	## size: the Gram sequence up to 8 columns, then the block-CGS sweeps whose dots chunks are read-only (r04: the three-term
	## step's axpy is applied by the update sweep; SLQ_DEFER_AXPY=0 is the stored form - bitwise the same numbers)
	pl = eng.LanczosPlan(op, 64, 30, 30)
	pl.generate_probes("rademacher", seed=77)
	pl.run()
	qf = pl.quadrature("log")
	pl.close()
	ref = oracle.quad_batch(A, np.asfortranarray(V[:, [0, 63]]), 30, 30, fun="log", fresh_q=True)
	np.testing.assert_allclose(qf[[0, 63]], ref, rtol=1e-10)
	os.environ["SLQ_DEFER_AXPY"] = "0"
	try:
		pl = eng.LanczosPlan(op, 64, 30, 30)
		pl.generate_probes("rademacher", seed=77)
		pl.run()
		assert np.array_equal(pl.quadrature("log"), qf)
		pl.close()
	finally:
		del os.environ["SLQ_DEFER_AXPY"]
	del V
	op.close()
	os.environ["SLQ_TILES"] = "0"
	try:
		op0 = eng.DeviceOperator(A)
		plan = eng.LanczosPlan(op0, 256, 30, 3)
		assert plan.describe()["tiles"] == 0
		plan.generate_probes("rademacher", seed=77)
		plan.run()
		np.testing.assert_allclose(plan.quadrature("log"), res[3], rtol=1e-11)
		plan.close()
		op0.close()
	finally:
		del os.environ["SLQ_TILES"]


def test_3d_laplacian_north_star_variant(oracle, eng):
	## the north_star's "nnz~7M" operator at reduced size (3D 7-point), oracle parity incl. full reorth
	A = laplacian_3d(14)
	rng = np.random.default_rng(3)
	X = np.asfortranarray(np.floor(rng.random((A.shape[0], 9)) * 2) * 2 - 1)
	op = eng.DeviceOperator(A)
	for orth in [0, 3, 30]:
		got = eng.quad_batch(op, X, 30, orth, fun="log")
		ref = oracle.quad_batch(A, X, 30, orth, fun="log", fresh_q=True)
		np.testing.assert_allclose(got, ref, rtol=1e-10)


def test_abi_error_behaviour(eng):
	from primate_amd import _capi

	A = laplacian_2d(6)
	op = eng.DeviceOperator(A)
	with pytest.raises(ValueError):
		eng.LanczosPlan(op, 0, 5, 0)
	with pytest.raises(ValueError):
		eng.LanczosPlan(op, 4, 0, 0)
	plan = eng.LanczosPlan(op, 4, 5, 0)
	with pytest.raises(ValueError):
		plan.run()  # no probes yet
	with pytest.raises(ValueError):
		plan.tridiag()
	with pytest.raises(ValueError):
		plan.basis(0)  # no keep_basis
	with pytest.raises(AssertionError):
		eng.DeviceOperator(sp.random(5, 6, 0.5).tocsr())
	with pytest.raises(AssertionError):
		eng.DeviceOperator(A.astype(np.int32))
	with pytest.raises(ValueError):
		eng.DeviceOperator(type("NoMatvec", (), {"shape": (3, 3), "dtype": np.dtype(np.float64)})())
	## malformed CSR is rejected by the library itself
	import ctypes as C

	rowptr = np.array([0, 2, 1], dtype=np.int32)
	colind = np.array([0, 5], dtype=np.int32)
	vals = np.ones(2)
	h = C.c_void_p()
	rc = _capi.lib().slq_csr_create(op.ctx._h, _capi.SLQ_F64, 2, 2, _capi.ptr(rowptr), _capi.ptr(colind), _capi.ptr(vals), C.byref(h))
	assert rc == _capi.SLQ_EINVAL and b"rowptr" in _capi.lib().slq_last_error()


def test_device_resident_probes_and_bandwidth_probe(eng, lap, golden):
	"""slq_plan_set_probes_device: probes handed over as a device pointer (here a torch tensor's) give
	bitwise the results of the host-pointer path; slq_measure_stream reports a sane HBM rate."""
	import torch

	L, op, V = lap
	plan = eng.LanczosPlan(op, V.shape[1], 20, 3)
	plan.set_probes(V)
	plan.run()
	q_host = plan.quadrature("log")
	t = torch.from_numpy(np.ascontiguousarray(V.T)).cuda()  # (nprobes, n) C-order == column-major n x nprobes
	torch.cuda.synchronize()
	plan.set_probes_device(t.data_ptr())
	plan.run()
	assert np.array_equal(plan.quadrature("log"), q_host)
	bw = op.ctx.measure_stream("triad", nbytes=1 << 28, reps=3)
	assert 1000.0 < bw < 8000.0


def test_opt_in_row_reordering_is_transparent(oracle, eng, monkeypatch):
	"""SLQ_REORDER=2 stores P A P^T (RCM inside each XCD chunk): every host-visible array (probes,
	basis, f(A)v, diag) must come back in the caller's row order and the values agree to rounding."""
	A = random_spd_graph(777, 5.0, seed=21)
	rng = np.random.default_rng(4)
	X = np.asfortranarray(rng.standard_normal((777, 9)))
	ref = oracle.quad_batch(A, X, 20, 3, fun="log", fresh_q=True, prefer="csr")
	monkeypatch.setenv("SLQ_REORDER", "2")
	op = eng.DeviceOperator(A)
	monkeypatch.delenv("SLQ_REORDER")
	np.testing.assert_allclose(eng.quad_batch(op, X, 20, 3, fun="log"), ref, rtol=1e-10)
	np.testing.assert_allclose(op.matmat(X), A @ X, rtol=1e-12, atol=1e-12)
	plan = eng.LanczosPlan(op, 9, 20, 20, keep_basis=True)
	plan.set_probes(X)
	assert np.array_equal(plan.get_probes(), X)
	plan.run()
	Q = plan.basis(2)
	np.testing.assert_allclose(Q[:, 0], X[:, 2] / np.linalg.norm(X[:, 2]), rtol=1e-13)
	np.testing.assert_allclose(Q.T @ Q, np.eye(20), atol=1e-10)
	np.testing.assert_allclose(plan.fun_action("identity"), A @ X, rtol=1e-9, atol=1e-9)
	acc = eng.DiagAccumulator(777, ctx=op.ctx)
	acc.update(plan, "identity")
	numer, denom, _, cnt = acc.get()
	np.testing.assert_allclose(numer, np.sum((A @ X) * X, axis=1), rtol=1e-9, atol=1e-9)
	np.testing.assert_allclose(denom, np.sum(X * X, axis=1), rtol=1e-12)
	## device-drawn probes are a function of (seed, probe id, CALLER row): the same with and without reordering
	op0 = eng.DeviceOperator(A)
	plan0 = eng.LanczosPlan(op0, 9, 20, 3)
	plan3 = eng.LanczosPlan(op, 9, 20, 3)
	for pdf in ("rademacher", "normal", "sphere"):
		plan0.generate_probes(pdf, seed=77, probe_offset=5)
		plan3.generate_probes(pdf, seed=77, probe_offset=5)
		assert np.array_equal(plan0.get_probes(), plan3.get_probes()), pdf
		plan0.run()
		plan3.run()
		np.testing.assert_allclose(plan3.quadrature("log"), plan0.quadrature("log"), rtol=1e-10)


def test_alpha_pass_upper_triangle_only_for_exactly_symmetric_csr(oracle, eng, monkeypatch):
	"""The alpha pass gathers only the upper triangle when the stored CSR is exactly symmetric, and keeps
	full rows otherwise: (a) symmetric operator, every launch sequence (merged alpha+dots pass; separate alpha
	pass with the upper triangle / cross term switched on or off; store-and-revisit sweeps) agrees with the oracle; (b) an operator that is NOT symmetric goes through the same
	arithmetic as the oracle's recurrence (lanczos.h never checks symmetry), so alpha/beta still match."""
	A = random_spd_graph(1501, 5.0, seed=3)
	rng = np.random.default_rng(8)
	X = np.asfortranarray(rng.standard_normal((1501, 20)))
	for orth in (0, 3):
		ref = oracle.quad_batch(A, X, 18, orth, fun="log", fresh_q=True, prefer="csr")
		for env in ({}, {"SLQ_SYM_ALPHA": "0"}, {"SLQ_CROSS": "0"}, {"SLQ_SYM_ALPHA": "0", "SLQ_CROSS": "0"}, {"SLQ_MERGED": "0"},
		            {"SLQ_MERGED": "0", "SLQ_CROSS": "0"}, {"SLQ_FUSED": "0"}):  # fmt: skip
			for k, v in env.items():
				monkeypatch.setenv(k, v)
			op = eng.DeviceOperator(A)
			np.testing.assert_allclose(eng.quad_batch(op, X, 18, orth, fun="log"), ref, rtol=1e-10, err_msg=f"{env} orth={orth}")
			op.close()
			for k in env:
				monkeypatch.delenv(k)
	## (b) perturb one off-diagonal entry: no longer symmetric
	B = A.copy().tolil()
	i, j = np.transpose(sp.triu(A, 1).nonzero())[7]
	B[i, j] = B[i, j] * 1.5
	B = B.tocsr()
	B.sort_indices()
	op = eng.DeviceOperator(B)
	for orth in (0, 3):
		plan = eng.LanczosPlan(op, 20, 12, orth)
		## (the Gram sequence moves A across an inner product - W_t . (A W_j) = (A W_t) . W_j - which only an exactly symmetric operator allows)
		assert plan.describe()["sequence"] == "fused" and plan.describe()["upper_alpha"] == 0, plan.describe()
		plan.set_probes(X)
		plan.run()
		a, b, _ = plan.tridiag()
		for c in (0, 7, 19):
			ar, br, Qr = np.zeros(13), np.zeros(13), np.zeros((1501, max(orth, 2)), order="F")
			oracle.lanczos(B, X[:, c].copy(), 12, 1e-8, orth, ar, br, Qr)
			np.testing.assert_allclose(a[c][:12], ar[:12], rtol=1e-9, atol=1e-9)
			np.testing.assert_allclose(b[c][1:12], br[1:12], rtol=1e-9, atol=1e-9)


def test_non_symmetric_operator_on_tiles_keeps_the_direct_projections(oracle, eng, monkeypatch):
	"""A tiled operator that is NOT exactly symmetric (one off-diagonal entry of a 5-point grid scaled): the ring-fed passes run the merged
	alpha+dots sequence - the Gram sequence needs A = A^T (r03 took it regardless: fixed in r04) - and alpha / beta follow the oracle's
	recurrence, which never checks symmetry (lanczos.h:127-136), on wide and narrow panels."""
	monkeypatch.setenv("SLQ_TILES", "2")
	B = laplacian_2d(100).tolil()
	B[4321, 4322] = -1.5
	B = B.tocsr()
	B.sort_indices()
	n = B.shape[0]
	rng = np.random.default_rng(5)
	op = eng.DeviceOperator(B)
	for P in (130, 40):
		X = np.asfortranarray(rng.standard_normal((n, P)))
		for orth in (3, 6):
			plan = eng.LanczosPlan(op, P, 14, orth)
			info = plan.describe()
			assert info["tiles"] == 2 and info["sequence"] == "fused" and info["upper_alpha"] == 0, info
			plan.set_probes(X)
			plan.run()
			a, b, _ = plan.tridiag()
			plan.close()
			for c in (0, P - 1):
				ar, br, Qr = np.zeros(15), np.zeros(15), np.zeros((n, orth), order="F")
				oracle.lanczos(B, X[:, c].copy(), 14, 1e-8, orth, ar, br, Qr)
				np.testing.assert_allclose(a[c][:14], ar[:14], rtol=1e-10, atol=1e-10)
				np.testing.assert_allclose(b[c][1:14], br[1:14], rtol=1e-10, atol=1e-10)
	op.close()


def test_non_local_operator_uses_stored_u_passes(oracle, eng, monkeypatch):
	"""A random graph whose neighbours lie all over the vector (> 4 gathers per row beyond 4096 rows): the merged
	pass stores u and the update pass reads it back (one gather pass per step); with SLQ_STORED_U=0 the
	store-and-revisit sweeps run instead. Both against the oracle, orth 1..4 and 0."""
	A = random_spd_graph(20000, 12.0, seed=2)
	rng = np.random.default_rng(6)
	X = np.asfortranarray(rng.standard_normal((A.shape[0], 40)))
	for orth in (0, 1, 2, 3, 4):
		ref = oracle.quad_batch(A, X, 14, orth, fun="exp", t=-0.05, fresh_q=True, prefer="csr")
		for env in ({}, {"SLQ_STORED_U": "0"}):
			for k, v in env.items():
				monkeypatch.setenv(k, v)
			op = eng.DeviceOperator(A)
			np.testing.assert_allclose(eng.quad_batch(op, X, 14, orth, fun="exp", t=-0.05), ref, rtol=1e-10, err_msg=f"{env} orth={orth}")
			op.close()
			for k in env:
				monkeypatch.delenv(k)


@pytest.mark.parametrize("dtype,rtol", [(np.float64, 1e-10), (np.float32, 3e-4)])
def test_pipelined_row_loop_with_empty_rows(oracle, eng, monkeypatch, dtype, rtol):
	"""The pipelined row loop of the dots/update passes (k_csr_pass<PIPE=1>: wide panels of operators with more than 5.5
	nonzeros per row) on a matrix with rows that store NOTHING - isolated nodes of an adjacency-like operator, masked rows.
	The prefetch state (row pointers two rows ahead, the next row's first indices) advances once per row whatever the
	row's length; an empty row must not stall it for the rows that follow in the same wave. eigen_operators.h:66-77 (the
	reference's sparse product has no special case for such rows either)."""
	rng = np.random.default_rng(12)
	n = 9000
	W = sp.random(n, n, density=9.0 / n, random_state=7, format="coo")
	W = (W + W.T).tolil()
	dead = np.concatenate([rng.choice(n, 40, replace=False), [0, 1, 2, n - 1, 4096, 4097]])  # scattered, adjacent, first and last rows
	W[dead, :] = 0
	W[:, dead] = 0
	A = (sp.diags(np.where(np.isin(np.arange(n), dead), 0.0, 20.0)) + W.tocsr()).tocsr().astype(dtype)
	A.eliminate_zeros()
	A.sort_indices()
	assert np.all(np.diff(A.indptr)[dead] == 0) and A.nnz / n > 5.5
	P = 130 if dtype == np.float64 else 260
	X = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1).astype(dtype)
	cols = [0, 64, P - 1]
	monkeypatch.setenv("SLQ_PIPE", "1")
	monkeypatch.setenv("SLQ_TILES", "0")
	monkeypatch.setenv("SLQ_FUSED", "2")  # the recompute passes whatever the gather distances
	op = eng.DeviceOperator(A)
	## both sequences of the generic passes: merged alpha+dots + update (SLQ_GRAM_CSR=0) and, the default since r04, alpha + update on Gram rows
	for gram_csr, seq in (("0", "fused"), ("1", "fused_gram")):
		monkeypatch.setenv("SLQ_GRAM_CSR", gram_csr)
		plan = eng.LanczosPlan(op, P, 12, 3)
		assert plan.describe()["pipelined"] == 1 and plan.describe()["sequence"] == seq, plan.describe()
		plan.close()
		for orth in (0, 3, 6):
			ref = oracle.quad_batch(A, np.asfortranarray(X[:, cols]), 12, orth, fun="exp", t=-0.01, fresh_q=True, prefer="csr")
			got = eng.quad_batch(op, X, 12, orth, fun="exp", t=-0.01)[cols]
			np.testing.assert_allclose(got, ref, rtol=rtol, err_msg=f"orth={orth} {seq}")
	op.close()


def test_update_sweep_skips_columns_nobody_projects_on(oracle, eng, monkeypatch):
	"""Deep windows (orth > 8: the store-and-revisit sweeps). The reference skips a projection per probe when it is below 2 eps sqrt(n) (lanczos.h:53,62); the update
	sweep does not even READ a ring column whose coefficient is zero for every probe of its panel (r04). `w -= 0 * x` is w, so the results are bitwise those of the
	sweep that reads everything (SLQ_SWEEP_SKIP=0); on a well-conditioned grid nearly every column of a 20-column window is skipped, on the reference's ill-conditioned
	full-reorth test matrix (tests/test_lanczos.py:11-20 shape: eigenvalues over seven decades) hardly any - and both follow the oracle."""
	rng = np.random.default_rng(13)
	A = laplacian_2d(90)
	n = A.shape[0]
	X = np.asfortranarray(np.floor(rng.random((n, 40)) * 2) * 2 - 1)
	op = eng.DeviceOperator(A)
	res = {}
	for skip in ("1", "0"):
		monkeypatch.setenv("SLQ_SWEEP_SKIP", skip)
		plan = eng.LanczosPlan(op, 40, 24, 24)
		plan.set_probes(X)
		plan.run()
		res[skip] = (plan.quadrature("log"), plan.tridiag(), plan.sweep_columns())
		plan.close()
	monkeypatch.delenv("SLQ_SWEEP_SKIP")
	assert np.array_equal(res["1"][0], res["0"][0]) and all(np.array_equal(a, b) for a, b in zip(res["1"][1], res["0"][1]))
	rd1, off1 = res["1"][2]
	rd0, off0 = res["0"][2]
	assert off1 == off0 > 0 and rd0 == off0 and rd1 < 0.5 * off1, (rd1, off1, rd0, off0)
	np.testing.assert_allclose(res["1"][0][[0, 39]], oracle.quad_batch(A, np.asfortranarray(X[:, [0, 39]]), 24, 24, fun="log", fresh_q=True), rtol=1e-10)
	## the same with the fp32 archive of finished vectors (SLQ_RING32=1, k_reorth_update32): skipping zero columns is bitwise neutral there too
	monkeypatch.setenv("SLQ_RING32", "1")
	r32 = {}
	for skip in ("1", "0"):
		monkeypatch.setenv("SLQ_SWEEP_SKIP", skip)
		plan = eng.LanczosPlan(op, 40, 24, 24)
		assert plan.describe()["sequence"] == "sweeps_ring32"
		plan.set_probes(X)
		plan.run()
		r32[skip] = (plan.quadrature("log"), plan.tridiag(), plan.sweep_columns())
		plan.close()
	monkeypatch.delenv("SLQ_SWEEP_SKIP")
	monkeypatch.delenv("SLQ_RING32")
	assert np.array_equal(r32["1"][0], r32["0"][0]) and all(np.array_equal(a, b) for a, b in zip(r32["1"][1], r32["0"][1]))
	## (hardly anything IS skipped there: against vectors rounded to fp32 the projections are 1e-8, not below 2 eps sqrt(n) - measured 251 of 300 columns read)
	assert r32["1"][2][0] <= r32["1"][2][1] and r32["0"][2][0] == r32["0"][2][1], (r32["1"][2], r32["0"][2])
	np.testing.assert_allclose(r32["1"][0], res["1"][0], rtol=1e-7)
	op.close()
	## an operator on which the window's projections are NOT small: dense SPD with eigenvalues over seven decades, full reorthogonalisation
	m = 300
	Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
	D = (Q * np.logspace(-3, 4, m)) @ Q.T
	D = np.asfortranarray((D + D.T) / 2)
	Xd = np.asfortranarray(rng.standard_normal((m, 20)))
	opd = eng.DeviceOperator(D)
	plan = eng.LanczosPlan(opd, 20, 40, 40)
	plan.set_probes(Xd)
	plan.run()
	got = plan.quadrature("log")
	rd, off = plan.sweep_columns()
	plan.close()
	opd.close()
	assert rd > 0.5 * off, (rd, off)
	np.testing.assert_allclose(got, oracle.quad_batch(D, Xd, 40, 40, fun="log", fresh_q=True), rtol=1e-8)


def test_opt_in_fp32_archive_ring(oracle, eng, monkeypatch):
	"""SLQ_RING32=1 (opt-in): finished Lanczos vectors archived as fp32, reorthogonalisation columns j-2 and older read
	from the archive. Not bit-compatible by construction; the bar here is 1e-7 relative per probe against the oracle
	(the north_star's is 1e-6; measured worst case 2e-9, profiles/r02_ring32_eval.json). Only plans with reorthogonalisation
	deeper than 8 columns and no kept basis take the path."""
	A = random_spd_graph(3000, 6.0, seed=21)
	rng = np.random.default_rng(4)
	X = np.asfortranarray(rng.standard_normal((3000, 70)))
	op = eng.DeviceOperator(A)
	monkeypatch.setenv("SLQ_RING32", "1")
	for deg, orth in ((40, 12), (40, 40)):
		plan = eng.LanczosPlan(op, 70, deg, orth)
		assert plan.describe()["sequence"] == "sweeps_ring32" and plan.describe()["ring_slots"] == 3
		plan.set_probes(X)
		plan.run()
		got = plan.quadrature("log")
		a, b, steps = plan.tridiag()
		assert np.all(steps == deg)
		plan.close()
		ref = oracle.quad_batch(A, X, deg, orth, fun="log", fresh_q=True)
		np.testing.assert_allclose(got, ref, rtol=1e-7)
	assert eng.LanczosPlan(op, 70, 40, 3).describe()["sequence"] != "sweeps_ring32"  # shallow reorthogonalisation: untouched
	assert eng.LanczosPlan(op, 8, 40, 40, keep_basis=True).describe()["sequence"] != "sweeps_ring32"  # kept basis: untouched
	monkeypatch.delenv("SLQ_RING32")
	assert eng.LanczosPlan(op, 70, 40, 40).describe()["sequence"] != "sweeps_ring32"


@pytest.mark.parametrize("variant", ["1", "2"])
def test_opt_in_lds_row_tiles_match_generic_passes(oracle, eng, monkeypatch, variant):
	"""SLQ_TILES=1: rows regrouped into compact clusters, every cluster one workgroup tile whose distinct panel rows are
	staged once in LDS (k_csr_tile_pass); SLQ_TILES=2: the same tiles fed through a ring of LDS images by loader waves
	(k_csr_ring_pass, ring-column counts up to 3; the rest take k_csr_tile_pass on the same tiles). Same per-probe values
	as the oracle for every ring-column count of the fused steps, both dtypes, 2-D and 3-D grids (with the XCD reordering
	on top), tile heights 24 / 16 / 7, a matrix with empty rows; operators whose tiles would share nothing keep the
	generic passes."""
	rng = np.random.default_rng(9)
	monkeypatch.setenv("SLQ_TILES", variant)
	if variant == "2":
		## enough tiles per workgroup (13 and 21) that every slot of the ring is reused several times
		for A, P, tol in ((laplacian_2d(200), 130, 1e-10), (laplacian_3d(40), 130, 1e-10), (laplacian_3d(40).astype(np.float32), 300, 3e-4)):
			n = A.shape[0]
			X = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1).astype(A.dtype)
			cols = [0, 64, P - 1]
			op = eng.DeviceOperator(A)
			plan = eng.LanczosPlan(op, P, 12, 3)
			assert plan.describe()["tiles"] == 2
			plan.close()
			## 5: steps with more than 3 ring columns take the generic passes on the tiles' row order; 12: above 8 the store-and-revisit
			## sweeps, whose SpMM + three-term sweep is the ring kernel's PASS_SPMM on this operator
			for o in (0, 3, 5, 12):
				ref = oracle.quad_batch(A, np.asfortranarray(X[:, cols]), 12, o, fun="log", fresh_q=True)
				np.testing.assert_allclose(eng.quad_batch(op, X, 12, o, fun="log")[cols], ref, rtol=tol, err_msg=f"ring n={n} {A.dtype} orth={o}")
			monkeypatch.setenv("SLQ_MGS", "1")  # exact MGS order: sweeps from the first step on
			ref = oracle.quad_batch(A, np.asfortranarray(X[:, cols]), 12, 3, fun="log", fresh_q=True)
			np.testing.assert_allclose(eng.quad_batch(op, X, 12, 3, fun="log")[cols], ref, rtol=tol, err_msg=f"ring n={n} {A.dtype} MGS order")
			monkeypatch.delenv("SLQ_MGS")
			op.close()
	cases = [(laplacian_2d(70), "24", "0"), (laplacian_3d(17), "16", "2"), (laplacian_2d(66), "7", "0")]
	for A, tr, reorder in cases:
		monkeypatch.setenv("SLQ_TILE_ROWS", tr)
		monkeypatch.setenv("SLQ_REORDER", reorder)
		n = A.shape[0]
		X = np.asfortranarray(np.floor(rng.random((n, 130)) * 2) * 2 - 1)  # wide panel (P > 64), 2 panels
		cols = [0, 1, 63, 64, 127, 129]
		op = eng.DeviceOperator(A)
		plan = eng.LanczosPlan(op, 130, 14, 3)
		assert plan.describe()["reordered"] == 1  # the clusters are a row order of their own
		assert plan.describe()["tiles"] == int(variant)
		plan.close()
		for o in (0, 1, 2, 3, 5, 8):
			ref = oracle.quad_batch(A, np.asfortranarray(X[:, cols]), 14, o, fun="log", fresh_q=True)
			np.testing.assert_allclose(eng.quad_batch(op, X, 14, o, fun="log")[cols], ref, rtol=1e-10, err_msg=f"n={n} orth={o}")
		op.close()
	A32 = laplacian_2d(70).astype(np.float32)
	X32 = np.asfortranarray(np.floor(rng.random((A32.shape[0], 260)) * 2) * 2 - 1).astype(np.float32)
	op = eng.DeviceOperator(A32)
	ref = oracle.quad_batch(A32, np.asfortranarray(X32[:, :4]), 14, 3, fun="log", fresh_q=True)
	np.testing.assert_allclose(eng.quad_batch(op, X32, 14, 3, fun="log")[:4], ref, rtol=3e-4)
	op.close()
	## irregular graph with empty rows: tiles are either feasible and correct, or declined
	G = random_spd_graph(6000, 3.0, seed=5)
	Xg = np.asfortranarray(rng.standard_normal((6000, 70)))
	op = eng.DeviceOperator(G)
	np.testing.assert_allclose(eng.quad_batch(op, Xg, 12, 3, fun="log")[:5], oracle.quad_batch(G, np.asfortranarray(Xg[:, :5]), 12, 3, fun="log", fresh_q=True), rtol=1e-10)
	op.close()


@pytest.mark.parametrize("case", ["lap2d_f64", "lap3d_f64", "lap3d_f32", "ragged_f64"])
def test_ring_fed_passes_every_panel_width_and_ring_depth(oracle, eng, monkeypatch, case):
	"""k_ring_pass (slq_ring.hpp): the ring-fed tile passes for the shapes the reference's drivers actually submit - panels
	of 16 and 32 lanes per row (hutch's batches of 32 probes, src/primate/trace.py:36,104-116; an 8-GPU shard of 256 probes)
	on tiles of 4 / 2 merged base tiles, and steps with 4..8 ring columns (any orth is legal, src/primate/include/lanczos.h:
	58-65) on the 8-wave form. Per-probe values against the oracle on identical probes for P in {20, 40, 64} (+ a wide panel
	through the same kernel, SLQ_RING_GEN=1) and orth in {0, 3, 4, 6, 8, 12}; every plan must actually be on the tiles."""
	rng = np.random.default_rng(31)
	monkeypatch.setenv("SLQ_TILES", "2")
	if case == "ragged_f64":
		## a band with random gaps and a few empty off-diagonals: ragged tiles, short last merged tile per chunk
		n = 30011
		offs = [1, 2, 150]
		D = [rng.uniform(0.2, 1.0, n - o) * (rng.random(n - o) > 0.15) for o in offs]
		B = sp.diags(D, offs, shape=(n, n))
		A = (B + B.T + sp.diags(np.full(n, 8.0))).tocsr()
		A.eliminate_zeros()
		A.sort_indices()
		tol, shapes = 1e-10, (20, 64)
	else:
		A = {"lap2d_f64": laplacian_2d(200), "lap3d_f64": laplacian_3d(40), "lap3d_f32": laplacian_3d(40).astype(np.float32)}[case]
		tol = 3e-4 if A.dtype == np.float32 else 1e-10
		shapes = (40, 100, 128) if A.dtype == np.float32 else (20, 40, 64)
	n = A.shape[0]
	op = eng.DeviceOperator(A)
	for P in shapes:
		X = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1).astype(A.dtype)
		cols = [0, P // 2, P - 1]
		plan = eng.LanczosPlan(op, P, 14, 3)
		info = plan.describe()
		plan.close()
		assert info["tiles"] == 2 and info["panel_width"] * A.dtype.itemsize in (256, 512), info
		for o in (0, 3, 4, 6, 8, 12):
			ref = oracle.quad_batch(A, np.asfortranarray(X[:, cols]), 14, o, fun="log", fresh_q=True)
			np.testing.assert_allclose(eng.quad_batch(op, X, 14, o, fun="log")[cols], ref, rtol=tol, err_msg=f"{case} P={P} orth={o}")
	## wide panels: deep steps on the 8-wave form; and everything through k_ring_pass instead of k_csr_ring_pass
	P = 300 if A.dtype == np.float32 else 130
	X = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1).astype(A.dtype)
	cols = [0, P // 2, P - 1]
	refs = {o: oracle.quad_batch(A, np.asfortranarray(X[:, cols]), 14, o, fun="log", fresh_q=True) for o in (0, 3, 5, 8, 12)}
	for gen in ("0", "1"):
		monkeypatch.setenv("SLQ_RING_GEN", gen)
		for o, ref in refs.items():
			np.testing.assert_allclose(eng.quad_batch(op, X, 14, o, fun="log")[cols], ref, rtol=tol, err_msg=f"{case} wide gen={gen} orth={o}")
	monkeypatch.delenv("SLQ_RING_GEN")
	## the switches that take the new forms out again (A/B runs): the generic passes on the tiles' row order
	monkeypatch.setenv("SLQ_RING_DEEP", "0")
	np.testing.assert_allclose(eng.quad_batch(op, X, 14, 6, fun="log")[cols], oracle.quad_batch(A, np.asfortranarray(X[:, cols]), 14, 6, fun="log", fresh_q=True), rtol=tol)
	monkeypatch.setenv("SLQ_RING_NARROW", "0")
	Xn = np.asfortranarray(X[:, :40])
	plan = eng.LanczosPlan(op, 40, 14, 3)
	assert plan.describe()["tiles"] == 0
	plan.close()
	np.testing.assert_allclose(eng.quad_batch(op, Xn, 14, 3, fun="log")[:3], oracle.quad_batch(A, np.asfortranarray(Xn[:, :3]), 14, 3, fun="log", fresh_q=True), rtol=tol)
	op.close()


@pytest.mark.parametrize("env", [{}, {"SLQ_RING_PAD_ROWS": "0"}, {"SLQ_SYM_ALPHA": "0"}, {"SLQ_RING_STAGED": "1"}, {"SLQ_RING_STAGED": "0"}])
def test_alpha_pass_stream_forms(oracle, eng, monkeypatch, env):
	"""The alpha-only ring pass (q_c^T A q_c, src/primate/include/lanczos.h:127-129) on its three streams: the upper triangle
	with rows padded to whole chunks of four entries (the branch-free consumer, default), the same stream unpadded
	(SLQ_RING_PAD_ROWS=0), the full rows (SLQ_SYM_ALPHA=0: what a non-symmetric pattern gets), and with either kind of loader forced
	on every panel width (SLQ_RING_STAGED: through registers / LDS-DMA) - wide, 64- and 20-probe panels,
	orth 0 (alpha + update) and 3 (Gram sequence), on a 5-point grid (3 upper entries per row: one padded chunk) and on a band
	with gaps whose rows hold 1..6 upper entries (one or two chunks, differing between the rows a wave walks together)."""
	rng = np.random.default_rng(77)
	monkeypatch.setenv("SLQ_TILES", "2")
	for k, v in env.items():
		monkeypatch.setenv(k, v)
	n = 20011
	offs = [1, 2, 3, 4, 5]
	D = [rng.uniform(0.2, 1.0, n - o) * (rng.random(n - o) > 0.25) for o in offs]
	B = sp.diags(D, offs, shape=(n, n))
	band = (B + B.T + sp.diags(np.full(n, 12.0))).tocsr()
	band.eliminate_zeros()
	band.sort_indices()
	for name, A in (("lap2d", laplacian_2d(150)), ("band", band)):
		op = eng.DeviceOperator(A)
		for P in (130, 64, 20):
			X = np.asfortranarray(np.floor(rng.random((A.shape[0], P)) * 2) * 2 - 1)
			cols = [0, P // 2, P - 1]
			plan = eng.LanczosPlan(op, P, 12, 3)
			info = plan.describe()
			plan.close()
			assert info["tiles"] == 2, info
			for o in (0, 3):
				ref = oracle.quad_batch(A, np.asfortranarray(X[:, cols]), 12, o, fun="log", fresh_q=True)
				np.testing.assert_allclose(eng.quad_batch(op, X, 12, o, fun="log")[cols], ref, rtol=1e-10, err_msg=f"{name} {env} P={P} orth={o}")
		op.close()


def test_opt_in_fused_update_and_alpha_pass(oracle, eng, monkeypatch):
	"""SLQ_FUSED_ALPHA=1 (opt-in, slq_ring_fa.hpp): the update pass of step j also takes step j + 1's alpha dot a fixed lag of tile
	rounds behind its own write front (same-XCD hand-off through L2 counters), entries that cross XCD chunks by a small edge kernel.
	Same alpha as the alpha-only pass up to the order of the sum (q_c . (A q_c - beta q_p), lanczos.h:127-129): per-probe values
	against the oracle on a 2-D and a 3-D grid, orth 1..3, and against the default sequence to rounding. Measured slower than the
	two passes it replaces, hence not the default."""
	monkeypatch.setenv("SLQ_TILES", "2")
	rng = np.random.default_rng(41)
	for A in (laplacian_2d(200), laplacian_3d(40)):
		n = A.shape[0]
		P = 130
		X = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1)
		cols = [0, P // 2, P - 1]
		op = eng.DeviceOperator(A)
		for orth in (1, 2, 3):
			base = eng.quad_batch(op, X, 14, orth, fun="log")
			monkeypatch.setenv("SLQ_FUSED_ALPHA", "1")
			plan = eng.LanczosPlan(op, P, 14, orth)
			assert plan.describe()["fused_alpha"] == 1 and plan.describe()["sequence"] == "fused_gram"
			plan.set_probes(X)
			plan.run()
			got = plan.quadrature("log")
			plan.close()
			monkeypatch.delenv("SLQ_FUSED_ALPHA")
			np.testing.assert_allclose(got, base, rtol=1e-12)
			np.testing.assert_allclose(got[cols], oracle.quad_batch(A, np.asfortranarray(X[:, cols]), 14, orth, fun="log", fresh_q=True), rtol=1e-10)
		op.close()


FUNS_LONG = [("log", {}), ("exp", {"t": -0.1}), ("inv", {}), ("numrank", {}), ("step", {"c": 1.0})]


def _oracle_rule_values(oracle, A, Xc, deg, orth):
	"""Per-probe quadrature values of every f in FUNS_LONG from ONE oracle run (nodes and weights of each probe's rule)."""
	_, nodes, weights, steps = oracle.quad_batch(A, Xc, deg, orth, fun="identity", fresh_q=True, prefer="csr", return_rule=True, nthreads=8)
	vn2 = np.sum(Xc.astype(np.float64) ** 2, axis=0)
	return {f: np.array([np.sum(oracle.apply_fun(f, nodes[i], **kw) * weights[i]) * vn2[i] for i in range(Xc.shape[1])]) for f, kw in FUNS_LONG}, steps


@pytest.mark.parametrize("case", ["lap2d_100", "lap3d_22"])
def test_gram_sequence_through_lost_orthogonality(oracle, eng, monkeypatch, case):
	"""The Gram sequence (the re-orthogonalisation projections assembled from Gram rows the update pass takes, no
	dots pass - the default on tiled operators for every orth 1..8) on LONG recurrences: k = 100 and 300 on a 100^2 / 22^3 grid,
	where Ritz values converge long before the run ends and q_{j+1}.q_{j-s} is no longer O(eps) for the window's columns - the
	regime in which the terms the sequence drops would matter if they did. Per-probe values of log, exp(-0.1 t), inv, numrank and a
	step function with its cut inside the spectrum against the oracle on identical probes, orth in {1, 3, 8}, panels of 20 / 64 /
	130 probes (16, 32 and 64 lanes per row), next to the merged sequence (SLQ_GRAM=0) on the same tiles. Bar for the smooth
	functions: 1e-8 (north_star: 1e-6; measured r04: 1.4e-12 worst, the merged sequence the same). The step function on a
	spectrum this degenerate is ill-posed in ANY arithmetic once ghost Ritz values sit next to the cut - the oracle itself moves by
	1e-3 when the probes change in their last bit - so its yardstick is the oracle's own 1-ulp sensitivity and the merged
	sequence's error. lanczos.h:43-66,133-136; tests/test_lanczos.py:11-20 (the reference's own full-reorth stability test)."""
	monkeypatch.setenv("SLQ_TILES", "2")
	A = laplacian_2d(100) if case == "lap2d_100" else laplacian_3d(22)
	n = A.shape[0]
	rng = np.random.default_rng(2024)
	op = eng.DeviceOperator(A)
	worst = 0.0
	for deg in (100, 300):
		for orth in (1, 3, 8):
			for P in (20, 64, 130):
				X = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1)
				cols = [0, 1, P // 2, P - 1]
				Xc = np.asfortranarray(X[:, cols])
				ref, steps = _oracle_rule_values(oracle, A, Xc, deg, orth)
				Xp = np.asfortranarray(Xc * (1 + np.finfo(np.float64).eps * np.sign(rng.standard_normal(Xc.shape))))
				refp, _ = _oracle_rule_values(oracle, A, Xp, deg, orth)
				err = {}
				for gram in ("1", "0"):
					monkeypatch.setenv("SLQ_GRAM", gram)
					plan = eng.LanczosPlan(op, P, deg, orth)
					info = plan.describe()
					assert info["tiles"] == 2 and info["sequence"] == ("fused_gram" if gram == "1" else "fused"), info
					plan.set_probes(X)
					plan.run()
					assert np.array_equal(plan.tridiag()[2][cols], steps)  # same number of steps as the oracle: no spurious stop
					for f, kw in FUNS_LONG:
						err[gram, f] = np.max(np.abs(plan.quadrature(f, **kw)[cols] - ref[f]) / np.abs(ref[f]))
					plan.close()
				monkeypatch.delenv("SLQ_GRAM")
				for f, _ in FUNS_LONG:
					sens = np.max(np.abs(refp[f] - ref[f]) / np.abs(ref[f]))
					tol = 1e-8 if f != "step" else max(1e-8, 30.0 * sens, 3.0 * err["0", f])
					assert err["1", f] <= tol, f"{case} k={deg} orth={orth} P={P} f={f}: gram {err['1', f]:.2e} merged {err['0', f]:.2e} oracle 1-ulp sensitivity {sens:.2e}"
					if f != "step":
						worst = max(worst, err["1", f])
	op.close()
	assert worst < 1e-10, worst  # (what was measured; the bar above is what is promised)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gram_sequence_on_an_ill_conditioned_operator(oracle, eng, monkeypatch, dtype):
	"""Advisor finding r03: the Gram sequence forms d_i as a difference of O(|A|) terms and applies the reference's skip threshold
	(lanczos.h:53,62) to it - exercise it where that is least comfortable: D L D with D spanning 1e-2 .. 1e2 on the 5-point pattern
	(condition ~1e8 times the grid's, strongly clustered small eigenvalues), fp64 and fp32, every window 1..8, k = 60. On such an
	operator a 60-step recurrence with a short window is itself unstable in any arithmetic: the fp64 device results sit 1e-4..1e-7
	from the fp64 oracle with EITHER sequence (partial re-orthogonalisation does not hold the basis together, rounding is amplified
	by 1/beta), and fp32 tridiagonals differ in their leading digits. What is asserted is therefore that the Gram sequence is no
	further from the fp64 oracle than the merged sequence (the direct dots) by more than a small factor, and within the north_star's
	1e-6 .. the fp32 oracle's own distance."""
	import scipy.sparse as spx

	monkeypatch.setenv("SLQ_TILES", "2")
	rng = np.random.default_rng(99)
	L2 = laplacian_2d(100)
	n = L2.shape[0]
	dsc = 10.0 ** rng.uniform(-2.0, 2.0, n)
	A64 = (spx.diags(dsc) @ L2 @ spx.diags(dsc)).tocsr()
	A64.sort_indices()
	A = A64.astype(dtype)
	op = eng.DeviceOperator(A)
	P, deg = 64, 60
	X = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1).astype(dtype)
	cols = [0, 1, P // 2, P - 1]
	Xc = np.asfortranarray(X[:, cols])
	for orth in range(1, 9):
		ref64 = oracle.quad_batch(A64, Xc.astype(np.float64), deg, orth, fun="log", fresh_q=True, prefer="csr")
		noise = 0.0
		if dtype == np.float32:
			noise = np.max(np.abs(oracle.quad_batch(A, Xc, deg, orth, fun="log", fresh_q=True, prefer="csr") - ref64) / np.abs(ref64))
		err = {}
		for gram in ("1", "0"):
			monkeypatch.setenv("SLQ_GRAM", gram)
			plan = eng.LanczosPlan(op, P, deg, orth)
			assert plan.describe()["sequence"] == ("fused_gram" if gram == "1" else "fused")
			plan.set_probes(X)
			plan.run()
			err[gram] = np.max(np.abs(plan.quadrature("log")[cols] - ref64) / np.abs(ref64))
			plan.close()
		monkeypatch.delenv("SLQ_GRAM")
		bar = max(1e-6, 10.0 * err["0"], 2.0 * noise)
		assert err["1"] <= bar, f"{np.dtype(dtype).name} orth={orth}: gram {err['1']:.2e} merged {err['0']:.2e} fp32-oracle noise {noise:.2e}"
	op.close()


def test_tall_skinny_mfma_products(eng):
	"""slq_dmat_gemm_tn / _nn (fp64 MFMA) against NumPy, ragged sizes on every edge."""
	rng = np.random.default_rng(0)
	for n, ma, mb in [(1000, 16, 16), (4099, 37, 70), (20011, 130, 5), (517, 3, 129)]:
		A, B = rng.standard_normal((n, ma)), rng.standard_normal((n, mb))
		dA, dB = eng.DeviceMatrix(n, ma + 2), eng.DeviceMatrix(n, mb + 3)
		dA.set(1, A)
		dB.set(2, B)
		np.testing.assert_allclose(dA.get(1, ma), A)
		C = dA.tn(1, ma, dB, 2, mb)
		np.testing.assert_allclose(C, A.T @ B, rtol=1e-12, atol=1e-10)
		M = rng.standard_normal((ma, mb))
		dB.add_product(2, dA, 1, M, alpha=-0.5, beta=1.0)
		np.testing.assert_allclose(dB.get(2, mb), B - 0.5 * A @ M, rtol=1e-12, atol=1e-11)
		dB.add_product(2, dA, 1, M, alpha=2.0, beta=0.0)
		np.testing.assert_allclose(dB.get(2, mb), 2.0 * A @ M, rtol=1e-12, atol=1e-11)
		assert np.all(dB.get(0, 2) == 0) and np.all(dA.get(0, 1) == 0)  # neighbours untouched
		with pytest.raises(ValueError):
			dA.add_product(0, dA, 1, np.zeros((ma, 3)))  # overlapping in/out columns
		dA.close(), dB.close()


@pytest.mark.gpu
def test_device_built_streams_equal_the_host_built_ones(oracle, eng, monkeypatch):
	"""r04 (slq_build.hpp): operators with ring-sized tiles have their stored CSR, upper triangle and tile streams
	built on the device. SLQ_DEVICE_BUILD=2 builds every array both ways and the library raises if one byte differs; here for a
	5-point and a 7-point grid, weighted symmetric and non-symmetric values (the latter: no upper triangle), fp64 and fp32, and the
	2- and 4-merged tiles of narrow panels; the values of a run are then checked against the oracle as everywhere else."""
	import scipy.sparse as sp

	monkeypatch.setenv("SLQ_TILES", "2")
	monkeypatch.setenv("SLQ_DEVICE_BUILD", "2")
	rng = np.random.default_rng(11)

	def weighted(A, symmetric):
		C = sp.coo_matrix(A)
		w = rng.uniform(0.5, 1.5, C.nnz)
		W = sp.csr_matrix((w, (C.row, C.col)), shape=A.shape)
		W = ((W + W.T) * 0.5).tocsr() if symmetric else W
		W = (W + sp.diags(np.asarray(abs(W).sum(axis=1)).ravel() + 1.0)).tocsr()
		W.sort_indices()
		return W

	cases = [("lap2d_150", laplacian_2d(150), True), ("lap3d_30", laplacian_3d(30), True), ("lap3d_24_f32", laplacian_3d(24, np.float32), True),
	         ("weighted 2-D", weighted(laplacian_2d(120), True), True), ("weighted 3-D, not symmetric", weighted(laplacian_3d(22), False), False)]  # fmt: skip
	for name, A, sym in cases:
		op = eng.DeviceOperator(A)  # (raises when the two builds differ)
		for P in (130, 64, 32) if A.dtype == np.float64 else (130, 64):  # (panel rows of 512, 256 and 128 bytes: unmerged, 2- and 4-merged tiles)
			plan = eng.LanczosPlan(op, P, 12, 3)
			info = plan.describe()
			assert info["tiles"] == 2 and info["upper_alpha"] == int(sym), (name, P, info)
			plan.generate_probes("rademacher", seed=3)
			V = plan.get_probes()[:, [0, P - 1]]
			plan.run()
			if sym:
				q = plan.quadrature("log")
				ref = oracle.quad_batch(sp.csr_matrix(A, dtype=np.float64), np.asfortranarray(V, dtype=np.float64), 12, 3, fun="log", fresh_q=True)
				assert np.max(np.abs(q[[0, P - 1]] / ref - 1.0)) < (1e-9 if A.dtype == np.float64 else 2e-4), (name, P)
			else:  # (the oracle's recurrence never checks symmetry either: alpha and beta)
				a, b, _ = plan.tridiag()
				for k, c in enumerate((0, P - 1)):
					ar, br, Qr = np.zeros(13), np.zeros(13), np.zeros((A.shape[0], 3), order="F")
					oracle.lanczos(A, V[:, k].copy(), 12, 1e-8, 3, ar, br, Qr)
					np.testing.assert_allclose(a[c][:12], ar[:12], rtol=1e-10, atol=1e-10)
					np.testing.assert_allclose(b[c][1:12], br[1:12], rtol=1e-10, atol=1e-10)
			plan.close()
		op.close()
	## and one operator the device build leaves to the host (no tiles: a random graph) still builds
	G = sp.random(6000, 6000, density=0.002, random_state=5, format="csr")
	G = (G + G.T + sp.identity(6000) * 10.0).tocsr()
	op = eng.DeviceOperator(G)
	plan = eng.LanczosPlan(op, 8, 5, 0)
	assert plan.describe()["tiles"] == 0
	plan.close(), op.close()
