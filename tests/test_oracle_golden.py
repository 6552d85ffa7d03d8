"""The oracle (oracle/slq_oracle.c) against vectors produced by the reference's own Python.

Every expected value here was written by tests/golden/make_golden.py from "pure" reference code
(src/primate/lanczos.py:196-238 NumPy twins, integrate.quadrature, tridiag.eigh_tridiag, fttr,
special.param_callable) unless a test says "injected". CPU only.
"""

import numpy as np
import pytest

from conftest import laplacian_2d

FUNS = {
	"identity": {}, "log": {}, "exp": {}, "sqrt": {}, "inv": {}, "abs": {},
	"smoothstep": {"a": 0.5, "b": 6.0}, "numrank": {},
}  # fmt: skip


def _run_lanczos(O, A, v, deg, orth, ncv, dtype=np.float64):
	al, be = np.zeros(deg + 1, dtype=dtype), np.zeros(deg + 1, dtype=dtype)
	Q = np.zeros((A.shape[0], ncv), dtype=dtype, order="F")
	steps = O.lanczos(A, v, deg, 1e-8, orth, al, be, Q)
	return al, be, Q, steps


def test_kat_full_reorth_matches_reference_twin(oracle, golden):
	## inputs of tests/test_lanczos.py:11-20 (reference); full reorth pins alpha/beta to rounding
	A, v0 = golden["kat_A"], golden["kat_v0"]
	al, be, Q, steps = _run_lanczos(oracle, A, v0, 50, 50, 50)
	assert steps == 50
	np.testing.assert_allclose(al[:50], golden["kat_alpha_o50_c50"], rtol=0, atol=1e-12 * np.abs(al).max())
	np.testing.assert_allclose(be[:50], golden["kat_beta_o50_c50"], rtol=0, atol=1e-12 * np.abs(be).max())
	## columns are defined up to rounding; compare the projector-free quantity |Q^T Q_ref| = I
	G = np.abs(Q.T @ golden["kat_Q_o50_c50"])
	np.testing.assert_allclose(G, np.eye(50), atol=1e-8)
	## the reference test's own assertion
	from scipy.linalg import eigvalsh_tridiagonal

	assert np.allclose(eigvalsh_tridiagonal(al[:50], be[1:50]), golden["kat_eigvalsh"])


@pytest.mark.parametrize("orth,ncv", [(0, 2), (3, 3), (3, 50), (10, 20)])
def test_kat_partial_reorth_head(oracle, golden, orth, ncv):
	## without full reorth the late coefficients are chaotic (SURVEY.md §7 "Parity metric"): pin
	## the head, where both implementations are still in the exact-arithmetic regime
	al, be, _, _ = _run_lanczos(oracle, golden["kat_A"], golden["kat_v0"], 50, orth, ncv)
	np.testing.assert_allclose(al[:6], golden[f"kat_alpha_o{orth}_c{ncv}"][:6], rtol=1e-9)
	np.testing.assert_allclose(be[:6], golden[f"kat_beta_o{orth}_c{ncv}"][:6], rtol=1e-9, atol=1e-300)


def test_early_stop_rule(oracle, golden):
	## lanczos.h:140-142: break when beta[j+1] < sqrt(n)*rtol, before the next column is written
	al, be, _, steps = _run_lanczos(oracle, golden["stop_A"], golden["stop_v"], 20, 20, 20)
	ref_b = golden["stop_beta"]
	assert steps == 5 and np.count_nonzero(ref_b[6:]) == 0 and np.count_nonzero(be[6:20]) == 0
	np.testing.assert_allclose(be[:5], ref_b[:5], rtol=1e-10)
	np.testing.assert_allclose(al[:5], golden["stop_alpha"][:5], rtol=1e-10)
	assert be[5] < np.sqrt(40) * 1e-8 and ref_b[5] < np.sqrt(40) * 1e-8


@pytest.mark.parametrize("orth", [0, 3, 20])
def test_laplacian_recurrence_and_rule(oracle, golden, orth):
	L, V = laplacian_2d(int(golden["lap_m"])), golden["lap_probes"]
	for prefer in ["csc", "csr"]:
		q, nodes, weights, steps = oracle.quad_batch(L, V, 20, orth, fun="log", return_rule=True, prefer=prefer)
		assert np.all(steps == 20)
		np.testing.assert_allclose(nodes, golden[f"lap_nodes_o{orth}"], rtol=0, atol=5e-13)
		np.testing.assert_allclose(weights, golden[f"lap_weights_o{orth}"], rtol=0, atol=5e-13)
		np.testing.assert_allclose(q, golden[f"lap_quad_log_o{orth}"], rtol=1e-12)
	for j in range(3):
		al, be, _, _ = _run_lanczos(oracle, L, V[:, j], 20, orth, 20)
		np.testing.assert_allclose(al[:20], golden[f"lap_alpha_o{orth}"][j], rtol=1e-9)
		np.testing.assert_allclose(be[:20], golden[f"lap_beta_o{orth}"][j], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("fun", list(FUNS) + ["exp_t"])
def test_laplacian_quad_all_builtin_functions(oracle, golden, fun):
	L, V = laplacian_2d(int(golden["lap_m"])), golden["lap_probes"]
	name, kw = ("exp", {"t": -0.1}) if fun == "exp_t" else (fun, FUNS[fun])
	for orth in [0, 3, 20]:
		q = oracle.quad_batch(L, V, 20, orth, fun=name, **kw)
		np.testing.assert_allclose(q, golden[f"lap_quad_{fun}_o{orth}"], rtol=1e-11)


def test_stale_ring_semantics_injected(oracle, golden):
	"""MatrixFunction.quad never clears Q between probes (operators.py:138-148), so for orth > 0
	the MGS sweep of probe j sees probe j-1's late Lanczos vectors. fresh_q=False reproduces that
	("injected": the reference's MatrixFunction ran over oracle.lanczos)."""
	L, V = laplacian_2d(int(golden["lap_m"])), golden["lap_probes"]
	for orth in [0, 3, 20]:
		q = oracle.quad_batch(L, V, 20, orth, fun="log", fresh_q=False)
		np.testing.assert_allclose(q, golden[f"mf_quad_log_o{orth}"], rtol=1e-12)
	## orth = 0 never calls orth_vector (lanczos.h:133), so stale == fresh there ...
	np.testing.assert_allclose(golden["mf_quad_log_o0"], golden["lap_quad_log_o0"], rtol=1e-12)
	## ... the first probe always sees a clean ring ...
	np.testing.assert_allclose(golden["mf_quad_log_o3"][0], golden["lap_quad_log_o3"][0], rtol=1e-12)
	## ... and later probes differ at O(1/sqrt(n)): a property of the reference, recorded here
	assert np.max(np.abs(golden["mf_quad_log_o3"][1:] / golden["lap_quad_log_o3"][1:] - 1)) > 1e-4


def test_tridiagonal_rule(oracle, golden):
	d, e = golden["tri_d"], golden["tri_e"]
	nodes, weights = oracle.quadrature_gw(d, e)
	np.testing.assert_allclose(nodes, golden["tri_nodes"], rtol=0, atol=1e-13)
	np.testing.assert_allclose(weights, golden["tri_weights"], rtol=0, atol=1e-13)
	ew, Z, rc = oracle.tridiag_ql(d, e, want_vectors=True)
	assert rc == 0
	order = np.argsort(ew)
	## Z is row-major "rows receive rotations": Z[k, i] = component k of eigenvector i
	np.testing.assert_allclose(np.abs(Z[:, order]), golden["tri_Y_abs"], atol=1e-10)
	np.testing.assert_allclose(oracle.fttr(golden["tri_nodes"], d, e, 30), golden["tri_fttr_raw"], rtol=1e-9)


def test_tridiag_accuracy_like_reference_test(oracle):
	## tests/test_tridiagonal.py:26-44 (reference): d = 150, eigenvalues to 1e-14 vs LAPACK
	from scipy.linalg import eigvalsh_tridiagonal

	for seed in [1234, 4756, 43, 102]:
		rng = np.random.default_rng(seed)
		d = rng.uniform(size=150, low=0.0, high=1.0)
		e = np.append([0.0], rng.uniform(size=149, low=0.0, high=0.5))
		ew, rc = oracle.tridiag_ql(d, e)
		assert rc == 0
		assert np.max(np.abs(np.sort(ew) - eigvalsh_tridiagonal(d, e[1:]))) <= 2e-14


def test_spectral_functions(oracle, golden):
	x = golden["fun_x"]
	with np.errstate(all="ignore"):
		for name, kw in {**FUNS, "softsign": {"q": 10}}.items():
			got, ref = oracle.apply_fun(name, x, **kw), golden[f"fun_{name}"]
			ok = np.isfinite(ref)
			np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-14, atol=0)
			assert np.array_equal(np.isnan(got), np.isnan(ref))
		np.testing.assert_allclose(oracle.apply_fun("exp", x, t=-0.1), golden["fun_exp_t"], rtol=1e-14)


def test_numpy_restatement_agrees_with_c(oracle, golden):
	L, V = laplacian_2d(int(golden["lap_m"])), golden["lap_probes"]
	for orth, ncv in [(0, 2), (3, 3), (3, 20), (20, 20)]:
		al, be, Q, _ = _run_lanczos(oracle, L, V[:, 1], 20, orth, ncv)
		al2, be2, Q2 = np.zeros(21), np.zeros(21), np.zeros((L.shape[0], ncv), order="F")
		oracle.np_lanczos(lambda x: L @ x, V[:, 1].copy(), 20, 1e-8, orth, al2, be2, Q2)
		np.testing.assert_allclose(al, al2, rtol=1e-9)
		np.testing.assert_allclose(be, be2, rtol=1e-9)
	th, tau = oracle.np_quadrature(golden["tri_d"], golden["tri_e"])
	np.testing.assert_allclose(th, golden["tri_nodes"], atol=1e-14)
	np.testing.assert_allclose(tau, golden["tri_weights"], atol=1e-14)


def test_operator_kinds_agree(oracle):
	rng = np.random.default_rng(3)
	L = laplacian_2d(9)
	x = rng.standard_normal(81)
	y = L @ x
	for dt, tol in [(np.float64, 1e-14), (np.float32, 1e-5)]:
		for A in [L.astype(dt), L.toarray().astype(dt)]:
			for prefer in ["csc", "csr"]:
				np.testing.assert_allclose(oracle.make_operator(A, dtype=dt, prefer=prefer).matvec(x), y, atol=tol * 10)

		class PyOp:  # the plugin surface of src/primate/include/pylinop.h:22-29
			shape, dtype = L.shape, np.dtype(dt)

			def matvec(self, v):
				return L @ v

		np.testing.assert_allclose(oracle.make_operator(PyOp(), dtype=dt).matvec(x), y, atol=tol * 10)
	with pytest.raises(ValueError):
		oracle.make_operator(object.__new__(type("NoMatvec", (), {"shape": (3, 3)})))


def test_c2_integer_structure(golden):
	## bit-exact items of BASELINE.md: nnz, n, row-degree histogram of the 1000x1000-grid Laplacian
	assert list(golden["c2_nnz_n"]) == [4996000, 1000000]
	assert list(golden["c2_rowdeg_hist"]) == [0, 0, 0, 4, 3992, 996004]
