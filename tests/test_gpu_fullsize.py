"""BASELINE.json configs 3-5 at FULL operator size, with reduced probe counts, checked through
size-independent properties (no oracle run at these sizes): identities between the quadrature and
the action paths, analytic answers where the operator has one, agreement across reorthogonalisation
depths. Needs a real MI355X and a few GB..100 GB of HBM.
"""

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import laplacian_3d

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
	from primate_amd import engine

	return engine


def test_config4_heat_kernel_diag_fp32_full_size(eng, oracle):
	"""configs[3]: diag(exp(-t L)) of the 126^3 7-point Laplacian, k = 50, fp32. The operator is a
	Kronecker sum, so the exact diagonal is an outer product of 1-D heat-kernel diagonals. The plan must be on the
	ring-fed tiles (the default path of this operator), and two of its columns are checked against the fp32 oracle at
	full size - quadrature and the action f(A)v - not only through statistics."""
	m, t, k, P = 126, 0.1, 50, 128
	A = laplacian_3d(m, dtype=np.float32)
	assert (A.shape[0], A.nnz) == (2000376, 13907376)  # bit-exact structure
	T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m)).toarray()
	w, U = np.linalg.eigh(T)
	d1 = ((U * np.exp(-t * w)) @ U.T).diagonal()
	exact = np.einsum("i,j,k->ijk", d1, d1, d1).ravel()
	op = eng.DeviceOperator(A)
	acc = eng.DiagAccumulator(A.shape[0], ctx=op.ctx)
	plan = eng.LanczosPlan(op, P, k, 3, keep_basis=True)
	assert plan.describe()["tiles"] == 2 and plan.describe()["panel_width"] == 128  # 512-byte panel rows: merged tiles (slq_ring.hpp)
	plan.generate_probes("rademacher", seed=1234)
	V = plan.get_probes()
	plan.run()
	acc.update(plan, "exp", t=-t)
	## the fp32 oracle (lanczos.h:92-149 restated, CSC product) on the first and the last column, same probes
	cols = [0, P - 1]
	Vc = np.asfortranarray(V[:, cols])
	ref_q = oracle.quad_batch(A, Vc, k, 3, fun="exp", t=-t, fresh_q=True)
	Y = plan.fun_action("exp", t=-t)
	## ... and the action itself on the last column: ||v|| Q Y (f(theta) * Y[0, :]) from the oracle's recurrence with the
	## whole basis kept (what MatrixFunction._matvec computes, src/primate/operators.py:113-124)
	al, be, Q = np.zeros(k + 1, dtype=np.float32), np.zeros(k + 1, dtype=np.float32), np.zeros((A.shape[0], k), dtype=np.float32, order="F")
	v = np.ascontiguousarray(V[:, P - 1])
	assert oracle.lanczos(A, v.copy(), k, 1e-8, 3, al, be, Q) == k
	th, Yv = np.linalg.eigh(np.diag(al[:k].astype(np.float64)) + np.diag(be[1:k].astype(np.float64), 1) + np.diag(be[1:k].astype(np.float64), -1))
	ref_y = np.linalg.norm(v.astype(np.float64)) * (Q.astype(np.float64) @ (Yv @ (np.exp(-t * th) * Yv[0, :])))
	np.testing.assert_allclose(Y[:, P - 1], ref_y, rtol=0, atol=2e-4 * np.abs(ref_y).max())
	del Q
	np.testing.assert_allclose(np.einsum("ij,ij->j", V[:, cols].astype(np.float64), Y[:, cols].astype(np.float64)), ref_q, rtol=3e-4)
	del V, Y
	numer, denom, rmean, cnt = acc.get()
	assert cnt == P and np.all(denom == P)  # Rademacher: v*v = 1 exactly, P times
	est = numer / denom
	assert np.linalg.norm(est - exact) / np.linalg.norm(exact) < 2.5 / np.sqrt(P) * 0.2  # measured 0.011 at P=512
	## sum of the diagonal = trace: compare with the quadrature path on the same probes
	plan2 = eng.LanczosPlan(op, P, k, 3)
	plan2.generate_probes("rademacher", seed=1234)
	plan2.run()
	assert plan2.describe()["tiles"] == 2
	q = plan2.quadrature("exp", t=-t)
	np.testing.assert_allclose(q[cols], ref_q, rtol=3e-4)  # the quadrature path against the oracle, per probe
	assert abs(numer.sum() / P - q.mean()) < 2e-4 * abs(q.mean())  # v^T f(A) v by action vs by quadrature (fp32)
	assert abs(q.mean() - exact.sum()) < 6 * q.std(ddof=1) / np.sqrt(P)


def test_config3_estrada_index_full_size(eng):
	"""configs[2]: tr exp(A) of a G(n, 16/n) graph, n = 5e5, k = 40: quadrature and action agree, and the
	reorthogonalisation depth does not move the quadrature."""
	n, k, P = 500000, 40, 32
	rng = np.random.default_rng(1234)
	mm = int(n * 16 / 2)
	i, j = rng.integers(0, n, mm), rng.integers(0, n, mm)
	keep = i != j
	W = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(n, n)).tocsr()
	W = ((W + W.T) > 0).astype(np.float64).tocsr()
	W.sort_indices()
	assert W.nnz == 7999866  # SURVEY.md §8 C3 asks for the exact count to be recorded
	op = eng.DeviceOperator(W)
	qs = {}
	for orth in (0, 3, k):
		plan = eng.LanczosPlan(op, P, k, orth)
		plan.generate_probes("rademacher", seed=1234)
		plan.run()
		qs[orth] = plan.quadrature("exp")
		a, b, steps = plan.tridiag()
		assert np.all(steps == k)
		np.testing.assert_allclose(a[:, 0] * n, eng_quad_identity(plan, op, P, k, orth), rtol=1e-10)
		plan.close()
	np.testing.assert_allclose(qs[3], qs[0], rtol=1e-8)
	np.testing.assert_allclose(qs[k], qs[0], rtol=1e-8)
	plan = eng.LanczosPlan(op, 16, k, 3, keep_basis=True)
	plan.generate_probes("rademacher", seed=1234)
	V = plan.get_probes()
	plan.run()
	Y = plan.fun_action("exp")
	np.testing.assert_allclose(np.einsum("ij,ij->j", V, Y), qs[3][:16], rtol=1e-8)  # v^T (f(A) v) == quadrature
	Y1 = plan.fun_action("identity")
	np.testing.assert_allclose(Y1, W @ V, rtol=1e-9, atol=1e-9)  # A v is in the Krylov space: exact


def test_config3_xtrace_as_worded_full_size(eng):
	"""configs[2] as BASELINE.json words it: xtrace of exp(A) on the G(5e5, 16/n) graph, k = 40, 512 sample
	vectors in batches of 128, device-drawn. The estimate lies within 3 sigma of a hutch estimate with the same
	budget, and the column-sharded driver with a world of one (every f(A)-product goes through its shard
	bookkeeping) returns the same number as the plain call."""
	from primate_amd.distributed import sharded_xtrace
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import hutch, xtrace

	n, k = 500000, 40
	rng = np.random.default_rng(1234)
	mm = int(n * 16 / 2)
	i, j = rng.integers(0, n, mm), rng.integers(0, n, mm)
	keep = i != j
	W = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(n, n)).tocsr()
	W = ((W + W.T) > 0).astype(np.float64).tocsr()
	W.sort_indices()
	M = MatrixFunction(W, fun="exp", deg=k, orth=3)
	est, info = xtrace(M, batch=128, count=512, seed=1234, device_rng=True, full=True)
	h, hinfo = hutch(M, pdf="device:rademacher", converge="count", count=512, seed=1234, batch=256, full=True)
	sigma = np.sqrt(float(np.ravel(hinfo.estimator._cov.covariance(ddof=1))[0]) / len(hinfo.estimator))  # standard error of the hutch mean
	assert abs(est - h) < 3 * sigma, (est, h, sigma)
	assert abs(est / 8.0589e7 - 1) < 5e-3  # round-1 measurement of this operator: 8.0589e7
	assert sharded_xtrace(M, count=512, batch=128, seed=1234, device_rng=True) == est  # no process group: the plain call
	## a world of one through the shard path: every f(A)-product is cut, "gathered" and copied back by the bookkeeping
	## the multi-rank driver uses (the all-gather itself degenerates to a device copy)
	same = xtrace(M, batch=128, count=512, seed=1234, device_rng=True, _shard=(0, 1, lambda src, ncols, dst: dst.copy_from(0, src, 0, ncols)))
	np.testing.assert_allclose(same, est, rtol=1e-12)


def test_config1_dense_spd_5000_on_the_matrix_cores(oracle, eng):
	"""configs[0] on the HIP path: dense SPD 5000 x 5000 (fp64 MFMA operator), Rademacher probes, k = 20. 64 and 128
	probes cover the 64-column launch and the two-halves launch of the fused three-term epilogue, 20 the 32-column
	one; four columns of each batch are checked against the oracle on the same probes, orth 0 and 3."""
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import hutch

	n, k = 5000, 20
	rng = np.random.default_rng(1234)
	B = rng.standard_normal((n, n))
	A = B @ B.T / n + np.eye(n)
	A = np.asfortranarray((A + A.T) / 2)
	op = eng.DeviceOperator(A)
	for P in (64, 128, 20):
		V = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1)
		cols = [0, 1, P // 2, P - 1]
		for orth in (0, 3):
			for fun in ("identity", "log"):
				got = eng.quad_batch(op, V, k, orth, fun=fun)
				ref = oracle.quad_batch(A, V[:, cols], k, orth, fun=fun, fresh_q=True)
				np.testing.assert_allclose(got[cols], ref, rtol=1e-10, err_msg=f"P={P} orth={orth} {fun}")
			## f = identity: the Gauss rule reproduces v^T A v exactly, every column
			np.testing.assert_allclose(eng.quad_batch(op, V, k, orth, fun="identity"), np.einsum("ij,ij->j", V, A @ V), rtol=1e-11)
	## the configs[0] call itself, on the device: hutch with 64 Rademacher probes, f = identity
	M = MatrixFunction(A, fun="identity", deg=k, orth=3)
	est, info = hutch(M, pdf="rademacher", converge="count", count=64, seed=1234, full=True)
	from primate_amd.random import isotropic

	V = isotropic(pdf="rademacher", seed=1234)(size=(n, 64))
	np.testing.assert_allclose(est, np.mean(np.einsum("ij,ij->j", V, A @ V)), rtol=1e-11)
	assert abs(est - np.trace(A)) < 6 * np.std(np.einsum("ij,ij->j", V, A @ V), ddof=1) / 8


def eng_quad_identity(plan, op, P, k, orth):
	"""f = identity: sum theta*tau*||v||^2 = v^T A v (exact for any k >= 1)."""
	return plan.quadrature("identity")


def test_config5_full_reorth_n1e7(eng):
	"""configs[4] (one GPU's slice, 8 probes): n = 1e7 banded SPD CSR, k = 80, full reorthogonalisation
	(81 ring slots resident = 104 GB). Full reorth and no reorth give the same Gauss rule sums."""
	n, k, P = 10_000_000, 80, 8
	rng = np.random.default_rng(1234)
	offs = [1, 2, 3, 57, 411, 977, 1993]
	S = sp.diags([rng.uniform(-1, 0, n - o) for o in offs], offs, shape=(n, n))
	S = (S + S.T).tocsr()
	A = (S + sp.diags(np.asarray(abs(S).sum(axis=1)).ravel() + 0.1)).tocsr()
	A.sort_indices()
	assert A.shape[0] == n and A.nnz == n + 2 * sum(n - o for o in offs)
	op = eng.DeviceOperator(A)
	out = {}
	for orth in (k, 0):
		plan = eng.LanczosPlan(op, P, k, orth)
		if orth == k:
			assert plan.workspace_bytes > 100e9
		plan.generate_probes("rademacher", seed=7)
		plan.run()
		q, nodes, weights = plan.quadrature("log", return_rule=True)
		np.testing.assert_allclose(weights.sum(axis=1), 1.0, atol=1e-12)
		assert nodes.min() > 0.09  # Gershgorin: diagonally dominant by 0.1
		out[orth] = (q, plan.quadrature("numrank"))
		plan.close()
	np.testing.assert_allclose(out[k][0], out[0][0], rtol=1e-9)
	np.testing.assert_allclose(out[k][1], n, rtol=1e-12)  # every Ritz value > 1e-6: full numerical rank
