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
	ring-fed tiles (the default path of this operator), and two of its columns are checked against the oracle at
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
	## the oracle on the first and the last column, same probes - run in fp64 on the same (exactly representable) operator and
	## probes: at n = 2e6 the fp32 oracle is the noisier of the two implementations (its reductions are one sequential fp32
	## accumulator: 0.7 % off on this column, basis orthogonal to 7e-3, against 3e-6 for the device's blocked sums -
	## scripts/check_c4_action.py), so the exact-arithmetic value of the same recurrence is the yardstick for fp32 here; the
	## fp32-vs-fp32 comparison stays at 40^3 (test_gpu_parity.py)
	cols = [0, P - 1]
	A64 = A.astype(np.float64)
	Vc = np.asfortranarray(V[:, cols].astype(np.float64))
	ref_q = oracle.quad_batch(A64, Vc, k, 3, fun="exp", t=-t, fresh_q=True)
	Y = plan.fun_action("exp", t=-t)
	## ... and the action itself on the last column: ||v|| Q Y (f(theta) * Y[0, :]) from the oracle's recurrence with the
	## whole basis kept (what MatrixFunction._matvec computes, src/primate/operators.py:113-124)
	al, be, Q = np.zeros(k + 1), np.zeros(k + 1), np.zeros((A.shape[0], k), order="F")
	v = np.ascontiguousarray(Vc[:, 1])
	assert oracle.lanczos(A64, v.copy(), k, 1e-8, 3, al, be, Q) == k
	th, Yv = np.linalg.eigh(np.diag(al[:k]) + np.diag(be[1:k], 1) + np.diag(be[1:k], -1))
	ref_y = np.linalg.norm(v) * (Q @ (Yv @ (np.exp(-t * th) * Yv[0, :])))
	np.testing.assert_allclose(Y[:, P - 1], ref_y, rtol=0, atol=2e-4 * np.abs(ref_y).max())
	del Q
	np.testing.assert_allclose(np.einsum("ij,ij->j", Vc, Y[:, cols].astype(np.float64)), ref_q, rtol=3e-4)
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
	## ... and fp32 against fp32 at full size: since r04 the fp32 oracle's reductions are blocked the way Eigen's packet reductions are (32 running
	## sums, oracle/slq_oracle_impl.h: dot) instead of one sequential accumulator, which was 0.7 % off at this n - the checker is no longer the noisy side
	plan3 = eng.LanczosPlan(op, P, k, 3)
	plan3.generate_probes("rademacher", seed=1234)
	V32 = np.asfortranarray(plan3.get_probes()[:, cols])
	plan3.close()
	ref_q32 = oracle.quad_batch(A, V32, k, 3, fun="exp", t=-t, fresh_q=True)
	np.testing.assert_allclose(ref_q32, ref_q, rtol=3e-4)  # the fp32 oracle against the fp64 one
	np.testing.assert_allclose(q[cols], ref_q32, rtol=3e-4)
	assert abs(numer.sum() / P - q.mean()) < 2e-4 * abs(q.mean())  # v^T f(A) v by action vs by quadrature (fp32)
	assert abs(q.mean() - exact.sum()) < 6 * q.std(ddof=1) / np.sqrt(P)


def test_config3_estrada_index_full_size(eng, oracle):
	"""configs[2]: tr exp(A) of a G(n, 16/n) graph, n = 5e5, k = 40: the oracle at full size on two columns of the quadrature at
	orth 0 and 3 (the operator's gathers are not cache-served: orth 3 must run the stored-u sequence) and on one column of the
	action exp(A)v; quadrature and action agree, and the reorthogonalisation depth does not move the quadrature
	(src/primate/trace.py:233-315 is the driver these feed)."""
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
		if orth == 3:
			assert plan.describe()["sequence"] == "fused_stored_u", plan.describe()
		plan.generate_probes("rademacher", seed=1234)
		Vo = np.asfortranarray(plan.get_probes()[:, [0, P - 1]])
		plan.run()
		qs[orth] = plan.quadrature("exp")
		a, b, steps = plan.tridiag()
		assert np.all(steps == k)
		if orth in (0, 3):  # the oracle on the first and last probe, full size (0.2 s per probe)
			np.testing.assert_allclose(qs[orth][[0, P - 1]], oracle.quad_batch(W, Vo, k, orth, fun="exp", fresh_q=True, prefer="csr"), rtol=1e-10, err_msg=f"orth={orth}")
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
	## the action itself against the oracle's recurrence with the whole basis kept: ||v|| Q Y (f(theta) * Y[0, :])
	## (MatrixFunction._matvec, src/primate/operators.py:113-124), last column
	al, be, Q = np.zeros(k + 1), np.zeros(k + 1), np.zeros((n, k), order="F")
	v = np.ascontiguousarray(V[:, 15])
	assert oracle.lanczos(W, v.copy(), k, 1e-8, 3, al, be, Q) == k
	th, Yv = np.linalg.eigh(np.diag(al[:k]) + np.diag(be[1:k], 1) + np.diag(be[1:k], -1))
	ref_y = np.linalg.norm(v) * (Q @ (Yv @ (np.exp(th) * Yv[0, :])))
	np.testing.assert_allclose(Y[:, 15], ref_y, rtol=0, atol=1e-9 * np.abs(ref_y).max())
	del Q
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


def circulant_band(n, offs=(1, 2, 3, 57, 411, 977, 1993), weights=(1.0, 0.85, 0.7, 0.9, 0.6, 0.75, 0.8), shift=0.1):
	"""Symmetric circulant band matrix with 2 len(offs) + 1 nonzeros per row (BASELINE.json configs[4]: n = 1e7, 15 per row):
	A[i, (i +- o) mod n] = -w_o, A[i, i] = 2 sum(w) + shift. Its eigenvalues are known in closed form,
	lambda_k = d - 2 sum_o w_o cos(2 pi k o / n), so the number of eigenvalues above any cut is exact. Returns (A, lambda)."""
	offs, w = np.asarray(offs), np.asarray(weights, dtype=np.float64)
	d = 2.0 * w.sum() + shift
	rel = np.concatenate([-offs[::-1], [0], offs])  # ascending relative columns
	val = np.concatenate([-w[::-1], [d], -w])
	indices = ((np.arange(n, dtype=np.int64)[:, None] + rel[None, :]) % n).astype(np.int32)
	data = np.broadcast_to(val, (n, len(rel))).copy()
	wrap = np.concatenate([np.arange(offs.max()), np.arange(n - offs.max(), n)])  # rows whose columns wrap around: sort them
	order = np.argsort(indices[wrap], axis=1)
	indices[wrap] = np.take_along_axis(indices[wrap], order, axis=1)
	data[wrap] = np.take_along_axis(data[wrap], order, axis=1)
	A = sp.csr_matrix((data.ravel(), indices.ravel(), np.arange(0, n * len(rel) + 1, len(rel), dtype=np.int32)), shape=(n, n))
	A.has_sorted_indices = True
	k = np.arange(n, dtype=np.float64)
	lam = np.full(n, d)
	for o, wo in zip(offs, w):
		lam -= 2.0 * wo * np.cos(2.0 * np.pi * ((k * o) % n) / n)
	return A, lam


def test_config5_eigencount_by_step_function_full_reorth_n1e7(eng):
	"""configs[4] as BASELINE.json words it: eigencount via the step function (src/primate/special.py:69-74) on an n = 1e7
	CSR with 15 nonzeros per row, k = 80, full reorthogonalisation (81 ring slots x 32 probes resident = 207 GB), one
	batch of one GPU's share of the 2048 probes. The operator is a symmetric circulant band, so the count of eigenvalues
	above a cut INSIDE the spectrum is known exactly: the estimate must lie within 3 standard errors plus the quadrature's
	own uncertainty - the Gauss weights of the two nodes next to the cut bracket the spectral measure there
	(Chebyshev-Markov-Stieltjes) - of it. Full and no reorthogonalisation give the same smooth-function rule sums."""
	n, k, P = 10_000_000, 80, 32
	A, lam = circulant_band(n)
	assert A.shape[0] == n and A.nnz == 15 * n and np.all(np.diff(A.indptr) == 15)
	cut = float(np.median(lam)) + 1e-3  # strictly inside the spectrum, on no eigenvalue cluster's edge
	exact = int(np.count_nonzero(lam >= cut))
	assert 0.4 * n < exact < 0.6 * n and lam.min() > 0.09
	op = eng.DeviceOperator(A)
	out = {}
	for orth, probes in ((k, P), (0, 8)):
		plan = eng.LanczosPlan(op, probes, k, orth)
		if orth == k:
			assert plan.workspace_bytes > 200e9
		plan.generate_probes("rademacher", seed=7)
		plan.run()
		q, nodes, weights = plan.quadrature("log", return_rule=True)
		np.testing.assert_allclose(weights.sum(axis=1), 1.0, atol=1e-12)
		assert nodes.min() > 0.09 and nodes.max() < lam.max() + 1e-9  # Ritz values inside the spectrum
		out[orth] = (q, plan.quadrature("step", c=cut), nodes, weights)
		np.testing.assert_allclose(plan.quadrature("numrank"), n, rtol=1e-12)  # all of the spectrum above 1e-6: that cut counts everything
		plan.close()
	np.testing.assert_allclose(out[k][0][:8], out[0][0], rtol=1e-9)
	counts, nodes, weights = out[k][1], out[k][2], out[k][3]
	est, stderr = counts.mean(), counts.std(ddof=1) / np.sqrt(P)
	## per probe: the two nodes next to the cut; the measure of [cut, inf) lies within their weights of the rule's sum
	j = np.array([np.searchsorted(nodes[i], cut) for i in range(P)])
	bracket = np.array([weights[i, max(j[i] - 1, 0)] + weights[i, min(j[i], k - 1)] for i in range(P)]) * n
	assert abs(est - exact) < 3 * stderr + bracket.mean(), (est, exact, stderr, bracket.mean())
	assert abs(est / exact - 1) < 0.05
	assert stderr < 0.01 * n  # full reorthogonalisation: the per-probe counts scatter by well under a per cent of n
