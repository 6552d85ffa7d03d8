"""Trace estimators with the reference's signatures (src/primate/trace.py). `hutch` is the SLQ hot
path driver: probes are drawn with the reference's NumPy stream (parity) or on the device
(`pdf="device:<name>"`, throughput), evaluated in lock-step batches by libslq, and folded into the
streaming estimator one sample at a time exactly as the reference's loop does.
"""

from __future__ import annotations

from typing import Callable, Optional, Union

import numpy as np

from .estimators import (
	ConfidenceCriterion,
	ConvergenceCriterion,
	CountCriterion,
	EstimatorResult,
	MeanEstimator,
	convergence_criterion,
)
from .operators import is_valid_operator
from .random import isotropic


def _quad_form(A) -> Callable:
	if hasattr(A, "quad"):
		return lambda v: A.quad(v)
	return lambda v: np.einsum("...i,...i->...", v.T, (A @ v).T)


def hutch(
	A,
	batch: int = 32,
	pdf: Union[str, Callable] = "rademacher",
	converge: Union[str, ConvergenceCriterion] = "default",
	seed: Union[int, np.random.Generator, None] = None,
	full: bool = False,
	callback: Optional[Callable] = None,
	**kwargs,
) -> Union[float, tuple]:
	"""Girard-Hutchinson estimate of tr(A) (or tr f(A) for a MatrixFunction); src/primate/trace.py:34-116.

	Same arguments and defaults as the reference. Two behaviours are preserved on purpose:
	  * with neither `full` nor `callback` the reference draws ONE probe per iteration (trace.py:114-115)
	    and tests the stopping rule after every sample. Here the probes of up to `batch` iterations are
	    drawn in one (N, m) call — identical values, because the F-ordered draw equals sequential
	    column draws (reference tests/test_random.py:23-39) — evaluated in one device run, and fed to
	    the estimator one by one, stopping at the same sample the reference would stop at;
	  * with `full`/`callback`, whole batches are drawn and folded at once (trace.py:104-110).
	One observable difference follows from the first point: the estimate and the stopping sample are the reference's, but
	a caller-supplied `np.random.Generator` is left further along its stream than the reference leaves it - up to
	`batch` - 1 probes (max(batch, 256) - 1 for a MatrixFunction under a count criterion) drawn past the stopping sample
	are discarded (and, for composite criteria, evaluated for nothing). Code that keeps drawing from the same generator
	after `hutch` returns sees different numbers than with the reference; pass a seed, or a generator of its own.
	"""
	f_dtype = is_valid_operator(A)
	N: int = A.shape[0]
	rng = np.random.default_rng(seed)
	## pdf="device:<name>" (extension): probes drawn on the GPU, ids 0, 1, 2, ... of the Philox stream `seed`
	dev_pdf = pdf[len("device:"):] if isinstance(pdf, str) and pdf.startswith("device:") else None
	if dev_pdf is not None:
		assert hasattr(A, "quad_generated"), "device-drawn probes need a MatrixFunction (primate_amd.operators)"
		dev_seed = int(seed) if isinstance(seed, (int, np.integer)) else int(rng.integers(0, 2**62))
		drawn = [0]

		def quad_next(m: int) -> np.ndarray:
			y = A.quad_generated(m, dev_pdf, dev_seed, drawn[0])
			drawn[0] += m
			return y

		pdf = None
	else:
		pdf = isotropic(pdf=pdf, seed=rng) if isinstance(pdf, str) else pdf
	estimator = MeanEstimator(covariance=True, record=kwargs.pop("record", False))
	if isinstance(converge, str) and converge == "default":
		converge = CountCriterion(count=200) | ConfidenceCriterion(confidence=0.95, atol=1.0, rtol=0.0)
	else:
		converge = convergence_criterion(converge, **kwargs)
	quad_form = _quad_form(A)
	if np.prod(A.shape) == 0:
		return 0.0 if not full else (0.0, EstimatorResult(estimator, converge))

	if full or callback is not None:
		result = EstimatorResult(estimator, converge)
		callback = (lambda x: x) if callback is None else callback
		while not converge(estimator):
			estimator.update(quad_next(batch) if dev_pdf is not None else quad_form(pdf(size=(N, batch)).astype(f_dtype, copy=False)))
			callback(result)
		result.message = converge.message(estimator)
		result.estimate, result.nit = estimator.estimate, len(estimator)
		return (estimator.estimate, result)

	## sample-at-a-time semantics, batch-at-a-time evaluation
	while not converge(estimator):
		m = batch
		if isinstance(converge, CountCriterion):
			m = max(1, min(batch if not hasattr(A, "quad") else max(batch, 256), converge.count - len(estimator)))
		ys = np.atleast_1d(quad_next(m) if dev_pdf is not None else quad_form(pdf(size=(N, m)).astype(f_dtype, copy=False)))
		for y in ys:
			estimator.update(y)
			if converge(estimator):
				break
	return estimator.estimate


def hutchpp(
	A,
	m: Optional[int] = None,
	batch: int = 32,
	mode: str = "reduced",
	pdf: Union[str, Callable] = "rademacher",
	seed: Union[int, np.random.Generator, None] = None,
	full: bool = False,
) -> Union[float, tuple]:
	"""Hutch++ (Meyer et al.): trace of the rank-nb sketch plus a Hutchinson estimate on its
	complement; src/primate/trace.py:119-182. `m` = sketch size (default n // 3). For a
	`MatrixFunction`, the products A @ W (nb columns at once) and the nb quadratic forms run as
	lock-step device batches."""
	f_dtype = is_valid_operator(A)
	N: int = A.shape[0]
	rng = np.random.default_rng(seed)
	draw = isotropic(pdf=pdf, seed=rng)
	quad_form = _quad_form(A)
	if np.prod(A.shape) == 0:
		return 0.0 if not full else (0.0, EstimatorResult())
	nb = (N // 3) if m is None else m
	nb += nb % 3
	from .operators import MatrixFunction

	if isinstance(A, MatrixFunction) and A._builtin is not None and A.dtype == np.float64 and not A._stale_ring and 0 < nb <= N:
		rng_ests, defl_ests = _hutchpp_device(A, nb, mode, draw)
		est = np.sum(rng_ests) + (1 / nb) * np.sum(defl_ests)
		if not full:
			return est
		result = EstimatorResult()
		result.estimate, result.nit = est, 2 * nb
		result.samples = np.concatenate([np.ravel(rng_ests), np.ravel(defl_ests)])
		return result.estimate, result
	W = draw(size=(N, nb)).astype(f_dtype, copy=False)
	Q = np.linalg.qr(A @ W, mode="reduced")[0]
	if mode == "full":
		rng_ests = np.einsum("...i,...i->...", A @ Q, Q)
	elif hasattr(A, "quad"):
		rng_ests = np.atleast_1d(A.quad(Q))  # all nb quadratic forms in one device run
	else:
		rng_ests = np.array([quad_form(q) for q in Q.T])
	tr_rng = np.sum(rng_ests)
	G = draw(size=(N, nb)).astype(f_dtype, copy=False)
	G -= Q @ (Q.T @ G)
	defl_ests = np.einsum("...i,...i->...", A @ G, G)
	tr_defl = (1 / nb) * np.sum(defl_ests)
	if not full:
		return tr_rng + tr_defl
	result = EstimatorResult()
	result.estimate = tr_rng + tr_defl
	result.nit = 2 * nb
	result.samples = np.concatenate([np.ravel(rng_ests), np.ravel(defl_ests)])
	return result.estimate, result


def _hutchpp_device(A, nb: int, mode: str, draw: Callable, chunk: int = 128) -> tuple:
	"""The body of `hutchpp` for a device `MatrixFunction` with every n x nb matrix resident in HBM
	(src/primate/trace.py:160-176): Y = f(A) W by lock-step Lanczos batches, Q by CholeskyQR2 on the matrix
	cores (Householder on the host if the sketch is numerically rank deficient), the nb quadratic forms of Q
	by Lanczos quadrature (mode "reduced", as `A.quad`) or as diag(Q^T f(A) Q) (mode "full"), the deflation
	G -= Q (Q^T G) as two tall-skinny products, and diag(G^T f(A) G). Probes come from the same NumPy stream
	as the host path, so the two agree to rounding."""
	from scipy.linalg import cholesky, solve_triangular

	from . import engine

	n = A.shape[0]
	name, kw = A._builtin
	ctx = A._op.ctx
	Wd, Qd, Zd = (engine.DeviceMatrix(n, nb, ctx=ctx) for _ in range(3))

	def apply_fun(src, dst):
		for c in range(0, nb, chunk):
			ns = min(chunk, nb - c)
			plan = A._plan(ns, True)
			plan.set_probes_device(src.col_ptr(c))
			plan.run(A._rtol)
			plan.fun_action_into(dst, c, name, **kw)

	def column_dots(X, Y) -> np.ndarray:  # diag(X^T Y), block by block
		out = np.empty(nb)
		for c in range(0, nb, chunk):
			ns = min(chunk, nb - c)
			out[c : c + ns] = np.diag(X.tn(c, ns, Y, c, ns))
		return out

	try:
		Wd.set(0, draw(size=(n, nb)))
		apply_fun(Wd, Zd)  # Y = f(A) W
		try:
			R1 = cholesky(Zd.tn(0, nb, Zd, 0, nb), lower=False)
			Wd.add_product(0, Zd, 0, solve_triangular(R1, np.eye(nb)), alpha=1.0, beta=0.0)  # W is free now
			R2 = cholesky(Wd.tn(0, nb, Wd, 0, nb), lower=False)
			Qd.add_product(0, Wd, 0, solve_triangular(R2, np.eye(nb)), alpha=1.0, beta=0.0)
		except np.linalg.LinAlgError:
			Qd.set(0, np.linalg.qr(Zd.get(0, nb), mode="reduced")[0])
		if mode == "full":
			apply_fun(Qd, Zd)
			rng_ests = column_dots(Zd, Qd)
		else:
			rng_ests = np.empty(nb)
			for c in range(0, nb, chunk):
				ns = min(chunk, nb - c)
				plan = A._plan(ns, False)
				plan.set_probes_device(Qd.col_ptr(c))
				plan.run(A._rtol)
				rng_ests[c : c + ns] = plan.quadrature(name, **kw)
		Wd.set(0, draw(size=(n, nb)))  # G
		Wd.add_product(0, Qd, 0, Qd.tn(0, nb, Wd, 0, nb), alpha=-1.0, beta=1.0)
		apply_fun(Wd, Zd)
		defl_ests = column_dots(Zd, Wd)
	finally:
		for d in (Wd, Qd, Zd):
			d.close()
	return rng_ests, defl_ests


def _leave_one_out_estimates(n: int, QtW: np.ndarray, QtZ: np.ndarray, ZtW: np.ndarray, R: np.ndarray, R_inv: np.ndarray, pdf) -> np.ndarray:
	"""XTrace's m exchangeable estimates from m x m summaries only (QtW = Q^T W, QtZ = Q^T A Q, ZtW = (A Q)^T W, A W = Q R).

	Estimate i is "trace of A on span(Q) with probe i's direction s_i taken out" plus "Hutchinson's estimate of probe i on
	what is left" (Epperly, Tropp & Webber, XTrace; what src/primate/trace.py:199-227 evaluates). s_i is the i-th row of
	R^{-1} scaled to unit length: removing column i of W from the sketch downdates Q by exactly that direction, so every
	term below is a column-wise product of m x m matrices - nothing of size n is touched. Returns an (m, 1) column."""
	m = QtW.shape[0]
	colsum = lambda X, Y: np.einsum("ij,ij->j", X, Y)  # noqa: E731  (x_i . y_i) for every column i
	S = R_inv.T / np.linalg.norm(R_inv, axis=1)  # downdate directions, unit columns
	s_w = colsum(S, QtW)  # s_i . (Q^T w_i)
	s_h_s = colsum(S, QtZ @ S)  # s_i^T (Q^T A Q) s_i
	G = QtZ @ QtW
	## part 1: tr(Q_(i)^T A Q_(i)) with Q_(i) = Q minus the direction s_i
	on_span = np.trace(QtZ) - s_h_s
	## part 2: w_i^T (I - Q_(i) Q_(i)^T) A (I - Q_(i) Q_(i)^T) w_i, expanded in the summaries (the A w_i = Q r_i terms cancel
	## against R; what survives pairs s_i with the residuals R - G and ZtW - QtZ^T QtW)
	off_span = colsum(QtW, G) - colsum(ZtW, QtW) + s_w * (colsum(S, R - G) + colsum(ZtW - QtZ.T @ QtW, S) + s_w * s_h_s)
	if pdf == "sphere":
		## probes on the sphere of radius sqrt(n): the projected probe is rescaled to the complement's dimension n - m + 1
		weight = (n - m + 1) / (n - np.linalg.norm(QtW, axis=0) ** 2 + (s_w * np.linalg.norm(S, axis=0)) ** 2)
	else:
		weight = 1.0
	return (on_span + weight * off_span)[:, None]


def _xtrace(W: np.ndarray, Z: np.ndarray, Q: np.ndarray, R: np.ndarray, R_inv: np.ndarray, pdf) -> np.ndarray:
	"""XTrace estimates for the m probes in W; Z = A Q, Q R = A W, R_inv = R^{-1} (trace.py:185-227)."""
	return _leave_one_out_estimates(W.shape[0], Q.T @ W, Q.T @ Z, Z.T @ W, R, R_inv, pdf)


def _qr_append_block(Q: np.ndarray, R: np.ndarray, R_inv: np.ndarray, Y: np.ndarray) -> tuple:
	"""QR factors of [Q R, Y] from those of Q R, a whole block at a time.

	The reference appends the ns new columns one by one with scipy's `qr_insert` (Givens, BLAS-2:
	O(n m) memory traffic per column, trace.py:298-300). Here the block is orthogonalised against Q
	with two classical Gram-Schmidt passes (BLAS-3) and factored on its own; R and R^{-1} grow by a
	block column. The factorisation is the same up to the signs of R's diagonal, to which the XTrace
	estimates are invariant (every term of `_xtrace` pairs a column of Q or S with itself)."""
	m, ns = Q.shape[1], Y.shape[1]
	C = np.zeros((m, ns))
	for _ in range(2):  # "twice is enough"
		Ci = Q.T @ Y
		Y = Y - Q @ Ci
		C += Ci
	Qn, Rn = np.linalg.qr(Y, mode="reduced")
	R_new = np.zeros((m + ns, m + ns))
	R_new[:m, :m], R_new[:m, m:], R_new[m:, m:] = R, C, Rn
	from scipy.linalg import solve_triangular

	Rn_inv = solve_triangular(Rn, np.eye(ns))
	Ri_new = np.zeros((m + ns, m + ns))
	Ri_new[:m, :m], Ri_new[m:, m:] = R_inv, Rn_inv
	Ri_new[:m, m:] = -R_inv @ C @ Rn_inv
	return np.c_[Q, Qn], R_new, Ri_new


def _xtrace_device(A, batch: int, draw: Callable, budget: int, record: bool, callback: Callable, result: EstimatorResult, device_rng=None, shard=None):
	"""xtrace for a device `MatrixFunction` with everything n-sized resident on the GPU: the sketches
	W, Q, Z are column-major device matrices; f(A)·(new probes) and f(A)·(new Q columns) are lock-step
	Lanczos batches whose outputs never leave HBM; the block Gram-Schmidt + CholeskyQR2 and the
	three m x m summaries are fp64 MFMA products (slq_dmat_gemm_tn / _nn). Only m x m matrices and the
	host-drawn probes cross PCIe. Same estimator as the host path (sign-of-R invariance as in
	`_qr_append_block`)."""
	from scipy.linalg import cholesky, solve_triangular

	from . import engine

	n = A.shape[0]
	name, kw = A._builtin
	ctx = A._op.ctx
	P = min(budget, n)
	Wd, Qd, Zd = (engine.DeviceMatrix(n, P, ctx=ctx) for _ in range(3))
	Yd, Td = engine.DeviceMatrix(n, batch, ctx=ctx), engine.DeviceMatrix(n, batch, ctx=ctx)
	R, R_inv = np.zeros((0, 0)), np.zeros((0, 0))
	estimator = MeanEstimator(record=record)
	m = 0

	def apply_fun_local(src: "engine.DeviceMatrix", c0: int, ns: int, dst: "engine.DeviceMatrix", o0: int):
		plan = A._plan(ns, True)
		plan.set_probes_device(src.col_ptr(c0))
		plan.run(A._rtol)
		plan.fun_action_into(dst, o0, name, **kw)

	Sd = Gd = None
	if shard is not None:
		rank, world, allgather = shard
		cmax = -(-int(batch) // world)  # columns per rank and block, padded to equal shards for the all-gather
		Sd, Gd = engine.DeviceMatrix(n, cmax, ctx=ctx), engine.DeviceMatrix(n, cmax * world, ctx=ctx)

	def apply_fun(src: "engine.DeviceMatrix", c0: int, ns: int, dst: "engine.DeviceMatrix", o0: int):
		if shard is None:
			return apply_fun_local(src, c0, ns, dst, o0)
		base, rem = divmod(ns, world)
		lo = rank * base + min(rank, rem)
		nloc = base + (1 if rank < rem else 0)
		if nloc > 0:
			apply_fun_local(src, c0 + lo, nloc, Sd, 0)
		allgather(Sd, cmax, Gd)
		for r in range(world):  # rank r's valid columns are the first base (+1) of its cmax-wide slab
			rl, rn = r * base + min(r, rem), base + (1 if r < rem else 0)
			if rn > 0:
				dst.copy_from(o0 + rl, Gd, r * cmax, rn)

	try:
		while m < P:
			ns = min(int(batch), P - m)
			if device_rng is None or shard is not None:
				if device_rng is None:
					Wd.set(m, draw(size=(n, ns)))
				else:
					plan = A._plan(ns, True)
					plan.generate_probes(device_rng[0], seed=device_rng[1], probe_offset=m)
					plan.get_probes_into(Wd, m)
				apply_fun(Wd, m, ns, Yd, 0)
			else:
				plan = A._plan(ns, True)
				plan.generate_probes(device_rng[0], seed=device_rng[1], probe_offset=m)
				plan.get_probes_into(Wd, m)
				plan.run(A._rtol)
				plan.fun_action_into(Yd, 0, name, **kw)
			## block Gram-Schmidt against the existing Q, twice
			Cm = np.zeros((m, ns))
			if m > 0:
				for _ in range(2):
					Ci = Qd.tn(0, m, Yd, 0, ns)
					Yd.add_product(0, Qd, 0, Ci, alpha=-1.0, beta=1.0)
					Cm += Ci
			## CholeskyQR2 of the remainder: Y -> T = Y R1^{-1} -> Q_new = T R2^{-1}; R_n = R2 R1
			try:
				R1 = cholesky(Yd.tn(0, ns, Yd, 0, ns), lower=False)
				Td.add_product(0, Yd, 0, solve_triangular(R1, np.eye(ns)), alpha=1.0, beta=0.0)
				R2 = cholesky(Td.tn(0, ns, Td, 0, ns), lower=False)
				Qd.add_product(m, Td, 0, solve_triangular(R2, np.eye(ns)), alpha=1.0, beta=0.0)
				Rn = R2 @ R1
			except np.linalg.LinAlgError:
				## numerically rank-deficient block: Householder QR of this block on the host
				Qn, Rn = np.linalg.qr(Yd.get(0, ns), mode="reduced")
				Qd.set(m, Qn)
			R_new = np.zeros((m + ns, m + ns))
			R_new[:m, :m], R_new[:m, m:], R_new[m:, m:] = R, Cm, Rn
			Rn_inv = solve_triangular(Rn, np.eye(ns))
			Ri_new = np.zeros((m + ns, m + ns))
			Ri_new[:m, :m], Ri_new[m:, m:] = R_inv, Rn_inv
			Ri_new[:m, m:] = -R_inv @ Cm @ Rn_inv
			R, R_inv = R_new, Ri_new
			apply_fun(Qd, m, ns, Zd, m)
			m += ns
			t_samples = _leave_one_out_estimates(n, Qd.tn(0, m, Wd, 0, m), Qd.tn(0, m, Zd, 0, m), Zd.tn(0, m, Wd, 0, m), R, R_inv, None)
			estimator = MeanEstimator(record=record)
			estimator.update(t_samples.ravel())
			result.estimator, result.estimate, result.nit = estimator, estimator.estimate, m
			callback(result)
	finally:
		for d in (Wd, Qd, Zd, Yd, Td, Sd, Gd):
			if d is not None:
				d.close()
	return result


def xtrace(
	A,
	batch: int = 32,
	pdf: Union[str, Callable] = "sphere",
	converge: Union[str, ConvergenceCriterion] = "default",
	seed: Union[int, np.random.Generator, None] = None,
	full: bool = False,
	callback: Optional[Callable] = None,
	**kwargs,
) -> Union[float, tuple]:
	"""XTrace estimator; src/primate/trace.py:233-315. Per batch: ns new probes, Y = A N, column-wise
	QR insertion, Z = [Z, A Q_new], leave-one-out estimates. For a `MatrixFunction` both products are
	lock-step device batches of ns columns.

	Stopping: the reference overwrites the caller's criterion with CountCriterion(n) in both
	branches (trace.py:271-275), i.e. it always samples n probes. That behaviour is the default here
	too; pass `count=` (extra keyword) to stop after that many probes instead (BASELINE.json
	configs[2] uses 512).
	"""
	assert batch >= 1, "Batch size must be positive."
	n = A.shape[0]
	callback = (lambda result: ...) if not callable(callback) else callback
	record = kwargs.pop("record", False)
	budget = int(kwargs.pop("count", n)) if isinstance(converge, str) else n
	estimator = MeanEstimator(record=record)
	stop = CountCriterion(count=min(budget, n))
	W = np.zeros(shape=(n, 0), order="F")
	Z = np.zeros(shape=(n, 0), order="F")
	Q, R = np.linalg.qr(np.zeros(shape=(n, 0)), mode="reduced")
	R_inv = np.zeros(shape=(0, 0))
	result = EstimatorResult()
	rng = np.random.default_rng(seed)
	draw = isotropic(pdf=pdf, seed=rng) if isinstance(pdf, str) else pdf
	## extra keyword: draw the probes on the device too (device path only; a different, equally valid stream)
	device_rng = (pdf, int(seed) if isinstance(seed, (int, np.integer)) else int(rng.integers(0, 2**62))) if (kwargs.pop("device_rng", False) and isinstance(pdf, str)) else None
	## Parity note: the reference rebinds `pdf` to the sampler closure before handing it to _xtrace
	## (trace.py:295 then :305), so its `pdf == "sphere"` test (trace.py:207) is never true and the
	## sphere rescaling is never applied. Same here: results match the reference for every pdf.
	pdf_name = None
	from .operators import MatrixFunction

	if (
		isinstance(A, MatrixFunction) and A._builtin is not None and A.dtype == np.float64 and not A._stale_ring
		and kwargs.pop("device", True)
	):  # fmt: skip
		_xtrace_device(A, batch, draw, stop.count, record, callback, result, device_rng, kwargs.pop("_shard", None))
		result.criterion = stop
		return (result.estimate, result) if full else result.estimate
	while not stop(estimator):
		ns = min(A.shape[1] - W.shape[1], int(batch), stop.count - W.shape[1])
		Nw = draw(size=(n, ns))
		Y = np.asarray(A @ Nw)
		Q, R, R_inv = _qr_append_block(Q, R, R_inv, Y)
		W = np.c_[W, Nw]
		Z = np.c_[Z, np.asarray(A @ Q[:, -ns:])]
		t_samples = _xtrace(W, Z, Q, R, R_inv, pdf_name)
		estimator = MeanEstimator(record=record)
		estimator.update(t_samples.ravel())
		result.estimator, result.estimate, result.nit = estimator, estimator.estimate, W.shape[1]
		callback(result)
	result.criterion = stop
	return (result.estimate, result) if full else result.estimate
