"""Diagonal estimators with the reference's signatures (src/primate/diagonal.py).

`diag(M)` for a `MatrixFunction` with a built-in spectral function and a fixed probe budget
(`converge="count"`) runs entirely on the device: f(A)v for a batch of probes from the retained
Lanczos basis, then the numer/denom/running-mean accumulation (`slq_diag_update`). Everything else
(plain matrices, adaptive stopping rules, callbacks) follows the reference loop on the host with the
operator product delegated to `A @ v`.

`xdiag(A, m)` is the exchangeable estimator, re-derived here as an average of leave-one-out estimates (`_exchangeable_diagonal`) and pinned by the
reference's own output (tests/golden: `xdiag_m40`).
"""

from __future__ import annotations

from typing import Callable, Optional, Union

import numpy as np

from .estimators import ConvergenceCriterion, CountCriterion, EstimatorResult, MeanEstimator, convergence_criterion
from .operators import MatrixFunction, is_valid_operator
from .random import isotropic


def diag(
	A,
	pdf: Union[str, Callable] = "rademacher",
	converge: Union[str, ConvergenceCriterion] = "tolerance",
	seed: Union[int, np.random.Generator, None] = None,
	full: bool = False,
	callback: Optional[Callable] = None,
	record: bool = False,
	batch: int = 64,
	**kwargs,
) -> Union[np.ndarray, tuple]:
	"""Girard-Hutchinson estimate of diag(A) (or diag f(A)); src/primate/diagonal.py:11-92.

	The returned estimate is the reference's: the running mean over probes of numer_t / denom_t with
	numer += (A v) * v, denom += v * v (diagonal.py:74-79). `batch` (extra argument) only sets how many
	probes the device path evaluates per run.
	"""
	f_dtype = is_valid_operator(A)
	N: int = A.shape[0]
	rng = np.random.default_rng(seed)
	pdf_fn = isotropic(pdf=pdf, seed=rng) if isinstance(pdf, str) else pdf
	estimator = MeanEstimator(dim=N, covariance=False, record=record)
	converge = convergence_criterion(converge, **kwargs)
	if np.prod(A.shape) == 0:
		return 0.0 if not full else (0.0, EstimatorResult())

	device_ok = (
		isinstance(A, MatrixFunction) and A._builtin is not None and isinstance(converge, CountCriterion)
		and callback is None and not record
	)  # fmt: skip
	if device_ok:
		from . import engine

		name, kw = A._builtin
		acc = engine.DiagAccumulator(N, ctx=A._op.ctx)
		done = 0
		while done < converge.count:
			m = min(batch, converge.count - done)
			V = np.asfortranarray(np.column_stack([pdf_fn(size=N).reshape(N, -1) for _ in range(m)]).astype(f_dtype, copy=False))
			plan = A._plan(m, True)
			plan.set_probes(V)
			plan.run(A._rtol)
			acc.update(plan, name, **kw)
			done += m
		numer, denom, running_mean, cnt = acc.get()
		acc.close()
		## fold into the estimator so `info.estimator` is populated like the reference's
		estimator._mean.mu = running_mean.astype(np.float64)
		estimator._mean.n = cnt
		estimator.n_samples = cnt
		if full:
			result = EstimatorResult(estimator, converge)
			result.estimate, result.nit = estimator.estimate, cnt
			result.info = {"numer": numer, "denom": denom}
			return estimator.estimate, result
		return estimator.estimate

	numer, denom = np.zeros(N, dtype=f_dtype), np.zeros(N, dtype=f_dtype)
	result = EstimatorResult(estimator, converge)
	while not converge(estimator):
		v = pdf_fn(size=N).astype(f_dtype, copy=False)
		u = np.asarray(A @ v).ravel()
		numer += u * v.ravel()
		denom += np.square(v.ravel())
		estimator.update(np.atleast_2d(numer / denom))
		if callback is not None:
			callback(result)
	if full or callback is not None:
		result.estimate, result.nit = estimator.estimate, len(estimator)
		return (estimator.estimate, result)
	return estimator.estimate


def _exchangeable_diagonal(W: np.ndarray, Y: np.ndarray, Q: np.ndarray, Z: np.ndarray, R: np.ndarray) -> np.ndarray:
	"""The XDiag estimate from one sketch (Epperly, Tropp & Webber, "XTrace: making the most of every sample", section on diagonals;
	what src/primate/diagonal.py:99-138 evaluates), written as the average of k leave-one-out estimates.

	W: the k probes; Y = A W = Q R; Z = A^T Q. For probe i let Q_(i) span the sketch WITHOUT column i: Q_(i) Q_(i)^T = Q (I - s_i s_i^T) Q^T with
	s_i the i-th row of R^{-1} scaled to unit length (the same downdate `trace._leave_one_out_estimates` uses). Estimate i is the
	diagonal of A on that span plus probe i's own Girard-Hutchinson term on what is left,

	    d_i = diag(Q_(i) Q_(i)^T A) + w_i * ((I - Q_(i) Q_(i)^T) A w_i),

	and d = mean_i d_i. With t_i = Q^T A w_i (column i of T = Z^T W):
	  * diag(Q_(i) Q_(i)^T A) = rowsum(Q * Z) - (Q s_i) * (Z s_i): the first part is common to all i, the second averages to rowsum((QS) * (ZS)) / k;
	  * (I - Q_(i) Q_(i)^T) A w_i = y_i - Q t_i + (Q s_i)(s_i . t_i): column i of the residual matrix below.
	Everything of size n is a product of n x k with k x k matrices."""
	from scipy.linalg import solve_triangular

	k = W.shape[1]
	R_inv = solve_triangular(R, np.eye(k))
	S = R_inv.T / np.linalg.norm(R_inv, axis=1)  # unit columns: the directions the k downdates remove
	T = Z.T @ W
	QS, ZS = Q @ S, Z @ S
	resid = Y - Q @ T + QS * np.einsum("ij,ij->j", S, T)  # column i: what probe i sees of A outside its own leave-one-out span
	on_span = np.einsum("ij,ij->i", Q, Z) - np.einsum("ij,ij->i", QS, ZS) / k
	off_span = np.einsum("ij,ij->i", W, resid) / k
	return on_span + off_span


def xdiag(A, m: Optional[int] = None, pdf: str = "sphere", seed: Union[int, np.random.Generator, None] = None) -> np.ndarray:
	"""Exchangeable diagonal estimator with the reference's signature and probe stream (src/primate/diagonal.py:99-138): about m / 2 products
	with A for the sketch and m / 2 with A^T for its image, then `_exchangeable_diagonal` (m = None: the full budget of 2 n).

	Over a `MatrixFunction` the two products are the hot path: each is ONE lock-step Lanczos batch of m / 2 columns on the GPU (f(A) W from the
	retained bases, then f(A) Q); the k x k algebra around them stays on the host like the reference's."""
	assert is_valid_operator(A) or isinstance(A, np.ndarray), "A must be a matrix or a linear operator"
	n = A.shape[0]
	budget = 2 * n if m is None else min(int(m) + int(m) % 2, 2 * n)  # (rounded up to an even count, at most 2 n: diagonal.py:106-107)
	k = budget // 2
	rng = np.random.default_rng(seed=seed)
	W = isotropic(pdf=pdf, seed=rng)(size=(n, k))
	Y = np.asarray(A @ W)
	Q, R = np.linalg.qr(Y, mode="reduced")
	Z = np.asarray(A.T @ Q)
	return _exchangeable_diagonal(W, Y, Q, Z, R)
