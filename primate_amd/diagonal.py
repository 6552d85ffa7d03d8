"""Diagonal estimators with the reference's signatures (src/primate/diagonal.py).

`diag(M)` for a `MatrixFunction` with a built-in spectral function and a fixed probe budget
(`converge="count"`) runs entirely on the device: f(A)v for a batch of probes from the retained
Lanczos basis, then the numer/denom/running-mean accumulation (`slq_diag_update`). Everything else
(plain matrices, adaptive stopping rules, callbacks) follows the reference loop on the host with the
operator product delegated to `A @ v`.
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
