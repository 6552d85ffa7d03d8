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
	"""
	f_dtype = is_valid_operator(A)
	N: int = A.shape[0]
	rng = np.random.default_rng(seed)
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
			v = pdf(size=(N, batch)).astype(f_dtype)
			estimator.update(quad_form(v))
			callback(result)
		result.message = converge.message(estimator)
		result.estimate, result.nit = estimator.estimate, len(estimator)
		return (estimator.estimate, result)

	## sample-at-a-time semantics, batch-at-a-time evaluation
	while not converge(estimator):
		m = batch
		if isinstance(converge, CountCriterion):
			m = max(1, min(batch if not hasattr(A, "quad") else max(batch, 256), converge.count - len(estimator)))
		ys = np.atleast_1d(quad_form(pdf(size=(N, m)).astype(f_dtype)))
		for y in ys:
			estimator.update(y)
			if converge(estimator):
				break
	return estimator.estimate
