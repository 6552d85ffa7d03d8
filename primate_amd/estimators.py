"""Streaming estimators and stopping rules: the host-side bookkeeping of the Monte-Carlo drivers.

Same names and semantics as the reference (src/primate/stats.py:7-86, src/primate/estimators.py):
O(1) scalar work per batch, so it stays on the host. The batch-Welford merge implemented by
`Covariance.update` is also what combines per-GPU shards (primate_amd/distributed.py).
"""

from __future__ import annotations

import inspect
from dataclasses import dataclass, field
from typing import Callable, Iterable, Optional, Union

import numpy as np


class Mean:
	"""Running mean over batches (stats.py:7-38)."""

	def __init__(self, dim: int = 1):
		self.dim, self.n, self.mu = dim, 0, np.zeros(dim)

	def mean(self) -> Union[float, np.ndarray]:
		if self.n == 0:
			return np.nan
		return self.mu.item() if self.dim == 1 else self.mu

	__call__ = mean

	@staticmethod
	def _as_batch(X, dim):
		X = np.atleast_1d(X)
		X = X[:, None] if X.ndim == 1 else X
		assert X.shape[1] == dim, f"Expected shape (n, {dim}), got {X.shape}"
		return X

	def update(self, X) -> None:
		X = self._as_batch(X, self.dim)
		m = X.shape[0]
		delta = X.mean(axis=0) - self.mu
		self.n += m
		self.mu += (m / self.n) * delta


class Covariance(Mean):
	"""Batch Welford mean + scatter matrix (stats.py:41-88): for a batch of m samples with mean xb,
	mu += (m/N') (xb - mu);  S += Xc^T Xc + (N m / N') d d^T  with d = xb - mu_old."""

	def __init__(self, dim: int = 1):
		super().__init__(dim)
		self.S = np.zeros((dim, dim))

	def covariance(self, ddof: int = 1) -> Union[float, np.ndarray]:
		if (self.n - ddof) <= 0:
			return np.inf
		cov = self.S / (self.n - ddof)
		return cov.item() if self.dim == 1 else cov

	__call__ = covariance

	def update(self, X) -> None:
		X = self._as_batch(X, self.dim)
		m = X.shape[0]
		xb = X.mean(axis=0)
		delta = xb - self.mu
		new_n = self.n + m
		self.mu += (m / new_n) * delta
		Xc = X - xb
		shift = np.outer(delta, delta) if self.dim > 1 else (delta * delta)
		self.S += (Xc.T @ Xc) + (self.n * m / new_n) * shift
		self.n = new_n

	def merge(self, n: int, mu: np.ndarray, S: np.ndarray) -> None:
		"""Fold in another accumulator's sufficient statistics (count, mean, scatter): the same
		formula with the other shard playing the role of the batch. Used for multi-GPU merges."""
		if n == 0:
			return
		mu = np.atleast_1d(mu).astype(float)
		delta = mu - self.mu
		new_n = self.n + n
		shift = np.outer(delta, delta) if self.dim > 1 else (delta * delta)
		self.S += np.atleast_2d(S) + (self.n * n / new_n) * shift
		self.mu += (n / new_n) * delta
		self.n = new_n


class MeanEstimator:
	"""Sample-mean estimator with optional covariance tracking (estimators.py:102-146)."""

	def __init__(self, dim: int = 1, covariance: bool = False, record: bool = False) -> None:
		self.n_samples = 0
		self.delta = np.full(shape=dim, fill_value=np.inf)
		self.values = [] if record else None
		if covariance:
			self._cov = Covariance(dim=dim)
		else:
			self._mean = Mean(dim=dim)

	def __len__(self) -> int:
		return self.n_samples

	@property
	def _acc(self):
		return self._cov if hasattr(self, "_cov") else self._mean

	@property
	def mean(self) -> Union[float, np.ndarray]:
		if hasattr(self, "_cov"):
			mu = np.atleast_1d(self._cov.mean())
			return mu.item() if len(mu) == 1 else np.ravel(mu)
		return self._mean()

	def update(self, x) -> None:
		x = np.atleast_1d(x)
		x = x[:, None] if x.ndim == 1 else x
		old = self._acc.mu.copy()
		self._acc.update(x)
		self.delta = self._acc.mu - old
		self.n_samples += x.shape[0]
		if self.values is not None:
			self.values.extend(x)

	@property
	def estimate(self) -> Union[float, np.ndarray]:
		return self.mean


class ConvergenceCriterion:
	"""Lazily evaluated stopping rule, composable with |, & and ~ (estimators.py:56-76)."""

	def __init__(self, operation: Callable):
		assert callable(operation)
		self._operation = operation

	def __or__(self, other):
		return ConvergenceCriterion(lambda est: self(est) or other(est))

	def __and__(self, other):
		return ConvergenceCriterion(lambda est: self(est) and other(est))

	def __invert__(self):
		return ConvergenceCriterion(lambda est: not self(est))

	def __call__(self, est) -> bool:
		return self._operation(est)

	def message(self, est) -> str:
		return "Composite convergence criterion"


class Estimator:
	"""Interface of an updateable estimator (src/primate/estimators.py:35-53): a length (samples seen), an
	`update(x)` and an `estimate`; `values` optionally records the samples, `delta` is the last change."""

	n_samples: int = 0
	values = None
	delta = np.inf

	def __len__(self) -> int:
		return self.n_samples

	def update(self, x): ...

	@property
	def estimate(self): ...


class ControlVariableEstimator(MeanEstimator):
	"""Mean of a scalar response corrected by control variates with known expectations `ecv`
	(src/primate/estimators.py:148-196): samples are rows [y, c_1, ..., c_m]; the estimate is
	mean(y) - alpha . (mean(c) - ecv) with alpha = Cov(c)^-1 Cov(c, y) re-estimated at every update unless
	given."""

	def __init__(self, ecv, alpha=None, record: bool = False):
		ecv = np.atleast_1d(ecv).astype(float).ravel()
		super().__init__(len(ecv), covariance=False, record=record)
		if alpha is not None:
			alpha = np.atleast_1d(alpha).astype(float).ravel()
			assert len(alpha) == len(ecv), "Coefficients alpha must have same length as the control variables."
		self.alpha, self.ecv = alpha, ecv
		self.cov = Covariance(dim=len(ecv) + 1)
		self._fit_alpha = alpha is None

	def update(self, samples):
		self.cov.update(np.atleast_1d(samples))
		self.n_samples = self.cov.n
		if self._fit_alpha:
			C = np.atleast_2d(self.cov(ddof=1))
			self.alpha = np.atleast_1d(C[0, 1] / C[1, 1]) if self.cov.dim == 2 else np.linalg.solve(C[1:, 1:], C[1:, 0])
		return self

	@property
	def estimate(self):
		if self.n_samples == 0:
			return np.nan
		return float(self.cov.mu[0] - np.dot(self.alpha, self.cov.mu[1:] - self.ecv))


def arr_summary(x) -> str:
	"""Short printable form of an estimate (src/primate/estimators.py:18-32); used in the criteria messages."""
	return _summary(x)


def _summary(x) -> str:
	if x is None:
		return "None"
	x = np.atleast_1d(x)
	if len(x) == 1:
		return f"{x.item():.3f}"
	with np.printoptions(precision=2, suppress=True, threshold=3, floatmode="fixed"):
		return np.array2string(x, separator=",")


class CountCriterion(ConvergenceCriterion):
	"""True once at least `count` samples were seen (estimators.py:205-218)."""

	def __init__(self, count: int):
		self.count = count

	def __call__(self, est) -> bool:
		return len(est) >= self.count

	def message(self, est) -> str:
		return f"Est: {_summary(np.array(est.estimate))} (#S:{ len(est) })"


class ToleranceCriterion(ConvergenceCriterion):
	"""||last change of the estimate|| < atol or < rtol ||estimate|| (estimators.py:221-243)."""

	def __init__(self, rtol: float = 0.01, atol: float = 1.49e-08, ord=2.0) -> None:
		self.rtol, self.atol, self.ord = rtol, atol, ord

	def __call__(self, est) -> bool:
		if est.estimate is None:
			return False
		err = np.linalg.norm(est.delta, ord=self.ord)
		return bool(err < self.atol or err < self.rtol * np.linalg.norm(np.atleast_1d(est.estimate), ord=self.ord))

	def message(self, est) -> str:
		return f"Est: {_summary(est.estimate)}(atol={self.atol:3f}, rtol={self.rtol:3f}, #S:{ len(est) })"


class ConfidenceCriterion(ConvergenceCriterion):
	"""CLT margin of error <= atol or relative standard error <= rtol (estimators.py:246-299):
	Student-t score below 30 samples, normal score afterwards; never before 3 samples."""

	def __init__(self, confidence: float = 0.95, atol: float = 0.00, rtol: float = 0.01) -> None:
		import scipy.special
		import scipy.stats

		assert 0 < confidence and confidence < 1, "Confidence must be in (0, 1)"
		self.atol = 0.0 if atol is None else atol
		self.rtol = 0.0 if rtol is None else rtol
		self.z = np.sqrt(2.0) * scipy.special.erfinv(confidence)
		self.t_scores = scipy.stats.t.ppf((confidence + 1.0) / 2.0, df=np.arange(30) + 1)
		self.confidence = confidence

	def _error(self, est: MeanEstimator) -> tuple:
		if est.n_samples < 3:
			return (np.inf, np.inf)
		std_err = est._cov.covariance() ** 0.5 / np.sqrt(est.n_samples)
		score = self.t_scores[est.n_samples] if est.n_samples < 30 else self.z
		return (score * std_err, abs(std_err / est.estimate))

	def __call__(self, est) -> bool:
		assert isinstance(est, MeanEstimator), "Must be a mean estimator"
		moe, rerr = self._error(est)
		return moe <= self.atol or rerr <= self.rtol

	def message(self, est) -> str:
		moe, _ = self._error(est)
		return f"Est: {_summary(est.estimate)} +/- {moe:.3f} ({self.confidence*100:.0f}% CI, #S:{ len(est) })"


class KneeCriterion(ConvergenceCriterion):
	"""Kneedle-style detection of the knee of the cumulative |change| of the running sample mean
	(estimators.py:302-333). Needs an estimator created with record=True."""

	def __init__(self, S: float = 1.0) -> None:
		self.S = S

	def __call__(self, est) -> bool:
		if est.values is None or len(est.values) < 3:
			return False
		vals = np.array(est.values).ravel()
		running = vals / np.arange(1, len(vals) + 1)
		y = np.cumsum(np.abs(np.diff(running)))
		y_norm = (y - y.min()) / (y.max() - y.min())
		diff_curve = y_norm - np.linspace(0, 1, len(y))
		peak = diff_curve[np.argmax(diff_curve)]
		threshold = peak - (self.S / (len(y) - 1))
		return bool(peak > threshold and diff_curve[-1] < threshold)

	def message(self, est) -> str:
		return f"Est: {_summary(est.estimate)} (#S:{ len(est) }, S={self.S:3f})"


CRITERIA = {"count": CountCriterion, "tolerance": ToleranceCriterion, "confidence": ConfidenceCriterion, "knee": KneeCriterion}


def convergence_criterion(criterion: Union[str, ConvergenceCriterion], **kwargs) -> ConvergenceCriterion:
	"""Name or instance -> criterion; kwargs not accepted by the criterion are ignored
	(estimators.py:344-354 with typing.restrict_kwargs)."""
	if isinstance(criterion, ConvergenceCriterion):
		return criterion
	assert isinstance(criterion, str) and criterion.lower() in CRITERIA, f"Invalid criterion {criterion}"
	cls = CRITERIA[criterion.lower()]
	ok = set(inspect.signature(cls).parameters)
	return cls(**{k: v for k, v in kwargs.items() if k in ok})


@dataclass
class EstimatorResult:
	"""What `full=True` returns next to the estimate (estimators.py:79-91)."""

	estimator: Optional[MeanEstimator] = None
	criterion: Union[ConvergenceCriterion, str, None] = None
	estimate: Union[float, np.ndarray] = 0.0
	message: str = ""
	nit: int = 0
	info: dict = field(default_factory=dict)

	def __iter__(self) -> Iterable:
		return iter((self.estimator, self.criterion, self.estimate, self.message, self.nit, self.info))
