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
	"""Regression-adjusted mean (control variates; counterpart of src/primate/estimators.py:148-196). A sample is a row
	[y, c_1 .. c_m]; E[c] = `ecv` is known. One joint scatter accumulator over the m + 1 columns is all the state; the
	coefficient vector is whatever was given, or else the least-squares slope of y on c, solved from the accumulated
	covariance blocks whenever it is asked for: a = Cov(c, c)^-1 Cov(c, y). estimate = mean(y) - a . (mean(c) - ecv)."""

	def __init__(self, ecv, alpha=None, record: bool = False):
		known = np.asarray(ecv, dtype=float).reshape(-1)
		super().__init__(known.size, covariance=False, record=record)
		self.ecv = known
		self._given = None
		if alpha is not None:
			self._given = np.asarray(alpha, dtype=float).reshape(-1)
			assert self._given.size == known.size, "Coefficients alpha must have same length as the control variables."
		self.cov = Covariance(dim=known.size + 1)

	@property
	def alpha(self):
		if self._given is not None:
			return self._given
		if self.cov.n < 2:
			return None
		C = np.atleast_2d(self.cov.covariance(ddof=1))
		cc, cy = C[1:, 1:], C[1:, 0]
		return cy / cc[0, 0] if cy.size == 1 else np.linalg.solve(cc, cy)

	def update(self, samples):
		self.cov.update(np.atleast_1d(samples))
		self.n_samples = self.cov.n
		return self

	@property
	def estimate(self):
		if self.n_samples == 0:
			return np.nan
		y_bar, c_bar = self.cov.mu[0], self.cov.mu[1:]
		return float(y_bar - self.alpha @ (c_bar - self.ecv))


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
	"""Stop when the two-sided `confidence` interval of the sample mean is narrow enough: half-width <= atol, or standard error
	of the mean <= rtol |mean| (counterpart of src/primate/estimators.py:246-299). The critical value is Student's t while
	fewer than 30 samples were seen - with n + 1 degrees of freedom at n samples, the reference's table as it is indexed -
	and the normal quantile from there on; below 3 samples nothing is decided."""

	_SMALL = 30  # samples below which the t table is consulted

	def __init__(self, confidence: float = 0.95, atol: float = 0.00, rtol: float = 0.01) -> None:
		from scipy.stats import norm, t as student

		assert 0 < confidence and confidence < 1, "Confidence must be in (0, 1)"
		self.confidence = confidence
		self.atol, self.rtol = (atol or 0.0), (rtol or 0.0)
		upper = 0.5 * (1.0 + confidence)  # two-sided interval -> upper-tail quantile
		self.z = float(norm.ppf(upper))
		self.t_scores = student.ppf(upper, df=1 + np.arange(self._SMALL))

	def critical_value(self, n: int) -> float:
		return float(self.t_scores[n]) if n < self._SMALL else self.z

	def _error(self, est: MeanEstimator) -> tuple:
		"""(half-width of the interval, standard error relative to |mean|)."""
		n = est.n_samples
		if n < 3:
			return (np.inf, np.inf)
		sem = float(np.sqrt(est._cov.covariance(ddof=1) / n))
		return (self.critical_value(n) * sem, abs(sem / est.estimate))

	def __call__(self, est) -> bool:
		assert isinstance(est, MeanEstimator), "Must be a mean estimator"
		half_width, rel = self._error(est)
		return half_width <= self.atol or rel <= self.rtol

	def message(self, est) -> str:
		return f"Est: {_summary(est.estimate)} +/- {self._error(est)[0]:.3f} ({self.confidence*100:.0f}% CI, #S:{ len(est) })"


class KneeCriterion(ConvergenceCriterion):
	"""Stop at the knee of the recorded path (counterpart of src/primate/estimators.py:302-333; needs record=True). The path is
	the total variation, up to each sample, of the sequence x_i / i. Rescaled to run from 0 to 1 it is compared with the
	straight line between its end points; the path has a knee once it bulges above that line by more than S / (m - 1),
	m = number of path points (the path ends ON the line, so the bulge is measured against 0 there)."""

	def __init__(self, S: float = 1.0) -> None:
		self.S = S

	def __call__(self, est) -> bool:
		if est.values is None or len(est.values) < 3:
			return False
		x = np.asarray(est.values, dtype=float).reshape(-1)
		scaled = x / np.arange(1, x.size + 1)
		path = np.add.accumulate(np.abs(scaled[1:] - scaled[:-1]))  # non-decreasing: first point is its minimum, last its maximum
		m, span = path.size, path[-1] - path[0]
		if not (span > 0 and self.S > 0):
			return False  # flat path (or no sensitivity): no knee to find
		bulge = (path - path[0]) / span - np.arange(m) / (m - 1)
		return bool(bulge.max() - bulge[-1] > self.S / (m - 1))

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
