"""`primate.stats` names (src/primate/stats.py): the streaming moments live in `estimators.py` here."""

from __future__ import annotations

from numbers import Number

import numpy as np

from .estimators import Covariance, Mean  # noqa: F401


def confidence_interval(a: np.ndarray, confidence: float = 0.95, sdist: str = "t") -> tuple:
	"""Two-sided confidence interval of the sample mean of `a` (stats.py:102-113): Student-t with the
	standard error of the mean (ddof = 1), or the normal approximation."""
	assert isinstance(confidence, Number) and 0.0 <= confidence <= 1.0, "Invalid confidence measure"
	from scipy import stats as sps

	a = np.asarray(a, dtype=float)
	mean, sem = float(np.mean(a)), float(np.std(a, ddof=1) / np.sqrt(len(a)))
	if sdist == "t":
		half = sps.t.ppf((1.0 + confidence) / 2.0, len(a) - 1) * sem
		return mean - half, mean + half
	if sdist == "normal":
		return sps.norm.interval(confidence, loc=mean, scale=sem)
	raise ValueError(f"Unknown sampling distribution '{sdist}'.")
