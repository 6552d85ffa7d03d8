"""Spectral-function registry: same names, parameters and values as the reference's
`primate.special` (src/primate/special.py:7-107). A string name selects the on-device evaluation
(libslq SLQ_FUN_*); a Python callable is applied on the host to the k quadrature nodes per probe.
"""

from __future__ import annotations

from typing import Any, Callable, Optional, Union

import numpy as np

_BUILTIN_MATRIX_FUNCTIONS = ["identity", "abs", "sqrt", "log", "inv", "exp", "smoothstep", "numrank"]
_EPS64 = np.finfo(np.float64).eps


def identity(x: Any) -> Any:
	return x


def exp(x: Optional[np.ndarray] = None, t: float = 1.0) -> Union[Callable, np.ndarray]:
	"""x -> exp(t x) (special.py:62-66)."""
	f = lambda z: np.exp(t * z)  # noqa: E731
	return f if x is None else f(x)


def step(x: Optional[np.ndarray] = None, c: float = 0.0, nonnegative: bool = False) -> Union[Callable, np.ndarray]:
	"""Indicator of x >= c, on |x| if `nonnegative` (special.py:69-74)."""

	def f(z):
		z = np.abs(z) if nonnegative else z
		return np.where(z < c, 0.0, 1.0)

	return f if x is None else f(x)


def smoothstep(x: Optional[np.ndarray] = None, a: float = 0.0, b: float = 1.0, deg: int = 3) -> Union[Callable, np.ndarray]:
	"""Cubic Hermite step from 0 at a to 1 at b (special.py:33-55)."""
	assert (deg % 2) == 1, "Degree must be odd"
	width = (b - a) if a != b else 1.0

	def f(z):
		y = np.clip((z - a) / width, 0.0, 1.0)
		return 3 * y**2 - 2 * y**3

	return f if x is None else f(x)


def softsign(x: Optional[np.ndarray] = None, q: int = 1) -> Union[Callable, np.ndarray]:
	"""sum_{i<=q} x (1-x^2)^i J_i with J_i = prod_{j<=i} (2j-1)/(2j), x clipped to [-1,1]
	(special.py:10-30)."""
	powers = np.arange(q + 1)
	coeffs = np.concatenate([[1.0], np.cumprod([(2 * j - 1) / (2 * j) for j in range(1, q + 1)])])

	def f(z):
		z = np.clip(np.atleast_1d(z), -1.0, 1.0)[:, None]
		return np.ravel(np.sum(z * (1 - z**2) ** powers * coeffs, axis=1))

	return f if x is None else f(x)


def builtin_spec(fun, **kwargs) -> Optional[tuple]:
	"""(name, kwargs) when `fun` can be evaluated on the device, else None."""
	if fun is None:
		return ("identity", {})
	if isinstance(fun, str):
		keys = {"exp": ["t"], "smoothstep": ["a", "b"], "numrank": ["threshold"], "softsign": ["q"], "step": ["c", "nonnegative"]}
		return (fun, {k: kwargs[k] for k in keys.get(fun, []) if k in kwargs})
	return getattr(fun, "_slq_builtin", None)


def param_callable(fun: Union[str, Callable, None], **kwargs) -> Callable:
	"""Name -> callable with the reference's defaults (special.py:78-107). The returned callable
	carries `_slq_builtin` so MatrixFunction can evaluate it on the device."""
	if isinstance(fun, str):
		assert fun in _BUILTIN_MATRIX_FUNCTIONS + ["softsign"], "If given as a string, matrix_function be one of the builtin functions."
	if fun is None or (isinstance(fun, str) and fun == "identity"):
		f, spec = identity, ("identity", {})
	elif callable(fun):
		return fun
	elif fun == "abs":
		f, spec = np.abs, ("abs", {})
	elif fun == "sqrt":
		f, spec = np.sqrt, ("sqrt", {})
	elif fun == "log":
		f, spec = (lambda x: np.log(np.maximum(x, _EPS64))), ("log", {})
	elif fun == "inv":
		f, spec = np.reciprocal, ("inv", {})
	elif fun == "exp":
		t = kwargs.pop("t", 1.0)
		f, spec = exp(t=t), ("exp", {"t": t})
	elif fun == "smoothstep":
		a, b = kwargs.pop("a", 0.0), kwargs.pop("b", 1.0)
		f, spec = smoothstep(a=a, b=b), ("smoothstep", {"a": a, "b": b})
	elif fun == "softsign":
		q = kwargs.pop("q", 10)
		f, spec = softsign(q=q), ("softsign", {"q": q})
	elif fun == "numrank":
		thr = kwargs.pop("threshold", 0.000001)
		f, spec = step(c=thr, nonnegative=True), ("numrank", {"threshold": thr})
	else:
		raise ValueError(f"Unknown function: {fun}.")
	if f in (np.abs, np.sqrt, np.reciprocal, identity):
		## ufuncs / shared functions cannot carry attributes: wrap them
		g = f
		f = lambda x: g(x)  # noqa: E731
	f._slq_builtin = spec
	return f
