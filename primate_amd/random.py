"""Isotropic probe vectors on the HOST, with the reference's exact NumPy stream order (parity
mode). Throughput runs draw probes on the device instead (LanczosPlan.generate_probes: Philox).

Contract reproduced from src/primate/random.py:22-41,47-80 and pinned by tests/golden:
  * output is float64, Fortran-ordered, filled in memory order by ONE Generator, so a batched
    (n, b) draw equals b sequential (n, 1) draws (reference tests/test_random.py:23-39);
  * rademacher = floor(2u)*2-1 with u = rng.random(); normal = rng.standard_normal();
    sphere = sqrt(n) g/||g|| per column.
"""

from __future__ import annotations

from typing import Callable, Optional, Union

import numpy as np

_ISO_DISTRIBUTIONS = {"rademacher": "rademacher", "normal": "normal", "sphere": "sphere", "signs": "rademacher", "gaussian": "normal"}


def _fill(rng: np.random.Generator, pdf: str, out: np.ndarray) -> None:
	if pdf == "rademacher":
		rng.random(out=out)
		np.multiply(out, 2, out=out)
		np.floor(out, out=out)
		np.multiply(out, 2, out=out)
		np.subtract(out, 1, out=out)
	else:
		rng.standard_normal(out=out, dtype=out.dtype)
		if pdf == "sphere":
			## divide by the column norms, then scale by sqrt(n): the reference's order of
			## operations (random.py:38-41), kept so the rounding is identical
			np.divide(out, np.sqrt(np.sum(out**2, axis=0, keepdims=True)), out=out)
			np.multiply(out, np.sqrt(out.shape[0]), out=out)


def isotropic(
	size: Union[int, tuple, None] = None,
	pdf: str = "rademacher",
	seed: Union[int, np.random.Generator, None] = None,
	out: Optional[np.ndarray] = None,
) -> Union[None, np.ndarray, Callable]:
	"""Same signature and behaviour as primate.random.isotropic (src/primate/random.py:47-80)."""
	assert pdf in _ISO_DISTRIBUTIONS, f"Invalid distribution '{pdf}' supplied."
	pdf = _ISO_DISTRIBUTIONS[pdf]
	rng = np.random.default_rng(seed)
	if out is not None:
		assert isinstance(out, np.ndarray)
		_fill(rng, pdf, out)
		return None

	def _isotropic(size: Union[int, tuple]):
		size = (size, 1) if isinstance(size, int) else size
		W = np.empty(shape=size, dtype=np.float64, order="F")
		_fill(rng, pdf, W)
		return W

	_isotropic.pdf = pdf
	return _isotropic if size is None else _isotropic(size)
