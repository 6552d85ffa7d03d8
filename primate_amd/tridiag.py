"""`primate.tridiag` names (src/primate/tridiag.py:25-62) on the device: implicit QL with Wilkinson shifts, one
wave per matrix (`k_eigh_tridiag`). The reference dispatches to LAPACK's MRRR (`method="mrrr"`/"auto") or its own
`tqli`; both are the same decomposition, so `method`/`maxiter` are accepted and ignored."""

from __future__ import annotations

import numpy as np

from .engine import eigh_tridiag_batch


def _prepare(d: np.ndarray, e: np.ndarray, method: str) -> tuple:
	assert method in {"tqli", "mrrr", "auto"}
	d, e = np.asarray(d, dtype=np.float64), np.asarray(e, dtype=np.float64)
	assert len(d) in {len(e) + 1, len(e)}, "Invalid diagonal/subdiagonal pair"
	e = np.append([0.0], e) if len(e) == len(d) - 1 else e
	return d, e


def eigh_tridiag(d: np.ndarray, e: np.ndarray, method: str = "auto", maxiter: int = 30) -> tuple:
	"""Ritz pairs (values ascending, vectors in columns) of the symmetric tridiagonal with diagonal `d` and
	subdiagonal `e` (length n with a leading 0, or n - 1)."""
	d, e = _prepare(d, e, method)
	w, Z = eigh_tridiag_batch(d[None, :], e[None, :], vectors=True)
	return w[0], Z[0]


def eigvalsh_tridiag(d: np.ndarray, e: np.ndarray, method: str = "auto", maxiter: int = 30) -> np.ndarray:
	"""Eigenvalues only, ascending."""
	d, e = _prepare(d, e, method)
	return eigh_tridiag_batch(d[None, :], e[None, :], vectors=False)[0]
