"""`primate.tqli` names (src/primate/tqli.py): tridiagonal QL with implicit shifts. The reference's version is a
Pythran-compiled Python loop; here the same decomposition runs on the device (`k_eigh_tridiag`) and the results are
written back in place, as the reference does."""

from __future__ import annotations

import numpy as np

from .engine import eigh_tridiag_batch


def sign(a: float, b: float) -> int:
	"""tqli.py:5-7, kept verbatim in behaviour (including its `b > 1` comparison)."""
	return int(b > 1) - int(a < 0) + 1


def tqli(d: np.ndarray, e: np.ndarray, Z: np.ndarray, max_iter: int = 30) -> None:
	"""In place: d <- eigenvalues of T(d, e) (not sorted in the reference either; here ascending), e <- 0, and, when
	Z is non-empty, Z <- Z V with V the eigenvectors (the reference rotates the columns of the Z it is given,
	tqli.py:15-90). `e` has length n with e[0] = 0."""
	assert len(d) == len(e), "Diagonal and subdiagonal should have same length (subdiagonal should be prefixed with 0)"
	assert np.isclose(e[0], 0.0), "Subdiagonal first element should be zero"
	if np.prod(Z.shape) == 0:
		d[:] = eigh_tridiag_batch(d[None, :], e[None, :], vectors=False)[0]
	else:
		w, V = eigh_tridiag_batch(d[None, :], e[None, :], vectors=True)
		d[:] = w[0]
		Z[:] = Z @ V[0]
	e[:] = 0.0
