"""Small dense helpers of the exchangeable estimators (host side, m x m algebra)."""

import numpy as np


def update_trinv(B_inv: np.ndarray, b: np.ndarray) -> np.ndarray:
	"""Inverse of the upper-triangular [B, b] from B^{-1} (src/primate/linalg.py:5-24): appending a
	column b (length n+1) to an n x n triangular B gives
	[[B^{-1}, -B^{-1} b[:n] / b[n]], [0, 1 / b[n]]]."""
	n, m = B_inv.shape
	assert n == m and len(b) == (n + 1), "B must be n x n and `b` must have length `n + 1`"
	b = np.asarray(b).reshape(n + 1)
	out = np.zeros((n + 1, n + 1))
	out[:n, :n] = B_inv
	out[n, n] = 1.0 / b[n]
	out[:n, n] = -(B_inv @ b[:n]) * out[n, n]
	return out
