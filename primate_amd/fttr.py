"""Forward three-term recurrence weights (src/primate/fttr.py; Laudadio, Mastronardi & Van Dooren 2023)
with the reference's in-place signature, evaluated on the device (`slq_fttr_batch`, kernel `k_fttr`)."""

from __future__ import annotations

import numpy as np

from .engine import fttr_batch


def ortho_poly(x: float, mu_sqrt_rec: float, a: np.ndarray, b: np.ndarray, z: np.ndarray, n: int) -> None:
	"""Values p_0(x), ..., p_{n-1}(x) of the orthonormal polynomials of the Jacobi matrix (a, b) into `z`
	(fttr.py:5-13): p_0 = mu_sqrt_rec, b_{i} p_i = (x - a_{i-1}) p_{i-1} - b_{i-1} p_{i-2}. Host helper (O(n))."""
	z[0] = mu_sqrt_rec
	if n > 1:
		z[1] = (x - a[0]) * z[0] / b[1]
	for i in range(2, n):
		z[i] = ((x - a[i - 1]) * z[i - 1] - b[i - 1] * z[i - 2]) / b[i]


def fttr(theta: np.ndarray, alpha: np.ndarray, beta: np.ndarray, k: int, weights: np.ndarray) -> None:
	"""weights[:k] <- Gaussian quadrature weights at the nodes theta[:k] of the Jacobi matrix (alpha, beta)
	(fttr.py:17-29), written in place like the reference's Pythran kernel."""
	theta, alpha, beta = (np.ascontiguousarray(v, dtype=np.float64) for v in (theta, alpha, beta))
	weights[:k] = fttr_batch(theta[None, :k], alpha[None, :], beta[None, :])[0]
