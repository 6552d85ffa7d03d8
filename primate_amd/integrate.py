"""Gauss quadrature from a Jacobi matrix — same signature as the reference's
`primate.integrate.quadrature` (src/primate/integrate.py:9-76); the eigen-solve runs in libslq's
on-device implicit-QL kernel instead of LAPACK stemr (src/primate/tridiag.py:10-11).
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from . import engine


def quadrature(
	d: np.ndarray,
	e: np.ndarray,
	deg: Optional[int] = None,
	quad: str = "gw",
	nodes: Optional[np.ndarray] = None,
	weights: Optional[np.ndarray] = None,
	**kwargs,
) -> tuple:
	"""Nodes (ascending eigenvalues of T(d, e)) and weights (squared first eigenvector components)
	of the degree-`deg` Gauss rule. `e` may have len(d) entries (e[0] = 0) or len(d) - 1."""
	d = np.asarray(d)
	e = np.asarray(e)
	deg = len(d) if deg is None else int(min(deg, len(d)))
	e = np.append([0], e) if len(e) == (len(d) - 1) else e
	assert len(d) == len(e) and np.isclose(e[0], 0.0), "Subdiagonal first element 'e[0]' must be close to zero"
	if quad in {"gw", "golub_welsch"}:
		theta, tau = engine.quadrature_batch(d[:deg][None, :], e[:deg][None, :])
		theta, tau = theta[0].astype(d.dtype, copy=False), tau[0].astype(d.dtype, copy=False)
	elif quad == "fttr":
		## nodes of the FULL Jacobi matrix, weights by the forward three-term recurrence over all of
		## (d, e) for the first `deg` nodes (integrate.py:65-69, fttr.py:17-29), both on the device
		theta, _ = engine.quadrature_batch(d[None, :], e[None, :])
		theta = theta[0].astype(d.dtype, copy=False)
		tau = np.zeros(len(theta), dtype=theta.dtype)
		tau[:deg] = engine.fttr_batch(theta[None, :], d[None, :], e[None, :], k=deg)[0]
	else:
		raise ValueError(f"Invalid quadrature method '{quad}' supplied")
	if nodes is not None and weights is not None:
		assert len(nodes) == deg and len(weights) == deg, "`nodes` and `weights` output arrays must be `deg` in length."
		np.copyto(nodes, theta)
		np.copyto(weights, tau)
	return theta, tau
