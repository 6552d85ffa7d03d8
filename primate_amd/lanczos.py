"""`lanczos()` / `rayleigh_ritz()` with the reference's signatures (src/primate/lanczos.py:25-164),
running on the MI355X through libslq's single-vector entry `slq_lanczos_*` — the C-ABI stand-in
for `primate._lanczos.lanczos` (src/primate/_lanczos.cpp:88-99).
"""

from __future__ import annotations

from typing import Any, Optional, Union

import numpy as np

from . import _capi, engine
from ._capi import check, ptr
from .integrate import quadrature


## Operators already on the device, by CONTENT of the sparse matrix they were built from: creating one costs 0.1-0.2 s of host-side analysis
## (row order, tile clusters, streams) and an upload, a Lanczos run over it 0.06 s - and the reference's MatrixFunction(A) costs nothing to construct
## (src/primate/operators.py:55-100), so drivers build one per call. Key: shape, dtype, GPU and a 64-bit hash of indptr / indices / data (xxh3: ~10 GB/s,
## 6 ms for configs[1]); values are weak references, so an operator lives exactly as long as something uses it. A matrix modified in place hashes
## differently and gets a new operator.
_OPERATORS: "weakref.WeakValueDictionary" = None


def _sparse_key(A, dtype):
	try:
		import xxhash
	except Exception:  # noqa: BLE001
		return None
	import scipy.sparse as sp

	if not (sp.issparse(A) and A.format == "csr"):
		return None
	h = xxhash.xxh3_64()
	for arr in (A.indptr, A.indices, A.data):
		h.update(memoryview(np.ascontiguousarray(arr)).cast("B"))
	ctx = engine.default_context()
	import os

	switches = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("SLQ_")))  # (libslq reads its switches when an operator is created)
	return (A.shape, str(A.dtype), str(np.dtype(dtype if dtype is not None else A.dtype)), ctx.device, int(A.nnz), h.intdigest(), switches)


def _as_device_operator(A, dtype=None) -> engine.DeviceOperator:
	global _OPERATORS
	if isinstance(A, engine.DeviceOperator):
		return A
	key = _sparse_key(A, dtype)
	if key is not None:
		import weakref

		if _OPERATORS is None:
			_OPERATORS = weakref.WeakValueDictionary()
		hit = _OPERATORS.get(key)
		if hit is not None and getattr(hit, "_h", None):
			return hit
		op = engine.DeviceOperator(A, dtype=dtype)
		_OPERATORS[key] = op
		return op
	cached = getattr(A, "_slq_device_operator", None)
	if cached is not None and getattr(cached, "_h", None) and (dtype is None or cached.dtype == np.dtype(dtype)):
		return cached
	op = engine.DeviceOperator(A, dtype=dtype)
	try:
		A._slq_device_operator = op  # ndarrays refuse attributes: then no cache
	except Exception:  # noqa: BLE001
		pass
	return op


def _native_lanczos(A, v: np.ndarray, deg: int, rtol: float, orth: int, alpha: np.ndarray, beta: np.ndarray, Q: np.ndarray) -> int:
	"""Drop-in for `_lanczos.lanczos(A, v, deg, rtol, orth, alpha, beta, Q)`: in-place outputs,
	ncv = Q.shape[1]. Returns the number of executed steps."""
	op = _as_device_operator(A, dtype=Q.dtype)
	assert Q.flags["F_CONTIGUOUS"] and Q.flags["WRITEABLE"] and Q.dtype == op.dtype
	assert alpha.dtype == op.dtype and beta.dtype == op.dtype
	v = np.array(v, dtype=op.dtype, copy=True).ravel()  # v is scratch at the native boundary
	fn = _capi.lib().slq_lanczos_f64 if op.dtype == np.float64 else _capi.lib().slq_lanczos_f32
	rc = fn(op.ctx._h, op._h, ptr(v), int(deg), float(rtol), int(orth), ptr(alpha), ptr(beta), ptr(Q), Q.shape[1])
	if rc == _capi.SLQ_ECALLBACK and getattr(op, "error", None) is not None:
		raise op.error
	return check(rc)


def lanczos(
	A,
	v0: Optional[np.ndarray] = None,
	deg: Optional[int] = None,
	rtol: float = 1e-8,
	orth: int = 0,
	sparse_mat: bool = False,
	return_basis: bool = False,
	seed: Union[int, np.random.Generator, None] = None,
	dtype: Optional[np.dtype] = None,
	**kwargs: Any,
) -> tuple:
	"""Paige-A27 Lanczos tridiagonalisation. Returns (a, b) = (alpha[:deg], beta[1:deg]) and, with
	return_basis, the n x deg Lanczos basis Q. Parameter clamps follow lanczos.py:78-90."""
	n: int = A.shape[0]
	deg = A.shape[1] if deg is None else min(deg, A.shape[1])
	assert deg > 0, "Number of steps must be positive!"
	if dtype is not None:
		f_dtype = np.dtype(dtype)
	elif hasattr(A, "dtype"):
		f_dtype = np.dtype(A.dtype)
	else:
		f_dtype = (A @ np.zeros(A.shape[1])).dtype  # the reference infers it the same way (lanczos.py:84)
	assert f_dtype.type in {np.float32, np.float64}, "Only 32- or 64-bit floating point numbers are supported."
	orth = deg if orth < 0 or orth > deg else orth
	ncv = int(np.clip(orth, 2, deg)) if not return_basis else deg
	if v0 is None:
		rng = np.random.default_rng(seed)
		v0 = rng.uniform(size=A.shape[1], low=-1.0, high=+1.0).astype(f_dtype)
	else:
		v0 = np.array(v0).astype(f_dtype)
	assert len(v0) == A.shape[1], "Invalid starting vector; must match the number of columns of A."
	alpha = kwargs.get("alpha", np.zeros(deg + 1, dtype=f_dtype))
	beta = kwargs.get("beta", np.zeros(deg + 1, dtype=f_dtype))
	Q = kwargs.get("Q", np.zeros((n, max(ncv, 2)), dtype=f_dtype, order="F"))
	assert isinstance(alpha, np.ndarray) and len(alpha) == deg + 1 and alpha.dtype == f_dtype and alpha.flags["WRITEABLE"]
	assert isinstance(beta, np.ndarray) and len(beta) == deg + 1 and beta.dtype == f_dtype and beta.flags["WRITEABLE"]
	assert Q.ndim == 2 and Q.shape[0] == n and Q.flags["F_CONTIGUOUS"] and Q.flags["WRITEABLE"]
	_native_lanczos(A, v0, deg, rtol, orth, alpha, beta, Q)
	if sparse_mat:
		from scipy.sparse import spdiags

		T = spdiags(data=[np.roll(beta, -1), alpha, beta], diags=(-1, 0, +1), m=deg, n=deg)
		return T if not return_basis else (T, Q)
	a, b = alpha[:deg], beta[1:deg]
	return (a, b) if not return_basis else ((a, b), Q)


def rayleigh_ritz(A, deg: Optional[int] = None, return_eigenvectors: bool = False, method: str = "RRR", **kwargs):
	"""Ritz values (and vectors) of A from a degree-`deg` Lanczos run (lanczos.py:120-164)."""
	n: int = A.shape[0]
	deg = A.shape[1] if deg is None else min(deg, A.shape[1])
	assert deg > 0, "Number of steps must be positive!"
	deg = int(np.clip(deg, 2, n))
	want_Q = kwargs.pop("return_basis", False)
	res = lanczos(A, deg=deg, return_basis=want_Q, **kwargs)
	(a, b), Q = res if want_Q else (res, None)
	if return_eigenvectors:
		from .tridiag import eigh_tridiag  # device QL (k <= 141: the k x k eigenvectors live in LDS)

		rw, Y = eigh_tridiag(a, b)
		return (rw, Y) if not want_Q else (rw, Y, Q)
	rw, _ = quadrature(a, np.append([0], b))
	return rw if not want_Q else (rw, Q)
