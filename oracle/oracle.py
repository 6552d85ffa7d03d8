"""TEST INFRASTRUCTURE — Python face of the CPU oracle (oracle/slq_oracle.c). NOT product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package primate_amd/ never does.

What is here:
  * build() / lib(): compile (gcc) and load oracle/_build/libslq_oracle.so.
  * make_operator(): wrap ndarray / scipy.sparse / object-with-matvec as the C operator struct,
    the oracle-side mirror of the six overloads of the reference FFI
    (src/primate/_lanczos.cpp:102-112).
  * lanczos(A, v, deg, rtol, orth, alpha, beta, Q): same signature and in-place semantics as the
    reference's `primate._lanczos.lanczos` (src/primate/_lanczos.cpp:88-99); used by
    tests/golden/make_golden.py to stand in for the unbuildable extension under the reference's
    own Python drivers.
  * quad_batch(), quadrature_gw(), tridiag_ql(), fttr(), apply_fun(): the rest of the path.
  * numpy restatements (np_*) of the same functions for cross-checking the C code.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libslq_oracle.so"
_SRCS = ["slq_oracle.c", "slq_oracle.h", "slq_oracle_impl.h", "slq_oracle_batch.h"]

OP_CSR, OP_CSC, OP_DENSE, OP_CALLBACK = 0, 1, 2, 3
FUN_IDS = {
	"identity": 0,
	"abs": 1,
	"sqrt": 2,
	"log": 3,
	"inv": 4,
	"exp": 5,
	"smoothstep": 6,
	"step": 7,
	"numrank": 7,
	"softsign": 8,
}


def fun_spec(fun, **kwargs) -> tuple:
	"""(fun_id, params[4]) following src/primate/special.py:78-107 defaults."""
	fun = "identity" if fun is None else fun
	p = np.zeros(4)
	if fun == "exp":
		p[0] = kwargs.get("t", 1.0)
	elif fun == "smoothstep":
		p[0], p[1] = kwargs.get("a", 0.0), kwargs.get("b", 1.0)
	elif fun == "numrank":
		p[0], p[1] = kwargs.get("threshold", 0.000001), 1.0
	elif fun == "step":
		p[0], p[1] = kwargs.get("c", 0.0), float(kwargs.get("nonnegative", False))
	elif fun == "softsign":
		p[0] = kwargs.get("q", 10)
	return FUN_IDS[fun], p


def build(force: bool = False) -> Path:
	"""Compile the oracle with gcc if the .so is missing or older than its sources."""
	stale = force or not _SO.exists()
	if not stale:
		t = _SO.stat().st_mtime
		stale = any((_HERE / s).stat().st_mtime > t for s in _SRCS)
	if stale:
		subprocess.run(["make", "-C", str(_HERE), "-B"], check=True, capture_output=True)
	return _SO


_lib = None


def _matvec_cb_type(ct):
	return C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(ct), C.POINTER(ct))


def _op_struct(ct):
	class Op(C.Structure):
		_fields_ = [
			("kind", C.c_int32),
			("_pad", C.c_int32),
			("nrows", C.c_int64),
			("ncols", C.c_int64),
			("ptr", C.c_void_p),
			("ind", C.c_void_p),
			("vals", C.c_void_p),
			("lda", C.c_int64),
			("matvec", _matvec_cb_type(ct)),
			("ctx", C.c_void_p),
		]

	return Op


_OPS = {np.dtype("float32"): _op_struct(C.c_float), np.dtype("float64"): _op_struct(C.c_double)}
_CT = {np.dtype("float32"): C.c_float, np.dtype("float64"): C.c_double}
_SUF = {np.dtype("float32"): "_f32", np.dtype("float64"): "_f64"}


def lib():
	global _lib
	if _lib is None:
		build()
		_lib = C.CDLL(str(_SO))
		_lib.oracle_apply_fun.restype = C.c_double
		_lib.oracle_apply_fun.argtypes = [C.c_int, C.c_void_p, C.c_double]
	return _lib


def _ptr(a):
	return None if a is None else a.ctypes.data_as(C.c_void_p)


class Operator:
	"""Owns the arrays behind a C oracle_operator struct (keeps them alive)."""

	def __init__(self, A, dtype=None, prefer: str = "csc"):
		import scipy.sparse as sp

		if dtype is None:
			dtype = getattr(A, "dtype", np.dtype("float64"))
		self.dtype = np.dtype(dtype)
		assert self.dtype in _OPS, "Only 32- or 64-bit floats are supported."
		ct = _CT[self.dtype]
		self.shape = tuple(A.shape)
		self._keep = []
		op = _OPS[self.dtype]()
		op.nrows, op.ncols = self.shape
		if isinstance(A, np.ndarray):
			M = np.asfortranarray(A, dtype=self.dtype)
			self._keep.append(M)
			op.kind, op.vals, op.lda = OP_DENSE, _ptr(M), M.shape[0]
		elif sp.issparse(A):
			M = (sp.csc_matrix(A) if prefer == "csc" else sp.csr_matrix(A)).astype(self.dtype)
			M.sort_indices()
			ptr = np.ascontiguousarray(M.indptr, dtype=np.int32)
			ind = np.ascontiguousarray(M.indices, dtype=np.int32)
			vals = np.ascontiguousarray(M.data, dtype=self.dtype)
			self._keep += [ptr, ind, vals]
			op.kind = OP_CSC if prefer == "csc" else OP_CSR
			op.ptr, op.ind, op.vals = _ptr(ptr), _ptr(ind), _ptr(vals)
		else:
			## Any object with .matvec and .shape (src/primate/include/pylinop.h:22-29)
			if not hasattr(A, "matvec"):
				raise ValueError("Supplied object is missing 'matvec' attribute.")
			if not hasattr(A, "shape"):
				raise ValueError("Supplied object is missing 'shape' attribute.")
			n_out, n_in = self.shape
			dt = self.dtype
			self.error = None

			def _cb(_ctx, x, y):
				try:
					xin = np.ctypeslib.as_array(x, shape=(n_in,))
					out = np.asarray(A.matvec(xin.copy())).astype(dt, copy=False).ravel()
					np.ctypeslib.as_array(y, shape=(n_out,))[:] = out[:n_out]
					return 0
				except Exception as e:  # noqa: BLE001
					self.error = e
					return 1

			cb = _matvec_cb_type(ct)(_cb)
			self._keep.append(cb)
			op.kind, op.matvec = OP_CALLBACK, cb
		self.c = op

	def matvec(self, x):
		x = np.ascontiguousarray(x, dtype=self.dtype)
		y = np.empty(self.shape[0], dtype=self.dtype)
		if self.c.kind == OP_CALLBACK:
			rc = self.c.matvec(None, x.ctypes.data_as(C.POINTER(_CT[self.dtype])), y.ctypes.data_as(C.POINTER(_CT[self.dtype])))
			assert rc == 0
		elif self.c.kind == OP_DENSE:
			getattr(lib(), "oracle_dense_matvec" + _SUF[self.dtype])(
				C.c_int64(self.shape[0]), C.c_int64(self.shape[1]), C.c_void_p(self.c.vals), C.c_int64(self.c.lda), _ptr(x), _ptr(y)
			)
		elif self.c.kind == OP_CSC:
			getattr(lib(), "oracle_csc_matvec" + _SUF[self.dtype])(
				C.c_int64(self.shape[0]), C.c_int64(self.shape[1]), C.c_void_p(self.c.ptr), C.c_void_p(self.c.ind),
				C.c_void_p(self.c.vals), _ptr(x), _ptr(y),
			)
		else:
			getattr(lib(), "oracle_csr_matvec" + _SUF[self.dtype])(
				C.c_int64(self.shape[0]), C.c_void_p(self.c.ptr), C.c_void_p(self.c.ind), C.c_void_p(self.c.vals), _ptr(x), _ptr(y)
			)
		return y


def make_operator(A, dtype=None, prefer: str = "csc") -> Operator:
	return A if isinstance(A, Operator) else Operator(A, dtype=dtype, prefer=prefer)


def _infer_dtype(A, *arrays):
	for a in arrays:
		if isinstance(a, np.ndarray) and a.dtype in _OPS:
			return a.dtype
	return np.dtype(getattr(A, "dtype", "float64"))


def lanczos(A, v, deg: int, rtol: float, orth: int, alpha, beta, Q) -> int:
	"""Drop-in for `primate._lanczos.lanczos` (src/primate/_lanczos.cpp:88-99).

	alpha/beta/Q are written in place; ncv = Q.shape[1] (:94). pybind11 passes `v` BY VALUE through
	an f_style|forcecast array_t (:82-83,90), so the caller's `v` is left untouched when a
	conversion copy is made; we always work on a copy. Sparse input goes through the CSC scatter
	matvec like Eigen's (eigen_operators.h:66-77). Returns the number of executed steps.
	"""
	dt = _infer_dtype(A, Q, alpha)
	op = make_operator(A, dtype=dt)
	assert Q.flags["F_CONTIGUOUS"] and Q.dtype == dt and alpha.dtype == dt and beta.dtype == dt
	assert len(alpha) >= deg + 1 and len(beta) >= deg + 1 and Q.shape[0] == op.shape[0]
	q = np.array(v, dtype=dt, copy=True).ravel()
	fn = getattr(lib(), "oracle_lanczos_recurrence" + _SUF[dt])
	fn.restype = C.c_int
	steps = fn(
		C.byref(op.c), _ptr(q), C.c_int(int(deg)), _CT[dt](rtol), C.c_int(int(orth)),
		_ptr(alpha), _ptr(beta), _ptr(Q), C.c_int64(Q.shape[1]),
	)  # fmt: skip
	if steps < 0:
		raise getattr(op, "error", None) or RuntimeError("operator callback failed")
	return steps


def tridiag_ql(d, e, want_vectors: bool = False, first_row_only: bool = False, maxiter: int = 60):
	"""Eigenvalues (unsorted) [and eigenvectors] of T(d, e); e has len(d) entries with e[0] = 0."""
	dt = np.dtype(d.dtype)
	n = len(d)
	dd = np.array(d, dtype=dt, copy=True)
	ee = np.array(e, dtype=dt, copy=True)
	if first_row_only:
		Z = np.zeros((1, n), dtype=dt)
		Z[0, 0] = 1
	elif want_vectors:
		Z = np.eye(n, dtype=dt)
	else:
		Z = np.zeros((0, n), dtype=dt)
	fn = getattr(lib(), "oracle_tridiag_ql" + _SUF[dt])
	fn.restype = C.c_int
	rc = fn(C.c_int(n), _ptr(dd), _ptr(ee), _ptr(Z), C.c_int(Z.shape[0]), C.c_int(maxiter))
	return (dd, Z, rc) if Z.shape[0] else (dd, rc)


def quadrature_gw(d, e, deg=None):
	"""(nodes, weights) as src/primate/integrate.py:57-64 computes them (ascending nodes)."""
	dt = np.dtype(d.dtype)
	deg = len(d) if deg is None else int(min(deg, len(d)))
	e = np.append([0], e).astype(dt) if len(e) == len(d) - 1 else np.asarray(e, dtype=dt)
	d = np.ascontiguousarray(d, dtype=dt)
	e = np.ascontiguousarray(e, dtype=dt)
	nodes, weights = np.zeros(deg, dtype=dt), np.zeros(deg, dtype=dt)
	work = np.zeros(3 * deg, dtype=dt)
	fn = getattr(lib(), "oracle_quadrature_gw" + _SUF[dt])
	fn.restype = C.c_int
	rc = fn(C.c_int(deg), _ptr(d), _ptr(e), _ptr(nodes), _ptr(weights), _ptr(work))
	assert rc == 0, "QL failed to converge"
	return nodes, weights


def fttr(theta, alpha, beta, k):
	dt = np.dtype(theta.dtype)
	n = len(alpha)
	w = np.zeros(len(theta), dtype=dt)
	p = np.zeros(n, dtype=dt)
	getattr(lib(), "oracle_fttr" + _SUF[dt])(
		_ptr(np.ascontiguousarray(theta)), _ptr(np.ascontiguousarray(alpha, dtype=dt)),
		_ptr(np.ascontiguousarray(beta, dtype=dt)), C.c_int(n), C.c_int(int(k)), _ptr(w), _ptr(p),
	)  # fmt: skip
	return w


def apply_fun(fun, x, **kwargs):
	fid, p = fun_spec(fun, **kwargs)
	L = lib()
	return np.array([L.oracle_apply_fun(fid, _ptr(p), float(t)) for t in np.atleast_1d(x)])


def quad_batch(
	A, X, deg: int, orth: int, fun="identity", rtol: float = 1e-8, fresh_q: bool = True, nthreads: int = 1,
	return_rule: bool = False, prefer: str = "csc", **fun_kwargs,
):  # fmt: skip
	"""x_j^T f(A) x_j for every column of X; see oracle/slq_oracle_batch.h."""
	dt = _infer_dtype(A, X)
	op = make_operator(A, dtype=dt, prefer=prefer)
	X = np.asfortranarray(np.atleast_2d(X.T).T if X.ndim == 1 else X, dtype=dt)
	n, b = X.shape
	assert n == op.shape[1]
	deg = min(int(deg), n)
	fid, p = fun_spec(fun, **fun_kwargs)
	out = np.zeros(b)
	nodes = np.zeros((b, deg), dtype=dt) if return_rule else None
	weights = np.zeros((b, deg), dtype=dt) if return_rule else None
	steps = np.zeros(b, dtype=np.int32)
	fn = getattr(lib(), "oracle_quad_batch" + _SUF[dt])
	fn.restype = C.c_int
	rc = fn(
		C.byref(op.c), _ptr(X), C.c_int64(X.shape[0]), C.c_int(b), C.c_int(deg), _CT[dt](rtol), C.c_int(int(orth)),
		C.c_int(fid), _ptr(p), C.c_int(int(fresh_q)), C.c_int(int(nthreads)), _ptr(out), _ptr(nodes), _ptr(weights),
		_ptr(steps),
	)  # fmt: skip
	if rc == -2:
		raise MemoryError("oracle_quad_batch: allocation failed")
	if rc < 0:
		raise getattr(op, "error", None) or RuntimeError("operator callback failed")
	return (out, nodes, weights, steps) if return_rule else out


## ---------------------------------------------------------------------------------------------
## NumPy restatements (second, independent statement of the same algorithm; small cases only)
## ---------------------------------------------------------------------------------------------
def np_orth_vector(v, U, start_idx, p, reverse=False):
	"""src/primate/include/lanczos.h:43-66."""
	n, m = U.shape
	tol = 2 * np.finfo(U.dtype).eps * np.sqrt(n)
	step = -1 if reverse else 1
	i = start_idx % m
	for _ in range(p):
		u = U[:, i]
		u_norm, s_proj = u @ u, v @ u
		if u_norm > tol and abs(s_proj) > tol:
			v -= (s_proj / u_norm) * u
		i = (i + step) % m


def np_lanczos(matvec, q, deg, rtol, orth, alpha, beta, V):
	"""src/primate/include/lanczos.h:92-149; same in-place contract as the C oracle."""
	n, ncv = V.shape
	tol = np.sqrt(n) * rtol
	pos = [ncv - 1, 0, 1]
	V[:, pos[0]] = 0
	V[:, 0] = q / np.linalg.norm(q)
	beta[0] = 0
	steps = 0
	for j in range(deg):
		p, c, nx = pos
		q[:] = matvec(V[:, c])
		q -= beta[j] * V[:, p]
		alpha[j] = V[:, c] @ q
		q -= alpha[j] * V[:, c]
		if orth > 0:
			np_orth_vector(q, V, c, orth, reverse=True)
		beta[j + 1] = np.linalg.norm(q)
		steps = j + 1
		if beta[j + 1] < tol or (j + 1) == deg:
			break
		V[:, nx] = q / beta[j + 1]
		pos = [c, nx, (j + 2) % ncv]
	return steps


def np_quadrature(d, e, deg=None):
	"""src/primate/integrate.py:57-64 with the same LAPACK call as src/primate/tridiag.py:10-11."""
	from scipy.linalg import eigh_tridiagonal

	deg = len(d) if deg is None else int(min(deg, len(d)))
	e = np.append([0], e) if len(e) == len(d) - 1 else e
	theta, ev = eigh_tridiagonal(d[:deg], e[1:deg])
	return theta, np.square(ev[0, :])
