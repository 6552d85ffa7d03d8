"""Object layer over the C-ABI: Context (one GPU + stream), DeviceOperator (operator plugin
resident on that GPU) and LanczosPlan (workspace + state of one batched lock-step Lanczos run).

Everything numeric happens in libslq's HIP kernels; this file only marshals arrays.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _capi
from ._capi import check, ptr


def fun_spec(fun, **kwargs) -> tuple:
	"""Map a built-in spectral function name to (fun_id, params) following the defaults of the
	reference registry (src/primate/special.py:78-107). Returns (None, None) for Python callables,
	which are applied on the host to the returned nodes."""
	if fun is None:
		fun = "identity"
	if not isinstance(fun, str):
		return None, None
	assert fun in _capi.FUN_IDS, "If given as a string, matrix_function be one of the builtin functions."
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
	return _capi.FUN_IDS[fun], p


class Context:
	"""One per (process, GPU). `device=None` picks LOCAL_RANK (torchrun) or the current device."""

	def __init__(self, device: Optional[int] = None, stream: Optional[int] = None):
		L = _capi.lib()
		if device is None:
			device = int(os.environ["LOCAL_RANK"]) if "LOCAL_RANK" in os.environ else -1
		h = C.c_void_p()
		check(L.slq_context_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
		self._h = h
		d = C.c_int(-1)
		check(L.slq_context_device(h, C.byref(d)))
		self.device = int(d.value)  # the resolved ordinal: what owns every pointer and stream this context hands out

	def synchronize(self):
		check(_capi.lib().slq_context_synchronize(self._h))

	def measure_stream(self, mode: str = "triad", nbytes: int = 1 << 31, reps: int = 10) -> float:
		"""Measured device bandwidth in GB/s: mode 'read' (2 read streams), 'triad' (in place, 2R+1W) or 'copy'."""
		g = C.c_double()
		check(_capi.lib().slq_measure_stream(self._h, {"read": 0, "triad": 1, "copy": 2}[mode], int(nbytes), int(reps), C.byref(g)))
		return g.value

	def meminfo(self) -> tuple:
		f, t = C.c_size_t(), C.c_size_t()
		check(_capi.lib().slq_context_meminfo(self._h, C.byref(f), C.byref(t)))
		return f.value, t.value

	def close(self):
		if getattr(self, "_h", None):
			_capi.lib().slq_context_destroy(self._h)
			self._h = None

	def __del__(self):
		try:
			self.close()
		except Exception:  # noqa: BLE001
			pass


_default_ctx: Optional[Context] = None


def default_context() -> Context:
	global _default_ctx
	if _default_ctx is None:
		_default_ctx = Context()
	return _default_ctx


class DeviceOperator:
	"""Operator plugin resident on the GPU: the counterpart of the reference's C++ operator
	wrappers (src/primate/include/eigen_operators.h:17-104, src/primate/include/pylinop.h:16-73).

	Accepted inputs mirror the six overloads of the reference FFI
	(src/primate/_lanczos.cpp:102-112): ndarray (dense), scipy.sparse (any format; stored as CSR
	with int32 indices), or any object with `.matvec` and `.shape` (host-callback fallback); beyond them a
	torch sparse-CSR tensor that already lives on the context's GPU (slq_csr_create_device).
	"""

	def __init__(self, A, dtype=None, ctx: Optional[Context] = None):
		import scipy.sparse as sp

		self.ctx = ctx or default_context()
		L = _capi.lib()
		## a torch sparse-CSR tensor on the GPU: slq_csr_create_device (its arrays are read from HBM, never through numpy)
		torch_csr = type(A).__module__.split(".")[0] == "torch" and str(getattr(A, "layout", "")) == "torch.sparse_csr"
		if torch_csr and dtype is None:
			dtype = {"torch.float64": np.float64, "torch.float32": np.float32}[str(A.dtype)]
		if dtype is None:
			dtype = getattr(A, "dtype", np.float64)
		self.dtype = np.dtype(dtype)
		dt = _capi.dtype_id(self.dtype)
		assert hasattr(A, "shape") and len(A.shape) >= 2, "Operator must be at least two dimensional."
		assert A.shape[0] == A.shape[1], "This function only works with square, symmetric matrices!"
		self.shape = (int(A.shape[0]), int(A.shape[1]))
		self._keep = []
		h = C.c_void_p()
		if torch_csr:
			import torch

			if not A.is_cuda:
				raise ValueError("a torch sparse-CSR operator must live on the GPU (use scipy.sparse for host matrices)")
			if A.device.index != self.ctx.device:
				raise ValueError(f"the matrix is on {A.device}, the context on GPU {self.ctx.device}")
			tdt = torch.float64 if self.dtype == np.float64 else torch.float32
			crow = A.crow_indices().to(torch.int32).contiguous()
			col = A.col_indices().to(torch.int32).contiguous()  # (sorted within each row, as torch builds them)
			val = A.values().to(tdt).contiguous()
			torch.cuda.synchronize(A.device)
			check(L.slq_csr_create_device(self.ctx._h, dt, self.shape[0], int(val.numel()), C.c_void_p(crow.data_ptr()), C.c_void_p(col.data_ptr()),
										 C.c_void_p(val.data_ptr()), C.byref(h)))  # fmt: skip
			self.kind, self.nnz = "csr", int(val.numel())
		elif isinstance(A, np.ndarray):
			M = np.asfortranarray(A, dtype=self.dtype)
			check(L.slq_dense_create(self.ctx._h, dt, M.shape[0], ptr(M), M.shape[0], C.byref(h)))
			self.kind, self.nnz = "dense", M.size
		elif sp.issparse(A):
			M = sp.csr_matrix(A)  # (shares A's arrays when A is CSR already: nothing below writes to them)
			if M.dtype != self.dtype:
				M = M.astype(self.dtype)
			if not M.has_sorted_indices:
				M = M.sorted_indices()
			rowptr = np.ascontiguousarray(M.indptr, dtype=np.int32)
			colind = np.ascontiguousarray(M.indices, dtype=np.int32)
			vals = np.ascontiguousarray(M.data, dtype=self.dtype)
			check(L.slq_csr_create(self.ctx._h, dt, M.shape[0], M.nnz, ptr(rowptr), ptr(colind), ptr(vals), C.byref(h)))
			self.kind, self.nnz = "csr", int(M.nnz)
		elif getattr(A, "_slq_kind", None) == "gram":
			## x -> B^T (B x) for a rectangular sparse B (primate_amd.operators.GramOperator): slq_csr_gram_create
			M = sp.csr_matrix(A.A).astype(self.dtype)
			M.sort_indices()
			rowptr = np.ascontiguousarray(M.indptr, dtype=np.int32)
			colind = np.ascontiguousarray(M.indices, dtype=np.int32)
			vals = np.ascontiguousarray(M.data, dtype=self.dtype)
			check(L.slq_csr_gram_create(self.ctx._h, dt, M.shape[0], M.shape[1], M.nnz, ptr(rowptr), ptr(colind), ptr(vals), C.byref(h)))
			self.kind, self.nnz = "gram", int(M.nnz)
		elif getattr(A, "_slq_kind", None) == "affine":
			## A + t B (primate_amd.operators.AffineOperator): slq_csr_affine_create; t follows A.set_parameter
			Ma, Mb = (sp.csr_matrix(Z).astype(self.dtype) for Z in (A.A, A.B))
			for Z in (Ma, Mb):
				Z.sort_indices()
			arrs = [np.ascontiguousarray(Z.indptr, dtype=np.int32) for Z in (Ma, Mb)] + [np.ascontiguousarray(Z.indices, dtype=np.int32) for Z in (Ma, Mb)]
			va, vb = (np.ascontiguousarray(Z.data, dtype=self.dtype) for Z in (Ma, Mb))
			check(L.slq_csr_affine_create(self.ctx._h, dt, Ma.shape[0], Ma.nnz, ptr(arrs[0]), ptr(arrs[2]), ptr(va), Mb.nnz, ptr(arrs[1]), ptr(arrs[3]), ptr(vb), C.byref(h)))
			self.kind, self.nnz = "affine", int(Ma.nnz + Mb.nnz)
			self._h = h
			self.set_parameter(getattr(A, "t", 0.0))
			A._device_ops = getattr(A, "_device_ops", [])
			import weakref

			A._device_ops.append(weakref.ref(self))
		elif hasattr(A, "matmat_device"):
			## GPU-resident plugin: A.matmat_device(X, Y, stream) receives two objects with
			## `__cuda_array_interface__` (shape (ncols, n), C order = column-major n x ncols) and a stream handle
			n, np_dt = self.shape[0], self.dtype
			self.error = None

			def _dcb(_user, dx, dy, nn, ncols, stream):
				try:
					A.matmat_device(_CudaArrayView(dx, nn * ncols, self, (ncols, nn), np_dt), _CudaArrayView(dy, nn * ncols, self, (ncols, nn), np_dt), stream)
					return 0
				except Exception as e:  # noqa: BLE001
					self.error = e
					return 1

			dcb = _capi.DEVICE_MATMAT_FN(_dcb)
			self._keep.append(dcb)
			check(L.slq_device_callback_create(self.ctx._h, dt, n, dcb, None, C.byref(h)))
			self.kind, self.nnz = "device_callback", 0
		else:
			if not hasattr(A, "matvec"):
				raise ValueError("Supplied object is missing 'matvec' attribute.")
			n, np_dt = self.shape[0], self.dtype
			self.error = None

			def _cb(_user, x, y):
				try:
					ct = C.c_double if np_dt == np.float64 else C.c_float
					xin = np.ctypeslib.as_array(C.cast(x, C.POINTER(ct)), shape=(n,))
					out = np.asarray(A.matvec(xin.copy())).astype(np_dt, copy=False).ravel()
					np.ctypeslib.as_array(C.cast(y, C.POINTER(ct)), shape=(n,))[:] = out[:n]
					return 0
				except Exception as e:  # noqa: BLE001
					self.error = e
					return 1

			cb = _capi.MATVEC_FN(_cb)
			self._keep.append(cb)
			check(L.slq_callback_create(self.ctx._h, dt, n, cb, None, C.byref(h)))
			self.kind, self.nnz = "callback", 0
		self._h = h

	def set_parameter(self, t: float) -> None:
		"""t of an affine operator A + t B (SparseEigenAffineOperator::set_parameter, eigen_operators.h:134-136)."""
		check(_capi.lib().slq_operator_set_parameter(self._h, float(t)))

	def matmat(self, X: np.ndarray) -> np.ndarray:
		X = np.asfortranarray(X.reshape(self.shape[1], -1), dtype=self.dtype)
		Y = np.empty_like(X, order="F")
		check(_capi.lib().slq_operator_matmat(self._h, ptr(X), X.shape[0], ptr(Y), Y.shape[0], X.shape[1]))
		return Y

	def close(self):
		if getattr(self, "_h", None):
			_capi.lib().slq_operator_destroy(self._h)
			self._h = None

	def __del__(self):
		try:
			self.close()
		except Exception:  # noqa: BLE001
			pass


class LanczosPlan:
	"""Workspace + state of one batched lock-step Lanczos run over `nprobes` probes."""

	def __init__(self, op: DeviceOperator, nprobes: int, deg: int, orth: int = 0, keep_basis: bool = False):
		self.op = op
		n = op.shape[0]
		self.nprobes = int(nprobes)
		self.deg = min(int(deg), n)
		self.orth = self.deg if orth < 0 or orth > self.deg else int(orth)
		self.keep_basis = bool(keep_basis)
		h = C.c_void_p()
		check(_capi.lib().slq_plan_create(op.ctx._h, op._h, self.nprobes, int(deg), int(orth), int(keep_basis), C.byref(h)))
		self._h = h

	@property
	def workspace_bytes(self) -> int:
		b = C.c_size_t()
		check(_capi.lib().slq_plan_workspace_bytes(self._h, C.byref(b)))
		return b.value

	def describe(self) -> dict:
		"""Panel geometry and launch sequence the library chose for this plan (slq_plan_describe)."""
		info = _capi.PlanInfo()
		check(_capi.lib().slq_plan_describe(self._h, C.byref(info)))
		d = {k: getattr(info, k) for k, _ in _capi.PlanInfo._fields_}
		d["sequence"] = {0: "sweeps", 1: "fused", 2: "fused_stored_u", 3: "sweeps_ring32", 4: "fused_gram"}[d["sequence"]]
		return d

	def set_probes(self, X: np.ndarray):
		X = np.asarray(X)
		X = X.reshape(-1, 1) if X.ndim == 1 else X
		assert X.shape == (self.op.shape[1], self.nprobes), f"probes must be {(self.op.shape[1], self.nprobes)}"
		X = np.asfortranarray(X, dtype=self.op.dtype)
		check(_capi.lib().slq_plan_set_probes(self._h, ptr(X), X.shape[0]))

	def set_probes_device(self, dptr: int):
		"""Probes already resident on this GPU: `dptr` is a device address of a contiguous column-major
		n x nprobes array of the operator dtype (e.g. torch_tensor.data_ptr() of a (nprobes, n) C-ordered
		tensor)."""
		check(_capi.lib().slq_plan_set_probes_device(self._h, C.c_void_p(int(dptr)), self.op.shape[0]))

	def generate_probes(self, pdf: str = "rademacher", seed: int = 0, probe_offset: int = 0):
		assert pdf in _capi.PDF_IDS, f"Invalid distribution '{pdf}' supplied."
		check(_capi.lib().slq_plan_generate_probes(self._h, _capi.PDF_IDS[pdf], int(seed), int(probe_offset)))

	def get_probes(self) -> np.ndarray:
		X = np.empty((self.op.shape[0], self.nprobes), dtype=self.op.dtype, order="F")
		check(_capi.lib().slq_plan_get_probes(self._h, ptr(X), X.shape[0]))
		return X

	def get_probes_into(self, out: "DeviceMatrix", o0: int):
		"""The current probes (as used: sphere draws have norm sqrt(n)) into columns [o0, o0 + nprobes) of `out`."""
		check(_capi.lib().slq_plan_get_probes_dmat(self._h, out._h, int(o0)))

	def run(self, rtol: float = 1e-8):
		rc = _capi.lib().slq_plan_run(self._h, float(rtol))
		if rc == _capi.SLQ_ECALLBACK and getattr(self.op, "error", None) is not None:
			raise self.op.error
		check(rc)

	def tridiag(self) -> tuple:
		"""(alpha, beta, steps): alpha/beta are (nprobes, deg+1) with beta[:, 0] = 0."""
		a = np.zeros((self.nprobes, self.deg + 1), dtype=self.op.dtype)
		b = np.zeros((self.nprobes, self.deg + 1), dtype=self.op.dtype)
		s = np.zeros(self.nprobes, dtype=np.int32)
		check(_capi.lib().slq_plan_get_tridiag(self._h, ptr(a), ptr(b), ptr(s)))
		return a, b, s

	def quadrature(self, fun="identity", return_rule: bool = False, **fun_kwargs):
		"""quad[i] = sum_k f(nodes[i,k]) weights[i,k] ||v_i||^2 ; optionally also (nodes, weights)."""
		fid, params = fun_spec(fun, **fun_kwargs)
		host_fun = fid is None
		want_rule = return_rule or host_fun
		quad = np.zeros(self.nprobes)
		nodes = np.zeros((self.nprobes, self.deg)) if want_rule else None
		weights = np.zeros((self.nprobes, self.deg)) if want_rule else None
		check(
			_capi.lib().slq_plan_quadrature(
				self._h, 0 if host_fun else fid, ptr(params), ptr(quad), ptr(nodes), ptr(weights)
			)
		)
		if host_fun:
			## arbitrary Python callables run on the host over the P x k nodes (operators.py:150);
			## the device returned sum(nodes*weights)*||v||^2, so ||v||^2 is recovered exactly
			ident = np.sum(nodes * weights, axis=1)
			vn2 = np.divide(quad, ident, out=np.zeros_like(quad), where=ident != 0)
			self._vnorm2 = vn2
			quad = np.array([np.sum(fun(nodes[i]) * weights[i]) for i in range(self.nprobes)]) * vn2
		return (quad, nodes, weights) if return_rule else quad

	def fun_action(self, fun="identity", **fun_kwargs) -> np.ndarray:
		"""Y[:, i] = f(A) x_i from the retained basis (needs keep_basis); built-in `fun` names only."""
		fid, params = fun_spec(fun, **fun_kwargs)
		assert fid is not None, "fun_action evaluates built-in function names on the device"
		Y = np.zeros((self.op.shape[0], self.nprobes), dtype=self.op.dtype, order="F")
		check(_capi.lib().slq_plan_fun_action(self._h, fid, ptr(params), ptr(Y), Y.shape[0]))
		return Y

	def fun_action_into(self, out: "DeviceMatrix", o0: int, fun="identity", **fun_kwargs):
		"""f(A) X of this run written straight into columns [o0, o0 + nprobes) of a DeviceMatrix."""
		fid, params = fun_spec(fun, **fun_kwargs)
		assert fid is not None
		check(_capi.lib().slq_plan_fun_action_dmat(self._h, fid, ptr(params), out._h, int(o0)))

	def basis(self, probe: int = 0) -> np.ndarray:
		Q = np.zeros((self.op.shape[0], self.deg), dtype=self.op.dtype, order="F")
		check(_capi.lib().slq_plan_get_basis(self._h, int(probe), ptr(Q), Q.shape[0]))
		return Q

	def profile_enable(self, enable: bool = True):
		check(_capi.lib().slq_plan_profile_enable(self._h, int(enable)))

	def profile_read(self, reset: bool = True) -> dict:
		pr = _capi.SlqProfile()
		check(_capi.lib().slq_plan_profile_read(self._h, C.byref(pr), int(reset)))
		return {k: {"ms": pr.ms[i], "launches": pr.launches[i]} for i, k in enumerate(_capi.KERNEL_CLASSES)}

	def sweep_columns(self, reset: bool = True) -> tuple:
		"""(read, offered): ring columns the deep-window update sweeps actually read against the ones their windows hold, summed over launches and panels - the
		sweep skips a column whose projection is below the reference's threshold for every probe of the panel (slq_plan_sweep_columns)."""
		a, b = C.c_uint64(), C.c_uint64()
		check(_capi.lib().slq_plan_sweep_columns(self._h, C.byref(a), C.byref(b), int(reset)))
		return int(a.value), int(b.value)

	def close(self):
		if getattr(self, "_h", None):
			_capi.lib().slq_plan_destroy(self._h)
			self._h = None

	def __del__(self):
		try:
			self.close()
		except Exception:  # noqa: BLE001
			pass


def quad_batch(
	op: DeviceOperator, X: Optional[np.ndarray], deg: int, orth: int = 0, fun="identity", rtol: float = 1e-8,
	nprobes: Optional[int] = None, pdf: str = "rademacher", seed: int = 0, probe_offset: int = 0,
	return_rule: bool = False, **fun_kwargs,
):  # fmt: skip
	"""One call for P probes (C-ABI slq_quad_batch): the batched form of the reference's per-probe
	Python loop (src/primate/operators.py:145-150)."""
	fid, params = fun_spec(fun, **fun_kwargs)
	assert fid is not None, "quad_batch takes built-in function names; use LanczosPlan.quadrature for callables"
	n = op.shape[0]
	deg_eff = min(int(deg), n)
	if X is not None:
		X = np.asarray(X)
		X = X.reshape(-1, 1) if X.ndim == 1 else X
		X = np.asfortranarray(X, dtype=op.dtype)
		if X.shape[0] != n:
			raise ValueError(f"X has {X.shape[0]} rows, the operator has {n}")
		nprobes = X.shape[1]
	quad = np.zeros(nprobes)
	nodes = np.zeros((nprobes, deg_eff)) if return_rule else None
	weights = np.zeros((nprobes, deg_eff)) if return_rule else None
	rc = _capi.lib().slq_quad_batch(
		op.ctx._h, op._h, ptr(X), n, _capi.PDF_IDS[pdf], int(seed), int(probe_offset), int(nprobes), int(deg),
		float(rtol), int(orth), fid, ptr(params), ptr(quad), ptr(nodes), ptr(weights),
	)  # fmt: skip
	if rc == _capi.SLQ_ECALLBACK and getattr(op, "error", None) is not None:
		raise op.error
	check(rc)
	return (quad, nodes, weights) if return_rule else quad


def fun_action_batch(op: DeviceOperator, X: np.ndarray, deg: int, orth: int = 0, fun="identity", rtol: float = 1e-8, **fun_kwargs) -> np.ndarray:
	"""Y = f(A) X for all columns of X in one call (C-ABI slq_fAv_batch): the batched form of
	`MatrixFunction._matvec` (src/primate/operators.py:102-124)."""
	fid, params = fun_spec(fun, **fun_kwargs)
	assert fid is not None, "fun_action_batch takes built-in function names"
	X = np.asarray(X)
	X = np.asfortranarray(X.reshape(-1, 1) if X.ndim == 1 else X, dtype=op.dtype)
	n = op.shape[0]
	if X.shape[0] != n:
		raise ValueError(f"X has {X.shape[0]} rows, the operator has {n}")
	Y = np.zeros((n, X.shape[1]), dtype=op.dtype, order="F")
	rc = _capi.lib().slq_fAv_batch(op.ctx._h, op._h, ptr(X), n, X.shape[1], int(deg), float(rtol), int(orth), fid, ptr(params), ptr(Y), n)
	if rc == _capi.SLQ_ECALLBACK and getattr(op, "error", None) is not None:
		raise op.error
	check(rc)
	return Y


def quadrature_batch(d: np.ndarray, e: np.ndarray, fun=None, ctx: Optional[Context] = None, **fun_kwargs):
	"""Gauss quadrature rules of a batch of Jacobi matrices on the device (slq_quadrature_batch).
	d, e: (nb, deg) with e[:, 0] ignored. Returns (nodes, weights) or, with `fun`, (quad, nodes, weights)."""
	ctx = ctx or default_context()
	d = np.ascontiguousarray(np.atleast_2d(d), dtype=np.float64)
	e = np.ascontiguousarray(np.atleast_2d(e), dtype=np.float64)
	assert d.shape == e.shape
	nb, deg = d.shape
	nodes, weights = np.zeros((nb, deg)), np.zeros((nb, deg))
	fid, params = (_capi.FUN_NONE, np.zeros(4)) if fun is None else fun_spec(fun, **fun_kwargs)
	quad = np.zeros(nb)
	check(_capi.lib().slq_quadrature_batch(ctx._h, nb, deg, ptr(d), ptr(e), fid, ptr(params), ptr(quad), ptr(nodes), ptr(weights)))
	return (nodes, weights) if fun is None else (quad, nodes, weights)


def eigh_tridiag_batch(d: np.ndarray, e: np.ndarray, vectors: bool = True, ctx: Optional[Context] = None):
	"""Eigenvalues (ascending) and, optionally, eigenvectors (columns) of a batch of symmetric tridiagonals on
	the device (slq_eigh_tridiag_batch). d, e: (nb, deg), e[:, 0] ignored."""
	ctx = ctx or default_context()
	d = np.ascontiguousarray(np.atleast_2d(d), dtype=np.float64)
	e = np.ascontiguousarray(np.atleast_2d(e), dtype=np.float64)
	assert d.shape == e.shape
	nb, deg = d.shape
	w = np.zeros((nb, deg))
	Z = np.zeros((nb, deg, deg)) if vectors else None
	check(_capi.lib().slq_eigh_tridiag_batch(ctx._h, nb, deg, ptr(d), ptr(e), ptr(w), ptr(Z)))
	return (w, Z) if vectors else w


def fttr_batch(theta: np.ndarray, alpha: np.ndarray, beta: np.ndarray, k: Optional[int] = None, ctx: Optional[Context] = None) -> np.ndarray:
	"""FTTR quadrature weights on the device (slq_fttr_batch); rows are independent rules."""
	ctx = ctx or default_context()
	theta = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
	alpha = np.ascontiguousarray(np.atleast_2d(alpha), dtype=np.float64)
	beta = np.ascontiguousarray(np.atleast_2d(beta), dtype=np.float64)
	nb, kk = theta.shape
	k = kk if k is None else int(k)
	assert alpha.shape == beta.shape and alpha.shape[0] == nb and k <= kk
	w = np.zeros((nb, k))
	check(_capi.lib().slq_fttr_batch(ctx._h, nb, alpha.shape[1], k, ptr(np.ascontiguousarray(theta[:, :k])), ptr(alpha), ptr(beta), ptr(w)))
	return w


class DiagAccumulator:
	"""Device-resident numer / denom / running-mean accumulators of the diagonal estimator."""

	def __init__(self, n: int, ctx: Optional[Context] = None):
		self.ctx = ctx or default_context()
		self.n = int(n)
		h = C.c_void_p()
		check(_capi.lib().slq_diag_create(self.ctx._h, self.n, C.byref(h)))
		self._h = h

	def update(self, plan: LanczosPlan, fun="identity", **fun_kwargs):
		fid, params = fun_spec(fun, **fun_kwargs)
		assert fid is not None, "the device diagonal path takes built-in function names"
		check(_capi.lib().slq_diag_update(self._h, plan._h, fid, ptr(params)))

	def get(self) -> tuple:
		"""(numer, denom, running_mean, count)."""
		nu, de, rm = np.zeros(self.n), np.zeros(self.n), np.zeros(self.n)
		cnt = C.c_int64()
		check(_capi.lib().slq_diag_get(self._h, ptr(nu), ptr(de), ptr(rm), C.byref(cnt)))
		return nu, de, rm, cnt.value

	def close(self):
		if getattr(self, "_h", None):
			_capi.lib().slq_diag_destroy(self._h)
			self._h = None

	def __del__(self):
		try:
			self.close()
		except Exception:  # noqa: BLE001
			pass


class _CudaArrayView:
	"""Flat fp64 device array described by the CUDA array interface (v2); keeps its owner alive."""

	def __init__(self, dptr: int, count: int, owner, shape=None, dtype=np.float64, device: Optional[int] = None):
		self._owner = owner
		## the GPU that owns the memory (the owner's Context); consumers must not assume torch's current device
		ctx = getattr(owner, "ctx", None)
		self.device_index = int(device) if device is not None else (int(ctx.device) if ctx is not None else None)
		self.__cuda_array_interface__ = {
			"shape": tuple(int(v) for v in shape) if shape is not None else (int(count),),
			"typestr": "<f8" if np.dtype(dtype) == np.float64 else "<f4", "data": (int(dptr), False), "version": 2,
		}  # fmt: skip


class DeviceMatrix:
	"""Column-major n x cols fp64 matrix on the GPU with the two tall-skinny products of the
	exchangeable estimators (slq_dmat_*: fp64 MFMA)."""

	def __init__(self, n: int, cols: int, ctx: Optional[Context] = None):
		self.ctx = ctx or default_context()
		self.n, self.cols = int(n), int(cols)
		h = C.c_void_p()
		check(_capi.lib().slq_dmat_create(self.ctx._h, self.n, self.cols, C.byref(h)))
		self._h = h

	def set(self, c0: int, X: np.ndarray):
		X = np.asfortranarray(np.asarray(X, dtype=np.float64).reshape(self.n, -1))
		check(_capi.lib().slq_dmat_set(self._h, int(c0), X.shape[1], ptr(X), self.n))

	def get(self, c0: int = 0, nc: Optional[int] = None) -> np.ndarray:
		nc = self.cols - c0 if nc is None else int(nc)
		X = np.empty((self.n, nc), order="F")
		check(_capi.lib().slq_dmat_get(self._h, int(c0), nc, ptr(X), self.n))
		return X

	def col_ptr(self, c0: int) -> int:
		p = C.c_void_p()
		check(_capi.lib().slq_dmat_ptr(self._h, int(c0), C.byref(p)))
		return p.value

	def generate(self, c0: int, nc: int, pdf: str = "rademacher", seed: int = 0, probe_offset: int = 0):
		"""Isotropic probes into columns [c0, c0+nc): the device Philox stream, probe ids probe_offset + column."""
		assert pdf in _capi.PDF_IDS, f"Invalid distribution '{pdf}' supplied."
		check(_capi.lib().slq_dmat_generate(self._h, int(c0), int(nc), _capi.PDF_IDS[pdf], int(seed), int(probe_offset)))

	def copy_from(self, d0: int, src: "DeviceMatrix", s0: int, nc: int):
		"""self[:, d0:d0+nc] = src[:, s0:s0+nc] (device to device)."""
		check(_capi.lib().slq_dmat_copy(self._h, int(d0), src._h, int(s0), int(nc)))

	def copy_rows_from(self, d0: int, dr0: int, src: "DeviceMatrix", s0: int, sr0: int, nrows: int, nc: int):
		"""self[dr0:dr0+nrows, d0:d0+nc] = src[sr0:sr0+nrows, s0:s0+nc] (device to device; the matrices may differ in height)."""
		check(_capi.lib().slq_dmat_copy_rows(self._h, int(d0), int(dr0), src._h, int(s0), int(sr0), int(nrows), int(nc)))

	def cuda_array(self, c0: int, nc: int):
		"""Columns [c0, c0+nc) as a flat object with `__cuda_array_interface__` (zero-copy view for
		torch.as_tensor(..., device="cuda"): the RCCL collectives of primate_amd.distributed)."""
		assert 0 <= c0 and nc > 0 and c0 + nc <= self.cols
		return _CudaArrayView(self.col_ptr(c0), self.n * int(nc), self)

	def tn(self, a0: int, ma: int, B: "DeviceMatrix", b0: int, mb: int) -> np.ndarray:
		"""self[:, a0:a0+ma].T @ B[:, b0:b0+mb]  (ma x mb, on the host)."""
		out = np.zeros((ma, mb))
		check(_capi.lib().slq_dmat_gemm_tn(self._h, int(a0), int(ma), B._h, int(b0), int(mb), ptr(out)))
		return out

	def add_product(self, o0: int, A: "DeviceMatrix", a0: int, Cm: np.ndarray, alpha: float = 1.0, beta: float = 1.0):
		"""self[:, o0:o0+mb] = beta * self[:, o0:o0+mb] + alpha * A[:, a0:a0+ma] @ Cm."""
		Cm = np.ascontiguousarray(Cm, dtype=np.float64)
		ma, mb = Cm.shape
		check(_capi.lib().slq_dmat_gemm_nn(self._h, int(o0), A._h, int(a0), ma, ptr(Cm), mb, float(alpha), float(beta)))

	def close(self):
		if getattr(self, "_h", None):
			_capi.lib().slq_dmat_destroy(self._h)
			self._h = None

	def __del__(self):
		try:
			self.close()
		except Exception:  # noqa: BLE001
			pass
