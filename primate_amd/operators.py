"""`MatrixFunction`: the SLQ operator x -> f(A) x / x^T f(A) x with the reference's interface
(src/primate/operators.py:36-161), evaluated by libslq on the MI355X.

What changes under the hood:
  * `.quad(X)` advances ALL columns of X in lock-step in one device run (the reference loops over
    columns in Python and crosses its FFI once per probe, operators.py:145-150);
  * the operator is uploaded once and stays resident (the reference copies it >= 3 times per probe,
    SURVEY.md §8a a6);
  * built-in spectral functions are reduced on the device; Python callables are applied on the
    host to the (P, deg) nodes.
"""

from __future__ import annotations

from typing import Any, Callable, Optional, Union

import numpy as np
from scipy.sparse.linalg import LinearOperator

from . import engine
from .lanczos import _as_device_operator
from .special import builtin_spec, param_callable

F64: np.dtype = np.dtype("float64")


def _has_product(A: Any) -> bool:
	return any(hasattr(A, a) for a in ("__matmul__", "matmul", "dot", "matvec"))


def is_valid_operator(A: Union[np.ndarray, LinearOperator]) -> np.dtype:
	"""Checks of operators.py:15-23; returns the floating dtype."""
	assert _has_product(A), "Invalid operator; must have an overloaded 'matvec' or 'matmul' method"
	assert hasattr(A, "shape") and len(A.shape) >= 2, "Operator must be at least two dimensional."
	assert A.shape[0] == A.shape[1], "This function only works with square, symmetric matrices!"
	f_dtype = (A @ np.zeros(A.shape[1])).dtype if not hasattr(A, "dtype") else A.dtype
	assert np.dtype(f_dtype).type in {np.float32, np.float64}, "Only 32- or 64-bit floats are supported."
	return np.dtype(f_dtype)


def is_linear_op(A: Any) -> bool:
	"""operators.py:26-33."""
	return bool(_has_product(A) and hasattr(A, "shape") and len(A.shape) >= 2 and A.shape[0] == A.shape[1])


class MatrixFunction(LinearOperator):
	"""Linear operator approximating f(A) by degree-`deg` Lanczos (operators.py:36-151).

	Parameters match the reference: A (ndarray / sparse / LinearOperator), fun (name or callable),
	deg, orth (number of most recent Lanczos vectors to re-orthogonalise against; <0 or >deg means
	deg), dtype, and **kwargs for the named function (e.g. t=, a=, b=, threshold=).

	stale_ring (extra, default False): reproduce the reference's `quad` bit of history — it never
	clears its Lanczos ring `_Q` between probes (operators.py:138-148), so with orth > 0 probe j's
	first orth-1 reorthogonalisation sweeps also project against probe j-1's last Lanczos vectors.
	True runs `quad` one probe at a time through the single-vector drop-in entry with a persistent
	ring, exactly like the reference (and as slowly: one device run per probe). The default False is
	the lock-step batched path, which starts every probe from a clean ring — what the reference's own
	`_matvec` enforces (operators.py:116) and what its first probe always sees.
	"""

	def __init__(
		self, A, fun: Union[str, Callable, None] = None, deg: int = 20, orth: int = 3, dtype: np.dtype = F64,
		stale_ring: bool = False, **kwargs,
	) -> None:  # fmt: skip
		assert is_linear_op(A), "Invalid operator `A`; must be dim=2 symmetric operator with defined matvec"
		assert deg >= 2, "Degree must be >= 2"
		self.shape = A.shape
		self.dtype = np.dtype(dtype)
		fun = (lambda x: x) if fun is None else fun
		self._builtin = builtin_spec(fun, **kwargs) if isinstance(fun, str) else None
		self.fun = param_callable(fun, **kwargs) if isinstance(fun, str) else fun
		self._deg = min(deg, A.shape[0])
		self._rtol = 1e-8
		self._orth = self._deg if orth < 0 or orth > self._deg else orth
		self._A = A
		self._op = _as_device_operator(A, dtype=self.dtype)
		self._plans: dict = {}
		self._stale_ring = bool(stale_ring)
		if self._stale_ring:  # the reference's persistent buffers (operators.py:69-77)
			self._alpha = np.zeros(self._deg + 1, dtype=self.dtype)
			self._beta = np.zeros(self._deg + 1, dtype=self.dtype)
			self._Q = np.zeros((A.shape[0], self._deg), dtype=self.dtype, order="F")
		self._nodes = np.zeros(self._deg, dtype=self.dtype)
		self._weights = np.zeros(self._deg, dtype=self.dtype)

	@property
	def degree(self) -> int:
		return self._deg

	@property
	def fun(self) -> Callable:
		return self._fun

	@fun.setter
	def fun(self, value: Callable) -> None:
		assert callable(value), "Function must be callable."
		out = value(np.ones(self.shape[1]))
		assert isinstance(out, np.ndarray), "Function must return array-like"
		assert out.shape[-1] == self.shape[0], "Last dimension of output must match number of rows."
		self._fun = value
		spec = getattr(value, "_slq_builtin", None)
		if spec is not None:
			self._builtin = spec
		elif getattr(self, "_fun_initialised", False):
			self._builtin = None
		self._fun_initialised = True

	def _adjoint(self):
		return self

	def _plan(self, nprobes: int, keep_basis: bool) -> engine.LanczosPlan:
		key = (nprobes, keep_basis)
		if key not in self._plans:
			## one cached plan per shape class; older ones are released to bound device memory
			for k in [k for k in self._plans if k[1] == keep_basis]:
				self._plans.pop(k).close()
			self._plans[key] = engine.LanczosPlan(self._op, nprobes, self._deg, self._orth, keep_basis=keep_basis)
		return self._plans[key]

	def quad(self, x: np.ndarray) -> np.ndarray:
		"""x^T f(A) x for every column of x by Lanczos quadrature (operators.py:126-151)."""
		x = np.asarray(x).astype(self.dtype, copy=False)
		x = np.atleast_2d(x).T if x.ndim == 1 else x
		if self._stale_ring:
			return self._quad_reference_ring(x)
		plan = self._plan(x.shape[1], False)
		plan.set_probes(x)
		plan.run(self._rtol)
		if self._builtin is not None:
			name, kw = self._builtin
			y, nodes, weights = plan.quadrature(name, return_rule=True, **kw)
		else:
			y, nodes, weights = plan.quadrature(self._fun, return_rule=True)
		## the reference leaves the last probe's rule in self._nodes/_weights (operators.py:149)
		self._nodes[:], self._weights[:] = nodes[-1], weights[-1]
		return y

	def quad_generated(self, nprobes: int, pdf: str = "rademacher", seed: int = 0, probe_offset: int = 0) -> np.ndarray:
		"""v_i^T f(A) v_i for `nprobes` probes DRAWN ON THE DEVICE (Philox stream of probe ids probe_offset..;
		slq_plan_generate_probes): the throughput form of `quad`, with no n x nprobes host array at all."""
		assert not self._stale_ring, "device-drawn probes start from a clean ring"
		plan = self._plan(int(nprobes), False)
		plan.generate_probes(pdf, seed=int(seed), probe_offset=int(probe_offset))
		plan.run(self._rtol)
		if self._builtin is not None:
			name, kw = self._builtin
			y, nodes, weights = plan.quadrature(name, return_rule=True, **kw)
		else:
			y, nodes, weights = plan.quadrature(self._fun, return_rule=True)
		self._nodes[:], self._weights[:] = nodes[-1], weights[-1]
		return y

	def _quad_reference_ring(self, x: np.ndarray) -> np.ndarray:
		"""The reference loop verbatim in structure (operators.py:145-150): per column, the native
		single-vector Lanczos on the persistent alpha/beta/Q, then the Gauss rule, then the sum."""
		from .integrate import quadrature
		from .lanczos import _native_lanczos

		y = np.zeros(x.shape[1])
		for j in range(x.shape[1]):
			xc = np.ascontiguousarray(x[:, j])
			_native_lanczos(self._op, xc, self._deg, self._rtol, self._orth, self._alpha, self._beta, self._Q)
			quadrature(self._alpha[: self._deg], self._beta[: self._deg], deg=self._deg, nodes=self._nodes, weights=self._weights)
			y[j] = np.sum(self._fun(self._nodes) * self._weights, axis=-1) * np.linalg.norm(xc) ** 2
		return y

	def _matvec(self, x: np.ndarray) -> np.ndarray:
		"""f(A) x ~= ||x|| Q Y (f(theta) * Y[0,:]) with the full Lanczos basis (operators.py:102-124)."""
		return self._matmat(np.asarray(x).reshape(-1, 1))

	def _matmat(self, X: np.ndarray) -> np.ndarray:
		X = np.asarray(X).astype(self.dtype, copy=False)
		plan = self._plan(X.shape[1], True)
		plan.set_probes(X)
		plan.run(self._rtol)
		if self._builtin is not None:
			name, kw = self._builtin
			return plan.fun_action(name, **kw)
		## Python callable: it can only be evaluated on the host, so the k x k eigenvectors of every probe's T come
		## back from the device eigensolver (one batched call) and the basis combination is done per probe
		a, b, _ = plan.tridiag()
		rws, Ys = engine.eigh_tridiag_batch(a[:, : self._deg], b[:, : self._deg], vectors=True, ctx=self._op.ctx)
		out = np.zeros((self.shape[0], X.shape[1]), dtype=self.dtype, order="F")
		for i in range(X.shape[1]):
			rw, Y = rws[i], Ys[i]
			coef = Y @ (np.ravel(self._fun(rw)) * Y[0, :])
			out[:, i] = np.linalg.norm(X[:, i]) * (plan.basis(i) @ coef)
		return out


def matrix_function(A, fun: Optional[Callable] = None, v: Optional[np.ndarray] = None, deg: int = 20):
	"""operators.py:155-161."""
	M = MatrixFunction(A, fun=fun, deg=deg)
	return M if v is None else M._matvec(v)


class GramOperator(LinearOperator):
	"""x -> A^T (A x) for a rectangular (sparse) A: the symmetric positive semidefinite operator behind the singular
	values of A (rank by `numrank`, nuclear / Schatten norms by `sqrt` of its spectrum, ...). The reference's native
	`SparseEigenLinearOperator<F, true>` (src/primate/include/eigen_operators.h:57-72), which its Python module never binds
	(`_lanczos.cpp:104-111`). On the device it is two panel SpMMs per Lanczos step (`slq_csr_gram_create`); the host
	methods are the plugin surface of every other operator."""

	_slq_kind = "gram"

	def __init__(self, A, dtype=None):
		import scipy.sparse as sp

		self.A = sp.csr_matrix(A)
		self.dtype = np.dtype(dtype if dtype is not None else (self.A.dtype if self.A.dtype in (np.float32, np.float64) else np.float64))
		self.A = self.A.astype(self.dtype)
		self.shape = (self.A.shape[1], self.A.shape[1])

	def _matvec(self, x: np.ndarray) -> np.ndarray:
		return self.A.T @ (self.A @ np.asarray(x).ravel())

	def _matmat(self, X: np.ndarray) -> np.ndarray:
		return self.A.T @ (self.A @ X)

	def _adjoint(self):
		return self


class AffineOperator(LinearOperator):
	"""A + t B for two sparse n x n matrices and a scalar t that changes between solves: the reference's native
	`SparseEigenAffineOperator` (src/primate/include/eigen_operators.h:106-137; unbound in its Python module). On the device
	both live on their union pattern as ONE CSR operator - every fused pass applies - and `set_parameter(t)` rewrites its
	values in place (`slq_csr_affine_create`, `slq_operator_set_parameter`), also for operators already wrapped in a
	`MatrixFunction`."""

	_slq_kind = "affine"

	def __init__(self, A, B, t: float = 0.0, dtype=None):
		import scipy.sparse as sp

		self.A, self.B = sp.csr_matrix(A), sp.csr_matrix(B)
		assert self.A.shape == self.B.shape and self.A.shape[0] == self.A.shape[1], "A and B must be square and of equal shape"
		self.dtype = np.dtype(dtype if dtype is not None else (self.A.dtype if self.A.dtype in (np.float32, np.float64) else np.float64))
		self.shape = self.A.shape
		self.t = float(t)

	def set_parameter(self, t: float) -> None:
		self.t = float(t)
		for ref in getattr(self, "_device_ops", []):
			op = ref()
			if op is not None and getattr(op, "_h", None):
				op.set_parameter(self.t)

	def _matvec(self, x: np.ndarray) -> np.ndarray:
		x = np.asarray(x).ravel()
		return (self.A @ x + self.t * (self.B @ x)).astype(self.dtype, copy=False)

	def _matmat(self, X: np.ndarray) -> np.ndarray:
		return (self.A @ X + self.t * (self.B @ X)).astype(self.dtype, copy=False)

	def _adjoint(self):
		return self


class Toeplitz(LinearOperator):
	"""Matrix-free symmetric-or-not Toeplitz operator with first column `c` and first row `r` (default r = c),
	applied by circulant embedding and two FFTs of length 2n (src/primate/operators.py:165-183). A host-side
	plugin: `lanczos`/`MatrixFunction` reach it through the callback operator (`slq_callback_create`)."""

	def __init__(self, c: np.ndarray, r: Optional[np.ndarray] = None, dtype: np.dtype = np.float64):
		self.c = np.array(c)
		self.r = np.array(c if r is None else r)
		n = len(self.c)
		## first column of the 2n x 2n circulant that contains T in its leading block: [c, 0, r_{n-1}, ..., r_1]
		self._symbol = np.fft.fft(np.concatenate((self.c, [0], self.r[:0:-1])))
		self._pad = np.zeros(2 * n)
		self.shape = (n, n)
		self.dtype = np.dtype(dtype)

	def _matvec(self, x: np.ndarray) -> np.ndarray:
		n = self.shape[0]
		assert np.size(x) == n, f"Invalid shape of input vector 'x'; must have length {n}"
		self._pad[:n] = np.ravel(x)
		y = np.fft.ifft(self._symbol * np.fft.fft(self._pad))
		return np.real(y[:n]).astype(self.dtype)


class TorchOperator(LinearOperator):
	"""A symmetric operator given as a function on GPU-resident torch tensors, `fn(X) -> A X` for X of shape
	(n, b) on this process's GPU. Used as `A` in `lanczos` / `MatrixFunction` / `hutch`, its products run inside
	the device Lanczos loop on libslq's stream without a host round trip (`slq_device_callback_create`): the
	GPU-native form of the reference's `LinearOperator` plugin surface (src/primate/operators.py:15-33)."""

	def __init__(self, fn: Callable, n: int, dtype=np.float64, device: Optional[int] = None):
		self.fn, self.shape, self.dtype = fn, (int(n), int(n)), np.dtype(dtype)
		## the GPU `fn` computes on: LOCAL_RANK under torchrun (what `engine.Context` picks too), else torch's current
		## device at construction time. The device products below do NOT consult torch.cuda.current_device().
		self.device = device

	def _device(self):
		import os

		import torch

		if self.device is None:
			self.device = int(os.environ["LOCAL_RANK"]) if "LOCAL_RANK" in os.environ else torch.cuda.current_device()
		return torch.device("cuda", int(self.device))

	def matmat_device(self, X, Y, stream) -> None:
		import torch

		## X, Y and the stream belong to the GPU of the DeviceOperator's Context, which the views carry
		owner = getattr(X, "device_index", None)
		dev = torch.device("cuda", int(owner)) if owner is not None else self._device()
		if self.device is not None and owner is not None:
			assert int(self.device) == int(owner), f"TorchOperator on cuda:{self.device} driven by a Context on cuda:{owner}"
		with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
			x, y = torch.as_tensor(X, device=dev), torch.as_tensor(Y, device=dev)  # (b, n) views of column-major n x b
			y.copy_(self.fn(x.T).T)

	def _matmat(self, X: np.ndarray) -> np.ndarray:
		import torch

		dev = self._device()
		return self.fn(torch.as_tensor(np.ascontiguousarray(X), device=dev).to(torch.float64 if self.dtype == np.float64 else torch.float32)).cpu().numpy()

	def _matvec(self, x: np.ndarray) -> np.ndarray:
		return self._matmat(np.asarray(x).reshape(-1, 1)).ravel()
