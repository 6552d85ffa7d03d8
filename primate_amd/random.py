"""Isotropic probe vectors on the HOST, with the reference's exact NumPy stream order (parity
mode). Throughput runs draw probes on the device instead (LanczosPlan.generate_probes: Philox).

Contract reproduced from src/primate/random.py:22-41,47-80 and pinned by tests/golden:
  * output is float64, Fortran-ordered, filled in memory order by ONE Generator, so a batched
    (n, b) draw equals b sequential (n, 1) draws (reference tests/test_random.py:23-39);
  * rademacher = floor(2u)*2-1 with u = rng.random(); normal = rng.standard_normal();
    sphere = sqrt(n) g/||g|| per column.
"""

from __future__ import annotations

import concurrent.futures
import multiprocessing
import os
from typing import Callable, Optional, Union

import numpy as np

_ISO_DISTRIBUTIONS = {"rademacher": "rademacher", "normal": "normal", "sphere": "sphere", "signs": "rademacher", "gaussian": "normal"}


_PARALLEL_MIN = 1 << 22  # elements; below this one thread is as fast


def _signs_from_uniform(rng: np.random.Generator, flat: np.ndarray) -> None:
	rng.random(out=flat)
	np.multiply(flat, 2, out=flat)
	np.floor(flat, out=flat)
	np.multiply(flat, 2, out=flat)
	np.subtract(flat, 1, out=flat)


def _fill_rademacher_parallel(rng: np.random.Generator, out: np.ndarray) -> bool:
	"""The SAME values as `_signs_from_uniform(rng, out)` drawn by several threads: `Generator.random` consumes
	exactly one 64-bit PCG64 output per double, so a worker whose generator is a copy of `rng` advanced by `a` steps
	produces elements [a, b) of the stream. `rng` itself is advanced by the total afterwards. Returns False (nothing
	drawn) when the shortcut does not apply: other bit generators, non-contiguous or small outputs."""
	bg = rng.bit_generator
	if not isinstance(bg, np.random.PCG64) or out.size < _PARALLEL_MIN or out.dtype != np.float64:
		return False
	flat = out.reshape(-1) if out.flags["C_CONTIGUOUS"] else (out.T.reshape(-1) if out.flags["F_CONTIGUOUS"] else None)
	if flat is None or not np.shares_memory(flat, out):
		return False
	st = bg.state
	if st.get("has_uint32", 0):  # a buffered 32-bit half would be consumed first by integer draws only; doubles ignore it
		pass
	workers = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else multiprocessing.cpu_count()))
	if workers == 1:
		return False
	bounds = np.linspace(0, flat.size, workers + 1).astype(np.int64)

	def work(a: int, b: int) -> None:
		g = np.random.Generator(np.random.PCG64())
		g.bit_generator.state = st
		g.bit_generator.advance(int(a))
		_signs_from_uniform(g, flat[a:b])

	with concurrent.futures.ThreadPoolExecutor(workers) as ex:
		for f in [ex.submit(work, int(a), int(b)) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]:
			f.result()
	bg.advance(int(flat.size))
	return True


def _fill(rng: np.random.Generator, pdf: str, out: np.ndarray) -> None:
	if pdf == "rademacher":
		if not _fill_rademacher_parallel(rng, out):
			_signs_from_uniform(rng, out)
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


class Isotropic:
	"""Batch filler with the reference's interface (src/primate/random.py:100-142): `values` is an (n, b)
	Fortran-ordered array that every `fill()` redraws in place.

	Host mode (default) keeps the reference's stream: one generator for `threads == 1`, otherwise `threads` child
	generators spawned from the seed, child i filling the i-th block of ceil(b / threads) columns, concurrently.
	`device=True` draws on the GPU instead (Philox stream of `slq_dmat_generate`, probe ids advancing by b per
	fill): `device_values` is the resident `DeviceMatrix`, `values` downloads it."""

	def __init__(self, size: tuple, pdf: str = "signs", seed: Union[int, np.random.SeedSequence, np.random.Generator, None] = None,
	             threads: Optional[int] = None, device: bool = False):  # fmt: skip
		assert pdf in _ISO_DISTRIBUTIONS, f"Invalid distribution '{pdf}' supplied."
		self.pdf = _ISO_DISTRIBUTIONS[pdf]
		self.shape = tuple(size)
		self.threads = multiprocessing.cpu_count() if threads is None else int(threads)
		self._device = bool(device)
		if self._device:
			from .engine import DeviceMatrix

			self._seed = int(seed) if isinstance(seed, (int, np.integer)) else int(np.random.default_rng(seed).integers(0, 2**62))
			self._drawn = 0
			self.device_values = DeviceMatrix(self.shape[0], self.shape[1])
			return
		rng = np.random.default_rng(seed)
		self._random_generators = [rng] if self.threads == 1 else rng.spawn(self.threads)
		self.executor = concurrent.futures.ThreadPoolExecutor(self.threads)
		self._values = np.zeros(self.shape, order="F")  # column blocks are contiguous: safe to fill concurrently
		self.step = int(np.ceil(self.shape[1] / self.threads))
		## (generator, column block) pairs, fixed for the life of the filler
		self._blocks = [
			(g, self._values[:, i * self.step : (i + 1) * self.step])
			for i, g in enumerate(self._random_generators)
			if i * self.step < self.shape[1]
		]

	@property
	def values(self) -> np.ndarray:
		return self.device_values.get() if self._device else self._values

	def fill(self) -> None:
		if self._device:
			self.device_values.generate(0, self.shape[1], self.pdf, seed=self._seed, probe_offset=self._drawn)
			self._drawn += self.shape[1]
			return
		if len(self._blocks) == 1:
			_fill(self._blocks[0][0], self.pdf, self._blocks[0][1])
			return
		jobs = [self.executor.submit(_fill, g, self.pdf, view) for g, view in self._blocks]
		for j in jobs:
			j.result()

	def __del__(self):
		ex = getattr(self, "executor", None)
		if ex is not None:
			ex.shutdown(False)


def symmetric(n: int, dist: str = "normal", pd: bool = False, ew: Optional[np.ndarray] = None, seed: Union[int, np.random.Generator, None] = None) -> np.ndarray:
	"""Random symmetric n x n matrix with prescribed eigenvalues (src/primate/random.py:145-181; the operator of
	BASELINE.json configs[0] and of the reference's tests). The eigenvectors are the Q factor of a random symmetric
	matrix whose strict upper triangle is drawn first (`dist`), then its diagonal (uniform); the eigenvalues `ew`
	default to uniform draws on [-1, 1], or [0, 1] with `pd`. The draw order is the reference's, so equal seeds give
	equal matrices."""
	rng = np.random.default_rng(seed)
	if dist not in ("uniform", "normal"):
		raise ValueError(f"Invalid distribution {dist} supplied")
	off = rng.uniform(size=n * (n - 1) // 2) if dist == "uniform" else rng.normal(size=n * (n - 1) // 2)
	B = np.zeros((n, n))
	B[np.triu_indices(n, k=1)] = off  # row-major upper triangle: scipy's squareform layout
	B += B.T
	B[np.diag_indices(n)] = rng.random(n)
	Q = np.linalg.qr(B)[0]
	ew = rng.uniform(size=n, low=0.0 if pd else -1.0, high=1.0) if ew is None else np.atleast_1d(ew)
	A = (Q * ew) @ Q.T
	return (A + A.T) / 2


def haar(n: int, ew: Optional[np.ndarray] = None, seed: Union[int, np.random.Generator, None] = None) -> np.ndarray:
	"""U diag(ew) U^T with U drawn uniformly from the orthogonal group O(n) (src/primate/random.py:184-200);
	`ew` defaults to uniform draws on [-1, 1], taken before U as in the reference."""
	from scipy.stats import ortho_group

	rng = np.random.default_rng(seed)
	og = ortho_group(n, seed=rng)
	ew = rng.uniform(size=n, low=-1.0, high=1.0) if ew is None else np.atleast_1d(ew)
	assert len(ew) == n, "Number of eigenvalues must be <= `n`"
	U = og.rvs()
	return (U * ew) @ U.T
