"""Shared fixtures. `-m "not gpu"` runs here on CPU; `-m gpu` runs on the MI355X box."""

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
	sys.path.insert(0, str(ROOT))


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
	"""A fresh checkout has no built libraries (they are git-ignored): build them once, as the driver's
	`__graft_entry__.build()` does. Building is all this does — a missing hipcc still fails the tests loudly."""
	import shutil

	from primate_amd import _capi

	if not _capi.LIB_PATH.exists() and (shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists()):
		import __graft_entry__ as G

		G.build_libslq()


@pytest.fixture(scope="session")
def golden():
	"""Vectors captured from the reference's own Python by tests/golden/make_golden.py."""
	return np.load(ROOT / "tests" / "golden" / "slq_golden.npz")


@pytest.fixture(scope="session")
def oracle():
	from oracle import oracle as O

	O.build()
	return O


def laplacian_2d(m: int, dtype=np.float64):
	"""2D 5-point Dirichlet Laplacian on an m x m grid (BASELINE.json configs[1] at size m)."""
	import scipy.sparse as sp

	T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))
	A = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsr().astype(dtype)
	A.sort_indices()
	return A


def laplacian_3d(m: int, dtype=np.float64):
	"""3D 7-point Dirichlet Laplacian on an m^3 grid (the north_star's nnz≈7M variant at m=100)."""
	import scipy.sparse as sp

	T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))
	I = sp.identity(m)
	A = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr().astype(dtype)
	A.sort_indices()
	return A
