"""Wall time of a one-shot call in a warm process: hutch(MatrixFunction(A), 256 device-drawn probes) from a SciPy matrix the library has not seen
(a fresh diagonal shift each time: the operator cache cannot serve it), against the same call's Lanczos batch alone.
    python scripts/time_oneshot.py        (SLQ_DEVICE_BUILD=0: the r03 host-side build)"""
import sys, time
from pathlib import Path
import numpy as np, scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d, laplacian_3d
from primate_amd import engine as eng
from primate_amd.operators import MatrixFunction
from primate_amd.trace import hutch

ctx = eng.default_context()
for name, A0 in (("lap2d_1000", laplacian_2d(1000)), ("lap3d_100", laplacian_3d(100))):
	n = A0.shape[0]
	hutch(MatrixFunction(A0, fun="log", deg=30), pdf="device:rademacher", converge="count", count=256, seed=1)  # warm: code object, plans' kernels
	for rep in range(3):
		A = (A0 + sp.identity(n, format="csr") * (1e-3 * (rep + 1))).tocsr()
		A.sort_indices()
		ctx.synchronize()
		t0 = time.perf_counter()
		M = MatrixFunction(A, fun="log", deg=30)
		t1 = time.perf_counter()
		est = hutch(M, pdf="device:rademacher", converge="count", count=256, seed=1234)
		ctx.synchronize()
		t2 = time.perf_counter()
		est2 = hutch(M, pdf="device:rademacher", converge="count", count=256, seed=1234)
		ctx.synchronize()
		t3 = time.perf_counter()
		print(f"{name}: MatrixFunction {1e3*(t1-t0):.1f} ms + first hutch {1e3*(t2-t1):.1f} ms = {1e3*(t2-t0):.1f} ms; the same hutch again {1e3*(t3-t2):.1f} ms; estimate {est:.6e} / {est2:.6e}", flush=True)
		del M
