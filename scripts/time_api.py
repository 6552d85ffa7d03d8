"""End-to-end wall time of the public API on configs[1] (operator creation, plan creation, probes, Lanczos,
quadrature, Python overheads): what a user of `hutch(MatrixFunction(A))` sees."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d
from primate_amd.operators import MatrixFunction
from primate_amd.trace import hutch
A = laplacian_2d(1000)
t = time.perf_counter(); M = MatrixFunction(A, fun="log", deg=30, orth=3); t_mf = time.perf_counter() - t
for label, kw in [("device probes", dict(pdf="device:rademacher")), ("NumPy probes (reference stream)", dict())]:
    for rep in range(2):
        t = time.perf_counter(); est = hutch(M, converge="count", count=256, seed=1234, batch=256, **kw); dt = time.perf_counter() - t
        print(f"{label}: call {rep}: {dt:.3f} s, estimate {est:.4f}", flush=True)
print(f"MatrixFunction(A) creation {t_mf:.3f} s; closed form 1166809.9081")
