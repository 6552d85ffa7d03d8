import sys, time
from pathlib import Path
import numpy as np, scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d, laplacian_3d
from primate_amd import engine as eng
ctx = eng.default_context()
for name, A in [("lap2d_1000", laplacian_2d(1000)), ("lap3d_100", laplacian_3d(100)), ("lap3d_126_f32", laplacian_3d(126, np.float32))]:
    t = time.perf_counter(); op = eng.DeviceOperator(A); ctx.synchronize(); dt = time.perf_counter() - t
    t = time.perf_counter(); plan = eng.LanczosPlan(op, 256, 30, 3); ctx.synchronize(); dp = time.perf_counter() - t
    print(f"{name}: operator create {dt:.3f} s, plan create {dp:.3f} s", flush=True)
    plan.close(); op.close()
