import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d
from primate_amd.engine import DeviceOperator, LanczosPlan
L2 = laplacian_2d(1000)
op = DeviceOperator(L2)
for orth in [0, 3]:
    for a, pad in [(8, 0), (4, 0), (4, 65536), (4, 45000), (6, 45000), (2, 0), (2, 100000), (3, 45000)]:
        os.environ["SLQ_BLOCKS_PER_CU_SPMM"] = str(a); os.environ["SLQ_ALPHA_LDS_PAD"] = str(pad)
        plan = LanczosPlan(op, 256, 30, orth)
        ts = []
        for it in range(4):
            plan.generate_probes("rademacher", seed=1234)
            if it == 1: plan.profile_enable(True); plan.profile_read()
            op.ctx.synchronize(); t0 = time.time()
            plan.run(); q = plan.quadrature("log")
            ts.append(time.time() - t0)
        prof = plan.profile_read()
        ks = {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in prof.items() if v["launches"] and k in ("spmm_3term", "axpy_norm", "reorth_dot", "reorth_update")}
        print(f"orth={orth} a={a} pad={pad} step={min(ts[1:])*1e3:.1f} ms  pmv/s={256*30/min(ts[1:]):.0f}  {ks}", flush=True)
        plan.close()
