import os, sys, time, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d
from primate_amd.engine import DeviceOperator, LanczosPlan
L2 = laplacian_2d(1000)
op = DeviceOperator(L2)
for orth in [0]:
    for pol in [0, 1, 2, 10, 11, 12]:
        for a in [4, 8]:
            os.environ["SLQ_SPMM_POLICY"] = str(pol); os.environ["SLQ_BLOCKS_PER_CU_SPMM"] = str(a)
            plan = LanczosPlan(op, 256, 30, orth)
            for it in range(3):
                plan.generate_probes("rademacher", seed=1234)
                if it == 1: plan.profile_enable(True); plan.profile_read()
                op.ctx.synchronize(); t0 = time.time()
                plan.run(); q = plan.quadrature("log")
                dt = time.time() - t0
            prof = plan.profile_read()
            ks = {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in prof.items() if v["launches"]}
            print(f"orth={orth} pol={pol} a={a} step={dt*1e3:.1f} ms  pmv/s={256*30/dt:.0f}  avg_ms/launch={ks} est={np.mean(q):.6f}", flush=True)
            plan.close()
