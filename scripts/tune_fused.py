import os, sys, time, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d
from primate_amd.engine import DeviceOperator, LanczosPlan
L2 = laplacian_2d(1000)
op = DeviceOperator(L2)
orths = [int(x) for x in os.environ.get("TUNE_ORTHS", "0,3,30").split(",")]
for orth in orths:
    for fused in [0, 1]:
        for a in ([8] if not fused else [4, 8]):
            for nt in [1] if not fused else [0, 1]:
                os.environ["SLQ_FUSED"] = str(fused); os.environ["SLQ_BLOCKS_PER_CU_SPMM"] = str(a); os.environ["SLQ_NT"] = str(nt)
                plan = LanczosPlan(op, 256, 30, orth)
                for it in range(3):
                    plan.generate_probes("rademacher", seed=1234)
                    if it == 1: plan.profile_enable(True); plan.profile_read()
                    op.ctx.synchronize(); t0 = time.time()
                    plan.run(); q = plan.quadrature("log")
                    dt = time.time() - t0
                prof = plan.profile_read()
                ks = {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in prof.items() if v["launches"]}
                print(f"orth={orth} fused={fused} a={a} nt={nt} step={dt*1e3:.1f} ms  pmv/s={256*30/dt:.0f}  avg_ms/launch={ks} est={np.mean(q):.9f}", flush=True)
                plan.close()
