"""Parameter sweep on C2 (n=1e6 Laplacian, 256 probes, k=30): per-kernel ms for a few env settings."""
import os, sys, time, json, itertools
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d
from primate_amd.engine import DeviceOperator, LanczosPlan

L2 = laplacian_2d(1000)
op = DeviceOperator(L2)
orths = [int(x) for x in os.environ.get("TUNE_ORTHS", "0,3").split(",")]
grid = [dict(a=a, s=s) for a in [2, 4] for s in [2, 4]] if "TUNE_GRID" not in os.environ else json.loads(os.environ["TUNE_GRID"])
for orth in orths:
    for cfg in grid:
        os.environ["SLQ_BLOCKS_PER_CU_SPMM"] = str(cfg["a"]); os.environ["SLQ_BLOCKS_PER_CU_STREAM"] = str(cfg["s"])
        plan = LanczosPlan(op, 256, 30, orth)
        for it in range(3):
            plan.generate_probes("rademacher", seed=1234)
            if it == 1: plan.profile_enable(True); plan.profile_read()
            op.ctx.synchronize(); t0 = time.time()
            plan.run(); q = plan.quadrature("log")
            dt = time.time() - t0
        prof = plan.profile_read()
        ks = {k: round(v["ms"] / 2 / max(v["launches"] / 2, 1), 3) for k, v in prof.items() if v["launches"]}
        print(f"orth={orth} cfg={cfg} step={dt*1e3:.1f} ms  pmv/s={256*30/dt:.0f}  avg_ms/launch={ks} est={np.mean(q):.3f}", flush=True)
        plan.close()
