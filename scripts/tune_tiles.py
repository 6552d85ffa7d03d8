import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d
from primate_amd.engine import DeviceOperator, LanczosPlan
L2 = laplacian_2d(1000)
for tiles, tr, bt in [(1, 4, 1), (1, 2, 1)]:
    os.environ["SLQ_TILES"] = str(tiles); os.environ["SLQ_TILE_ROWS"] = str(tr); os.environ["SLQ_BLOCKS_PER_CU_TILED"] = str(bt)
    t0 = time.time(); op = DeviceOperator(L2); tup = time.time() - t0
    for orth in [0, 3]:
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
        print(f"tiles={tiles} TR={tr} blocks/CU={bt} upload={tup:.2f}s orth={orth} step={min(ts[1:])*1e3:.1f} ms  pmv/s={256*30/min(ts[1:]):.0f}  {ks} est={np.mean(q):.6f}", flush=True)
        plan.close()
    op.close()
