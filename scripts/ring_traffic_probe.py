"""HBM traffic of the tiled passes on operators with a known amount of sharing between tiles (run under
scripts/pmc_counters.sh with PMC_SCRIPT): `paths` = disjoint 14-node paths (a tile's rows read nothing outside the tile:
traffic must equal the algorithmic bytes), `lap2d` / `lap3d` = the bench grids.

    python scripts/ring_traffic_probe.py paths|lap2d|lap3d [orth]
"""
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d, laplacian_3d  # noqa: E402
from primate_amd.engine import DeviceOperator, LanczosPlan  # noqa: E402

kind = sys.argv[1]
orth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if kind == "paths":
	T = sp.diags([-1.0, 2.5, -1.0], [-1, 0, 1], shape=(14, 14))
	A = sp.kron(sp.identity(71429), T).tocsr()  # n = 1,000,006
elif kind == "lap2d":
	A = laplacian_2d(1000)
else:
	A = laplacian_3d(100)
op = DeviceOperator(A)
plan = LanczosPlan(op, 256, 10, orth)
print(plan.describe(), flush=True)
for rep in range(3):
	plan.generate_probes("rademacher", seed=rep)
	plan.run()
op.ctx.synchronize()
