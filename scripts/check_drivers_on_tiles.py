"""The drivers over an operator that takes the LDS tiles (90,000-row grid): xtrace with resident sketches, hutch with device
probes and diag give the same answers as on the generic passes (SLQ_TILES=0) to rounding - the row order is the library's own
business.    python scripts/check_drivers_on_tiles.py"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d  # noqa: E402
from primate_amd.diagonal import diag  # noqa: E402
from primate_amd.operators import MatrixFunction  # noqa: E402
from primate_amd.trace import hutch, xtrace  # noqa: E402

A = laplacian_2d(300)
res = {}
for mode in ("default", "0"):
	if mode == "0":
		os.environ["SLQ_TILES"] = "0"
	M = MatrixFunction(A, fun="exp", deg=20, orth=3, t=-0.5)
	e, info = xtrace(M, batch=128, seed=3, count=256, full=True, device_rng=True)
	h = hutch(M, pdf="device:rademacher", converge="count", count=256, seed=3)
	d = diag(M, converge="count", count=128, seed=5)
	print(mode, float(e), float(h), float(np.sum(d)), {k: pl.describe()["tiles"] for k, pl in M._plans.items()}, flush=True)
	res[mode] = (float(e), float(h), np.asarray(d))
print("relative differences: xtrace %.2e, hutch %.2e, diag %.2e" % (abs(res["default"][0] / res["0"][0] - 1), abs(res["default"][1] / res["0"][1] - 1),
      np.max(np.abs(res["default"][2] / res["0"][2] - 1))))
assert abs(res["default"][0] / res["0"][0] - 1) < 1e-9 and abs(res["default"][1] / res["0"][1] - 1) < 1e-11 and np.max(np.abs(res["default"][2] / res["0"][2] - 1)) < 1e-9
print("ok")
