"""Time of the fused update + alpha pass under the timing switches (SLQ_FA_MODE). usage: python scripts/fa_time.py [workload] [orth]"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d, laplacian_3d
from primate_amd import engine as eng
w = sys.argv[1] if len(sys.argv) > 1 else "lap2d_1000"
orth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind, m = w.split("_")
A = laplacian_2d(int(m)) if kind == "lap2d" else laplacian_3d(int(m))
op = eng.DeviceOperator(A)
for label, env in (("separate", {"SLQ_FUSED_ALPHA": "0"}), ("fused", {}), ("fused, no poll", {"SLQ_FA_MODE": "1"}), ("fused, no A items", {"SLQ_FA_MODE": "2"}), ("fused, neither", {"SLQ_FA_MODE": "3"})):
	for k, v in env.items():
		os.environ[k] = v
	plan = eng.LanczosPlan(op, 256, 30, orth)
	plan.generate_probes("rademacher", seed=5); plan.run(); plan.quadrature("log")
	plan.profile_enable(True); plan.profile_read(reset=True)
	for it in range(2):
		plan.generate_probes("rademacher", seed=6 + it); plan.run(); q = plan.quadrature("log")
	pr = plan.profile_read(reset=True)
	plan.close()
	for k in env:
		del os.environ[k]
	print(f"{w} orth {orth} {label:20s}: update {pr['reorth_update']['ms']/pr['reorth_update']['launches']:.3f} ms/launch, alpha class {pr['spmm_3term']['ms']/2:.2f} ms/run, mean {q.mean():.6e}", flush=True)
