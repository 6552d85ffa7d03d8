"""Fused update + alpha pass (slq_ring_fa.hpp) against the separate alpha pass and the oracle: values and times.
usage: python scripts/fa_check.py [small|c2|3d|all]"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d, laplacian_3d
from oracle import oracle
from primate_amd import engine as eng

oracle.build()
what = sys.argv[1] if len(sys.argv) > 1 else "small"
cases = []
if what in ("small", "all"):
	os.environ["SLQ_TILES"] = "2"
	cases += [("lap2d_200", laplacian_2d(200), 130, 14), ("lap3d_40", laplacian_3d(40), 130, 14), ("lap2d_300", laplacian_2d(300), 256, 30)]
if what in ("c2", "all"):
	cases += [("lap2d_1000", laplacian_2d(1000), 256, 30)]
if what in ("3d", "all"):
	cases += [("lap3d_100", laplacian_3d(100), 256, 30)]
for name, A, P, deg in cases:
	n = A.shape[0]
	op = eng.DeviceOperator(A)
	for orth in (3, 1, 2):
		res = {}
		for fa in ("1", "0"):
			os.environ["SLQ_FUSED_ALPHA"] = fa
			plan = eng.LanczosPlan(op, P, deg, orth)
			info = plan.describe()
			plan.generate_probes("rademacher", seed=5)
			V = plan.get_probes()[:, [0, P // 2, P - 1]] if fa == "1" else None
			plan.run()
			q = plan.quadrature("log")
			a, b, st = plan.tridiag()
			## timing: 3 runs with per-kernel events
			plan.profile_enable(True)
			plan.profile_read(reset=True)
			t0 = time.perf_counter()
			for it in range(3):
				plan.generate_probes("rademacher", seed=6 + it)
				plan.run()
				plan.quadrature("log")
			dt = (time.perf_counter() - t0) / 3
			prof = plan.profile_read(reset=True)
			plan.close()
			res[fa] = (q, a, b, info, dt, prof, V)
		del os.environ["SLQ_FUSED_ALPHA"]
		q1, a1, b1, i1, t1, p1, V = res["1"]
		q0, a0, b0, i0, t0_, p0, _ = res["0"]
		ref = oracle.quad_batch(A, np.asfortranarray(V), deg, orth, fun="log", fresh_q=True)
		cols = [0, P // 2, P - 1]
		print(f"{name} P={P} k={deg} orth={orth}: fused_alpha {i1['fused_alpha']}/{i0['fused_alpha']} seq {i1['sequence']} | fa vs sep quad {np.max(np.abs(q1/q0-1)):.1e} alpha {np.max(np.abs(a1-a0)):.1e} beta {np.max(np.abs(b1-b0)):.1e} | "
		      f"fa vs oracle {np.max(np.abs(q1[cols]/ref-1)):.1e} sep vs oracle {np.max(np.abs(q0[cols]/ref-1)):.1e} | ms/run fa {t1*1e3:.2f} sep {t0_*1e3:.2f} | "
		      f"update ms fa {p1['reorth_update']['ms']/3:.2f} sep {p0['reorth_update']['ms']/3:.2f}, alpha class ms fa {p1['spmm_3term']['ms']/3:.2f} sep {p0['spmm_3term']['ms']/3:.2f}", flush=True)
	op.close()
