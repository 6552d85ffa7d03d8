"""Gram sequence on the generic passes (k_csr_pass<PASS_UPDATEG>, r04): narrow panels (P = 16, 8, 1) and small operators, SLQ_GRAM_CSR=0 against the
default; per-probe parity with the oracle on both. usage: python scripts/ab_gram_csr.py"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d, laplacian_3d
from oracle import oracle
from primate_amd import engine as eng
oracle.build()
for name, A, deg in (("lap2d_1000", laplacian_2d(1000), 30), ("lap2d_200", laplacian_2d(200), 30), ("lap3d_40", laplacian_3d(40), 30)):
	n = A.shape[0]
	op = eng.DeviceOperator(A)
	for P in (16, 8, 1, 64 if n < 65536 else 16):
		for orth in (3, 6):
			row = {}
			rng = np.random.default_rng(3)
			X = np.asfortranarray(np.floor(rng.random((n, P)) * 2) * 2 - 1)
			ref = oracle.quad_batch(A, np.asfortranarray(X[:, [0, P - 1]]), deg, orth, fun="log", fresh_q=True)
			for g in ("0", "1"):
				os.environ["SLQ_GRAM_CSR"] = g
				plan = eng.LanczosPlan(op, P, deg, orth)
				info = plan.describe()
				plan.set_probes(X); plan.run(); q = plan.quadrature("log")
				reps = 5
				t0 = time.perf_counter()
				for it in range(reps):
					plan.generate_probes("rademacher", seed=it); plan.run()
				plan.quadrature("log")
				dt = (time.perf_counter() - t0) / reps
				plan.close()
				row[g] = (dt, np.max(np.abs(q[[0, P - 1]] / ref - 1)), info["sequence"], info["tiles"])
			del os.environ["SLQ_GRAM_CSR"]
			print(f"{name} P={P} orth={orth}: {row['0'][2]}/tiles {row['0'][3]} {row['0'][0]*1e3:.3f} ms (err {row['0'][1]:.1e}) -> {row['1'][2]}/tiles {row['1'][3]} {row['1'][0]*1e3:.3f} ms (err {row['1'][1]:.1e})  x{row['0'][0]/row['1'][0]:.2f}", flush=True)
	op.close()
