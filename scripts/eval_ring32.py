"""Evaluation of the opt-in fp32 archive ring (SLQ_RING32=1) on the GPU box: per-probe quadrature error
against the CPU oracle with and without it, on identical probes, over operators / degrees / depths / functions; and the
step time with and without it on configs[1] at orth = 30. Prints one JSON object (kept under profiles/)."""

import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
from conftest import laplacian_2d, laplacian_3d  # noqa: E402
from oracle import oracle  # noqa: E402
from primate_amd import engine as eng  # noqa: E402

oracle.build()
rng = np.random.default_rng(5)


def random_spd(n, deg):
	m = int(n * deg / 2)
	i, j = rng.integers(0, n, m), rng.integers(0, n, m)
	W = sp.coo_matrix((rng.uniform(0.1, 1.0, m), (i, j)), shape=(n, n)).tocsr()
	W = W + W.T
	A = (sp.diags(np.asarray(abs(W).sum(axis=1)).ravel() + rng.uniform(0.05, 1.0, n)) - W).tocsr()
	A.sort_indices()
	return A


def quad(op, X, deg, orth, fun, kw, ring32):
	os.environ["SLQ_RING32"] = "1" if ring32 else "0"
	plan = eng.LanczosPlan(op, X.shape[1], deg, orth)
	assert (plan.describe()["sequence"] == "sweeps_ring32") == bool(ring32 and orth > 8)
	plan.set_probes(X)
	plan.run(1e-8)
	q = plan.quadrature(fun, **kw)
	plan.close()
	return q


cases = []
ops = [("lap2d_120", laplacian_2d(120)), ("lap3d_24", laplacian_3d(24)), ("graph_20000_6", random_spd(20000, 6.0)), ("graph_3000_12", random_spd(3000, 12.0))]
worst = {"fp64_vs_oracle": 0.0, "ring32_vs_oracle": 0.0, "ring32_vs_fp64": 0.0, "ring32_mean_vs_oracle_mean": 0.0}
for name, A in ops:
	op = eng.DeviceOperator(A)
	n = A.shape[0]
	X = np.asfortranarray(np.floor(rng.random((n, 24)) * 2) * 2 - 1)
	for deg in (30, 60):
		for orth in (9, 12, deg):
			for fun, kw in (("log", {}), ("exp", {"t": -0.1}), ("inv", {}), ("numrank", {})):
				ref = oracle.quad_batch(A, X, deg, orth, fun=fun, fresh_q=True, **kw)
				q64, q32 = quad(op, X, deg, orth, fun, kw, False), quad(op, X, deg, orth, fun, kw, True)
				rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))  # noqa: E731
				c = {"op": name, "deg": deg, "orth": orth, "fun": fun, "fp64_vs_oracle": rel(q64, ref), "ring32_vs_oracle": rel(q32, ref),
				     "ring32_vs_fp64": rel(q32, q64), "ring32_mean_vs_oracle_mean": abs(q32.mean() / ref.mean() - 1)}  # fmt: skip
				cases.append(c)
				for k in worst:
					worst[k] = max(worst[k], c[k])
	op.close()

## speed and bytes: configs[1] operator, 256 probes, k = 30, full reorthogonalisation
A = laplacian_2d(1000)
op = eng.DeviceOperator(A)
timing = {}
for ring32 in (0, 1):
	os.environ["SLQ_RING32"] = str(ring32)
	plan = eng.LanczosPlan(op, 256, 30, 30)
	info = plan.describe()
	for it in range(3):
		plan.generate_probes("rademacher", seed=1234, probe_offset=0)
		if it == 1:
			plan.profile_enable(True)
			plan.profile_read(reset=True)
			op.ctx.synchronize()
			t0 = time.time()
		plan.run(1e-8)
		q = plan.quadrature("log")
	dt = (time.time() - t0) / 2
	prof = plan.profile_read(reset=True)
	kb, kl = bench.kernel_bytes(A.shape[0], A.nnz, 8, 256, info["panel_width"], 30, 30, sequence=info["sequence"], upper_alpha=bool(info["upper_alpha"]))
	timing["ring32" if ring32 else "fp64"] = {
		"ms_per_step": round(dt * 1e3, 2), "estimate": float(q.mean()), "workspace_GB": round(plan.workspace_bytes / 1e9, 2), "sequence": info["sequence"],
		"alg_GB_per_step": round(sum(kb.values()) / 1e9, 1),
		"kernels": {k: {"ms": round(v["ms"] / 2, 2), "alg_GBps": round(kb[k] * 2 / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in prof.items() if k in kb and v["launches"]},
	}  # fmt: skip
	plan.close()
print(json.dumps({"worst": worst, "timing_configs1_orth30": timing, "cases": cases}))
