"""Full-size runs of BASELINE.json configs[2]-[4] on ONE GPU: wall time, probe-matvecs/s, and for the Lanczos loop of
each the per-kernel roofline (HIP events on the library's stream, the same algorithmic-byte model as bench.py).

    python scripts/run_configs.py [c3 c3x c4 c5 ...]   ->  gpurun_out/<RUN_TAG, default r04>_configs.json  (copied to profiles/ when judged)

c3  = configs[2] operator, hutch (quadrature) at orth 0 and 3          c3x = configs[2] as worded: xtrace, 512 vectors
c4  = configs[3]: diag(exp(-t L)), 126^3 7-point grid, fp32, k = 50, 1024 probes
c5  = configs[4]: n = 1e7, 15 nnz/row, k = 80, full reorthogonalisation: one GPU's 256-probe share (2048 probes / 8 GPUs)
"""

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
from conftest import laplacian_3d  # noqa: E402
from primate_amd.engine import DeviceOperator, DiagAccumulator, LanczosPlan  # noqa: E402

which = sys.argv[1:] or ["c3", "c3x", "c4", "c5"]
out_path = ROOT / "gpurun_out" / f"{os.environ.get('RUN_TAG', 'r04')}_configs.json"
out = json.loads(out_path.read_text()) if out_path.exists() else {}
if out.get("_meta", {}).get("kernel_sha256") not in (None, bench.kernel_sources_sha256()):
	out = {}  # (entries measured on other kernel sources are not carried along)
out["_meta"] = {"kernel_sha256": bench.kernel_sources_sha256(), "peak_GBps": bench.HBM_PEAK_GBS,
                "note": "per-kernel ms are HIP events on the library's stream over the timed batches; alg_GBps = bench.kernel_bytes / ms"}  # fmt: skip


def lanczos_roofline(plan, A, b, deg, orth, batches):
	"""Per-kernel-class time and algorithmic GB/s of the Lanczos loop, from the plan's HIP-event profile."""
	info = plan.describe()
	n, nnz, s = A.shape[0], A.nnz, A.dtype.itemsize
	## (every config here draws its probes on the device; Rademacher ones need no norm sweep since r04)
	kb, kl = bench.kernel_bytes(n, nnz, s, b, info["panel_width"], deg, orth, sequence=info["sequence"], upper_alpha=bool(info["upper_alpha"]), norm_sweep=False,
	                            last_nostore=not plan.keep_basis, sweep_live_frac=(lambda rd, off: rd / off if off else 1.0)(*plan.sweep_columns(reset=True)))
	prof = plan.profile_read(reset=True)
	rows = {}
	for k, v in prof.items():
		if v["launches"] == 0:
			continue
		rows[k] = {"ms_per_batch": round(v["ms"] / batches, 3), "launches_per_batch": v["launches"] / batches}
		if k in kb and kl[k] and v["ms"] > 0:
			gbps = kb[k] * batches / (v["ms"] * 1e-3) / 1e9
			rows[k].update(alg_GB_per_launch=round(kb[k] / kl[k] / 1e9, 3), alg_GBps=round(gbps, 1), frac_of_peak=round(gbps / bench.HBM_PEAK_GBS, 4))
	loop_ms = sum(v["ms"] for k, v in prof.items() if k in kb) / batches
	dom = max((k for k in rows if k in kb), key=lambda k: rows[k]["ms_per_batch"])
	return {"plan": info, "resident_probes_b": b, "workspace_GB": round(plan.workspace_bytes / 1e9, 2), "kernels": rows, "dominant": dom,
	        "loop_ms_per_batch": round(loop_ms, 2), "loop_alg_GBps": round(sum(kb.values()) / (loop_ms * 1e-3) / 1e9, 1)}  # fmt: skip


def random_graph():
	n = 500000
	rng = np.random.default_rng(1234)
	mm = int(n * 16 / 2)
	i, j = rng.integers(0, n, mm), rng.integers(0, n, mm)
	keep = i != j
	W = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(n, n)).tocsr()
	W = ((W + W.T) > 0).astype(np.float64).tocsr()
	W.sort_indices()
	return W


if "c3" in which:
	W, k, P, B = random_graph(), 40, 512, 256
	op = DeviceOperator(W)
	res = {"n": W.shape[0], "nnz": int(W.nnz), "k": k, "probes": P, "dtype": "f64"}
	for orth in (0, 3):
		plan = LanczosPlan(op, B, k, orth)
		plan.generate_probes("rademacher", seed=1234)
		plan.run()
		plan.quadrature("exp")  # warm-up
		plan.profile_enable(True)
		plan.profile_read(reset=True)
		qs = []
		op.ctx.synchronize()
		t0 = time.time()
		for c in range(0, P, B):
			plan.generate_probes("rademacher", seed=1234, probe_offset=c)
			plan.run()
			qs.append(plan.quadrature("exp"))
		dt = time.time() - t0
		q = np.concatenate(qs)
		roof = lanczos_roofline(plan, W, B, k, orth, P // B)
		## gather-aware bound: on G(n, p) every stored nonzero is an L2 miss - its panel row (panel width x 8 B) comes from
		## HBM - so a gathering launch moves nnz x row bytes per panel next to its algorithmic bytes; MI355X_MICROARCH.md gives
		## 7.4-7.9 TB/s for random whole rows of a table this size (Indexed rows: gather into LDS), 7.65 used here
		pw = roof["plan"]["panel_width"]
		gather = float(W.nnz) * pw * 8 * (B // pw)
		for kname, row in roof["kernels"].items():
			if kname in ("spmm_3term", "reorth_dot") and "alg_GB_per_launch" in row and (kname == "spmm_3term" or roof["plan"]["sequence"] == "fused_stored_u"):
				ms = row["ms_per_batch"] / row["launches_per_batch"]
				row.update(gather_bytes_per_launch=int(gather), gather_GBps=round(gather / (ms * 1e-3) / 1e9, 1), frac_of_gather_bound=round(gather / (ms * 1e-3) / 1e9 / 7650.0, 4))
		res[f"hutch_orth{orth}"] = dict(seconds=round(dt, 4), probe_matvecs_per_s=round(P * k / dt, 1), estimate=float(q.mean()),
		                                stderr=float(q.std(ddof=1) / np.sqrt(P)), **roof)  # fmt: skip
		plan.close()
	out["configs[2]_hutch"] = res
	print(json.dumps({"c3": res}), flush=True)
	op.close()

if "c3x" in which:
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import hutch, xtrace

	W, k, P = random_graph(), 40, 512
	M = MatrixFunction(W, fun="exp", deg=k, orth=3)
	xtrace(M, batch=128, seed=1, count=128, device_rng=True)  # warm-up (plans, kernels)
	t0 = time.time()
	marks = []
	est, info = xtrace(M, batch=128, seed=1234, count=P, full=True, device_rng=True, callback=lambda r: marks.append((r.nit, float(r.estimate), round(time.time() - t0, 4))))
	dt = time.time() - t0
	t0 = time.time()
	hd = hutch(M, pdf="device:rademacher", converge="count", count=P, seed=1234)
	dth = time.time() - t0
	out["configs[2]_xtrace"] = dict(n=W.shape[0], nnz=int(W.nnz), k=k, vectors=P, batch=128, orth=3, xtrace_seconds=round(dt, 4), xtrace_estimate=float(est),
	                                 f_A_products=2 * P, probe_matvecs_per_s=round(2 * P * k / dt, 1), progress=marks,
	                                 hutch_device_rng_seconds=round(dth, 4), hutch_device_rng_estimate=float(hd))  # fmt: skip
	print(json.dumps({"c3x": out["configs[2]_xtrace"]}), flush=True)

if "c4" in which:
	m, t, k, P, B = 126, 0.1, 50, 1024, 256
	A = laplacian_3d(m, dtype=np.float32)
	op = DeviceOperator(A)
	n = A.shape[0]
	T = (sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))).toarray()
	w, U = np.linalg.eigh(T)
	d1 = ((U * np.exp(-t * w)) @ U.T).diagonal()
	exact = np.einsum("i,j,k->ijk", d1, d1, d1).ravel()
	acc = DiagAccumulator(n, ctx=op.ctx)
	plan = LanczosPlan(op, B, k, 3, keep_basis=True)
	plan.generate_probes("rademacher", seed=99)
	plan.run()  # warm-up
	plan.profile_enable(True)
	plan.profile_read(reset=True)
	op.ctx.synchronize()
	t0 = time.time()
	for c in range(0, P, B):
		plan.generate_probes("rademacher", seed=1234, probe_offset=c)
		plan.run()
		acc.update(plan, "exp", t=-t)
	numer, denom, rmean, cnt = acc.get()
	dt = time.time() - t0
	est = numer / denom
	out["configs[3]"] = dict(n=n, nnz=int(A.nnz), k=k, probes=P, dtype="f32", orth=3, seconds=round(dt, 4), probe_matvecs_per_s=round(P * k / dt, 1),
	                         rel_l2_err=float(np.linalg.norm(est - exact) / np.linalg.norm(exact)), expected_stat_err=float(1 / np.sqrt(P)),
	                         **lanczos_roofline(plan, A, B, k, 3, P // B))  # fmt: skip
	print(json.dumps({"c4": out["configs[3]"]}), flush=True)
	plan.close()
	acc.close()
	op.close()

if "c5" in which:
	## one GPU's share of configs[4] as worded - eigencount by the step function, full reorthogonalisation: 2048 probes / 8 GPUs
	## = 256 probes, in batches of as many as the ring admits. The operator is the symmetric circulant band of
	## tests/test_gpu_fullsize.py (15 per row, closed-form spectrum), the cut its median: the count is known exactly.
	## C5_RING32=1 runs the opt-in fp32 archive of finished vectors: twice the probes per batch.
	from test_gpu_fullsize import circulant_band

	n, k, share = 10_000_000, 80, int(os.environ.get("C5_PROBES", 256))
	A, lam = circulant_band(n)
	cut = float(np.median(lam)) + 1e-3
	exact = int(np.count_nonzero(lam >= cut))
	del lam
	op = DeviceOperator(A)
	for ring32 in ([0, 1] if os.environ.get("C5_RING32", "both") == "both" else [int(os.environ["C5_RING32"])]):
		B = int(os.environ.get("C5_BATCH", 64 if ring32 else 32))  # 81 ring slots x n x 32 x 8 B = 207 GB of the 288 GB
		if ring32:
			os.environ["SLQ_RING32"] = "1"
		plan = LanczosPlan(op, B, k, k)
		plan.profile_enable(True)
		plan.profile_read(reset=True)
		qs = []
		op.ctx.synchronize()
		t0 = time.time()
		for c in range(0, share, B):
			plan.generate_probes("rademacher", seed=1234, probe_offset=c)
			plan.run()
			qs.append(plan.quadrature("step", c=cut))
			print(f"c5 ring32={ring32} batch at probe {c}: {time.time() - t0:.1f} s", flush=True)
		dt = time.time() - t0
		q = np.concatenate(qs)
		key = "configs[4]_one_gpu_share" + ("_ring32" if ring32 else "")
		out[key] = dict(n=n, nnz=int(A.nnz), k=k, probes=share, batch=B, orth=k, fun=f"step(c={cut:.6f})", seconds=round(dt, 3), probe_matvecs_per_s=round(share * k / dt, 1),
		                eigencount=float(q.mean()), exact_count=exact, rel_err=float(q.mean() / exact - 1), stderr=float(q.std(ddof=1) / np.sqrt(share)),
		                **lanczos_roofline(plan, A, B, k, k, share // B))  # fmt: skip
		print(json.dumps({"c5": out[key]}), flush=True)
		plan.close()
		os.environ.pop("SLQ_RING32", None)
	op.close()

out_path.parent.mkdir(exist_ok=True)
out_path.write_text(json.dumps(out, indent=1))
