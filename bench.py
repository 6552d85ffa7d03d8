"""bench.py — BASELINE.json's metric on BASELINE.json's config, on N GPUs of one node.

Metric: probe-matvecs/s (1 probe-matvec = one Lanczos step for one probe: SpMV + three-term
update + reorthogonalisation sweep + norm) on configs[1]: logdet by SLQ of the
2D 5-point Laplacian, n = 1,000,000, nnz = 4,996,000, k = 30, 256 Rademacher probes per GPU,
fp64. A "step" is one full pass of the hot path over one probe batch: draw 256 probes on the
device, 30 lock-step Lanczos steps, on-device Gauss quadrature, sum f(theta)*tau*||v||^2, and
(N > 1) one RCCL all-reduce of the three sufficient statistics of the trace estimator. The CSR
operator is resident in HBM before the timed region. Probes shard across GPUs with no data-path
collective (weak scaling: 256 probes per GPU; probe ids are global, so results do not depend on N).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--orth R] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no WORLD_SIZE in the environment (a bare `python bench.py --gpus N`) the process starts its own N
ranks (spawn_ranks: child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before anything here touches
torch or HIP), relays rank 0's line and exits with the worst child's code.

--scaling strong: --probes is the GLOBAL batch of a step and rank r advances its contiguous shard (256 probes over 8 GPUs =
32 per GPU, the narrow-panel plans); the line then says "scaling": "strong". Default: weak (256 per GPU).

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def laplacian_2d(m: int, dtype=np.float64):
	import scipy.sparse as sp

	T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))
	A = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsr().astype(dtype)
	A.sort_indices()
	return A


def laplacian_3d(m: int, dtype=np.float64):
	import scipy.sparse as sp

	T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))
	I = sp.identity(m)
	A = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr().astype(dtype)
	A.sort_indices()
	return A


## ---- algorithmic-byte model -------------------------------------
def contract_bytes_per_probe_matvec(n, nnz, s, b, j, orth):
	"""B(j) = [(s+4) nnz + 4(n+1)]/b + (8 + 2 r_j) s n, r_j = min(j+1, orth)."""
	r = 0 if orth == 0 else min(j + 1, orth)
	return ((s + 4) * nnz + 4 * (n + 1)) / b + (8 + 2 * r) * s * n


FUSED_MAX_R = 8  # slq_kernels.hpp:kFusedMaxR


def kernel_bytes(n, nnz, s, b, pw, deg, orth, chunk=16, fused=True, sequence=None, upper_alpha=True, norm_sweep=True, last_nostore=True, sweep_live_frac=1.0):
	"""Algorithmic HBM bytes of every launch of one `run`, per kernel class: each
	vector panel a kernel touches is read or written once, the CSR arrays once per panel of `pw`
	probes. Launch sequences per Lanczos step (slq.hip:enqueue_run; `LanczosPlan.describe()["sequence"]`):
	  "fused" (CSR with gather locality, r_j <= 8): the SpMM is recomputed in each pass and only the last pass writes —
	    alpha pass [spmm_3term] at r = 0, merged alpha+dots pass [reorth_dot] at r >= 1, update pass
	    [reorth_update | axpy_norm];
	  "fused_stored_u" (CSR without gather locality, 1 <= r_j <= 8): the merged pass also stores u and the update pass
	    reads it back instead of gathering again; its r = 0 steps run the sweeps;
	  "sweeps" (deeper reorthogonalisation, other operators): spmm_3term writes w, then reorth_dot /
	    reorth_update (or axpy_norm) revisit it in place."""
	if sequence is None:
		sequence = "fused" if fused else "sweeps"
	npan = math.ceil(b / pw)
	vec = s * n * b
	csr = npan * ((s + 4) * nnz + 4 * (n + 1))
	out = {"spmm_3term": 0.0, "axpy_norm": 0.0, "reorth_dot": 0.0, "reorth_update": 0.0}
	launches = dict.fromkeys(out, 0)
	if norm_sweep:  # ||v||^2 of the probes (not for Rademacher probes drawn on the device: n, known - r04)
		out["axpy_norm"] += vec
		launches["axpy_norm"] += 1
	for j in range(deg):
		r = 0 if orth == 0 else min(j + 1, orth)
		rd = (1 if j == 0 else 2) + max(r - 2, 0)  # q_c (gather), q_p, ring columns beyond those two
		## the fused update pass of a run's LAST step stores nothing (r04: W_deg is never read, only its norm; plans without a kept basis)
		wr = 0 if (last_nostore and j == deg - 1) else 1
		if sequence == "fused_gram" and 1 <= r <= FUSED_MAX_R:
			## alpha-only pass (q_c alone: the -beta q_c.q_p part and every projection come from Gram rows the update passes take) and
			## the update pass; steps with r = 0 never occur in this sequence (orth >= 1), deeper ones fall through to the sweeps
			nz = (nnz + n) // 2 if upper_alpha else nnz
			out["spmm_3term"] += npan * ((s + 4) * nz + 4 * (n + 1)) + vec
			launches["spmm_3term"] += 1
			out["reorth_update"] += csr + (rd + wr) * vec
			launches["reorth_update"] += 1
			continue
		if sequence in ("fused", "fused_gram") and r <= FUSED_MAX_R:
			if r == 0:
				# alpha pass (orth = 0 only; with r >= 1 alpha comes out of the dots pass): q_c only (q_c.q_p comes
				# from the previous update pass's cross term), over the upper triangle of an exactly symmetric CSR
				nz = (nnz + n) // 2 if upper_alpha else nnz
				out["spmm_3term"] += npan * ((s + 4) * nz + 4 * (n + 1)) + vec
				launches["spmm_3term"] += 1
			if r > 0:
				out["reorth_dot"] += csr + rd * vec
				launches["reorth_dot"] += 1
			k = "reorth_update" if r > 0 else "axpy_norm"
			out[k] += csr + (rd + wr) * vec
			launches[k] += 1
			continue
		if sequence == "fused_stored_u" and 1 <= r <= FUSED_MAX_R:
			out["reorth_dot"] += csr + (rd + 1) * vec  # gathers, reads the ring rows, WRITES u
			launches["reorth_dot"] += 1
			out["reorth_update"] += (rd + 2) * vec  # reads u and the ring rows, writes w; no gather
			launches["reorth_update"] += 1
			continue
		out["spmm_3term"] += csr + (2 if j == 0 else 3) * vec  # gather q_c, read q_p, write w
		launches["spmm_3term"] += 1
		## "sweeps_ring32" (SLQ_RING32 opt-in): ring columns i >= 2 are read from the fp32 archive (half the bytes), and
		## the last update sweep also writes w there
		half = sequence == "sweeps_ring32"
		cols = lambda a, b: sum(0.5 if (half and i >= 2) else 1.0 for i in range(a, b))  # noqa: E731
		if r == 0:
			out["axpy_norm"] += 3 * vec  # read w, q_c; write w
			launches["axpy_norm"] += 1
		else:
			ch = 8 if half else chunk  # slq_kernels.hpp: kReorthChunk32 / kReorthChunk
			for i0 in range(0, r, ch):
				rc = min(ch, r - i0)
				## plain ring: every dots chunk is read-only since r04 (the axpy `w -= cB W_c` is applied in registers and stored by the
				## update sweep): w + the chunk's columns, and W_c once more in later chunks; the archive form stores in its first chunk
				out["reorth_dot"] += (cols(i0, i0 + rc) + ((2 if i0 == 0 else 1) if half else (1 if i0 == 0 else 2))) * vec
				launches["reorth_dot"] += 1
			## (the update sweep reads a ring column only when some probe of the panel projects on it: sweep_live_frac = columns read / columns
			## offered, measured - LanczosPlan.sweep_columns(); the fp32-archive form reads them all)
			out["reorth_update"] += (cols(0, r) * (1.0 if half else sweep_live_frac) + 2 + (0.5 if half else 0.0)) * vec
			launches["reorth_update"] += 1
	return out, launches


def pmc_traffic(workload, P, deg, orth, dom):
	"""HBM bytes per launch of kernel class `dom` from the committed PMC summary (scripts/collect_profiles.sh ->
	scripts/summarise_pmc.py), and where that number comes from. The counters are collected in their own rocprofv3
	passes, not in this run: the summary records the sha256 of the kernel sources it was measured on, and a summary
	taken on other sources yields traffic = None."""
	pmc = ROOT / "profiles" / "pmc_summary.json"
	if not pmc.exists():
		return None, None
	try:
		rec = json.loads(pmc.read_text())
	except Exception:  # noqa: BLE001
		return None, None
	meta = rec.get("_meta", {})
	src = {"file": "profiles/pmc_summary.json", "tag": meta.get("tag"), "kernel_sha256": meta.get("kernel_sha256"), "current": True}
	if meta.get("kernel_sha256") != kernel_sources_sha256():
		src["current"] = False
		return None, src
	return rec.get(f"{workload}/P{P}/k{deg}/orth{orth}", {}).get(dom, {}).get("hbm_bytes_per_launch"), src


def kernel_sources_sha256():
	import hashlib

	h = hashlib.sha256()
	for f in ("primate_amd/csrc/slq_common.hpp", "primate_amd/csrc/slq_kernels.hpp", "primate_amd/csrc/slq_ring.hpp", "primate_amd/csrc/slq_ring_fa.hpp", "primate_amd/csrc/slq_build.hpp", "primate_amd/csrc/slq.hip"):
		h.update((ROOT / f).read_bytes())
	return h.hexdigest()


def shard_range(nprobes, rank, world):
	"""Contiguous block [lo, hi) of global probe ids owned by `rank` (primate_amd.distributed.shard_range; first ranks take the remainder)."""
	base, rem = divmod(int(nprobes), int(world))
	lo = rank * base + min(rank, rem)
	return lo, lo + base + (1 if rank < rem else 0)


def measure(ctx, workload, dtype, P, deg_req, orth_req, steps, warmup, fun, rank, world, dist, red_dev, profiled=True, stream_rates=True, scaling="weak"):
	"""One bench measurement: `warmup` untimed + `steps` timed passes of the hot path on `workload`; returns the JSON line.
	scaling = "weak": every rank advances P probes per step; "strong": P is the GLOBAL batch of a step and rank r advances its
	contiguous shard of it (256 probes over 8 GPUs = 32 per GPU: the narrow-panel plans; src/primate/trace.py:36 draws 32 per batch)."""
	import torch

	from primate_amd.engine import DeviceOperator, LanczosPlan

	kind, m = workload.split("_")
	np_dt = np.float64 if dtype == "f64" else np.float32
	A = laplacian_2d(int(m), dtype=np_dt) if kind == "lap2d" else laplacian_3d(int(m), dtype=np_dt)
	n, nnz, s = A.shape[0], A.nnz, A.dtype.itemsize
	t_create = time.perf_counter()
	op = DeviceOperator(A, ctx=ctx)  # CSR resident in HBM before the timed region
	ctx.synchronize()
	create_s = time.perf_counter() - t_create  # host analysis (reordering, tiles, streams) + upload: what a one-shot call pays on top
	deg = min(deg_req, n)
	orth = deg if orth_req < 0 or orth_req > deg else orth_req
	P_global = P * world if scaling == "weak" else P  # probes of one step over all ranks
	if scaling == "strong":
		lo, hi = shard_range(P_global, rank, world)
		assert hi > lo, f"--scaling strong: {P_global} probes do not reach rank {rank} of {world}"
		P = hi - lo
	else:
		lo = rank * P
	t_plan = time.perf_counter()
	plan = LanczosPlan(op, P, deg, orth)
	plan_s = time.perf_counter() - t_plan

	def step(it: int):
		## probe ids are global: step `it` draws ids [it * P_global, (it + 1) * P_global), rank r its block of them
		plan.generate_probes("rademacher", seed=1234, probe_offset=it * P_global + lo)
		plan.run(1e-8)
		q = plan.quadrature(fun)  # device QL + reduction; returns P doubles (synchronises)
		if dist is not None:
			st = torch.tensor([q.sum(), (q * q).sum(), float(len(q))], dtype=torch.float64, device=red_dev)
			dist.all_reduce(st)  # RCCL over xGMI: the only collective on the path
			return st
		return q

	def barrier():
		if dist is not None:
			dist.barrier()
		torch.cuda.synchronize()
		ctx.synchronize()

	for it in range(warmup):
		step(it)
	plan.sweep_columns(reset=True)
	## per-kernel HIP events over the timed region (the roofline object needs them); BENCH_NO_PROFILE=1 times the
	## same steps without them (hipGraph replay) to show what the instrumentation costs
	plan.profile_enable(profiled)
	plan.profile_read(reset=True)
	barrier()
	t0 = time.perf_counter()
	ests = []
	for it in range(steps):
		out = step(warmup + it)
		ests.append(out)
	barrier()
	elapsed = time.perf_counter() - t0
	if dist is not None:
		tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
		dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
		elapsed = float(tmax.item())
	prof_steps = steps
	if not profiled:  # A/B mode: kernel events from ONE extra step outside the timed region
		plan.profile_enable(True)
		plan.profile_read(reset=True)
		step(warmup + steps)
		barrier()
		prof_steps = 1
	prof = plan.profile_read(reset=True)
	plan.profile_enable(False)

	if dist is not None:
		tot = torch.stack(ests).sum(0).cpu().numpy()
		estimate = tot[0] / tot[2]
	else:
		estimate = float(np.mean(np.concatenate(ests)))

	probe_matvecs = P_global * deg * steps
	value = probe_matvecs / elapsed
	ms_per_step = elapsed / steps * 1e3

	## ---- roofline of the dominant kernel (HIP events on the kernels' own stream) --------------
	info = plan.describe()  # panel geometry and launch sequence the library chose
	cols_read, cols_offered = plan.sweep_columns(reset=True)  # (deep windows: ring columns the update sweeps read / were offered)
	pw, fused = info["panel_width"], not info["sequence"].startswith("sweeps")
	kb, kl = kernel_bytes(n, nnz, s, P, pw, deg, orth, sequence=info["sequence"], upper_alpha=bool(info["upper_alpha"]),
	                      norm_sweep=os.environ.get("SLQ_KNOWN_NORM", "1") == "0",  # (the bench draws Rademacher probes on the device)
	                      last_nostore=os.environ.get("SLQ_LAST_STORE", "0") == "0",
	                      sweep_live_frac=(cols_read / cols_offered) if cols_offered else 1.0)
	cand = {k: prof[k]["ms"] for k in kb if prof[k]["launches"] > 0}
	dom = max(cand, key=cand.get)
	launches = prof[dom]["launches"]
	avg_ms = prof[dom]["ms"] / launches
	alg_bytes_per_launch = kb[dom] / kl[dom]
	achieved = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9
	traffic, traffic_source = pmc_traffic(workload, P, deg, orth, dom)
	## the rate this card actually sustains for the sweeps' access shape (SURVEY.md §8d asks for both)
	measured = {m: round(ctx.measure_stream(m, nbytes=1 << 31, reps=5), 1) for m in ("read", "triad")} if stream_rates else None
	roofline = {
		"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
		"frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
		"measured_stream_GBps": measured, "frac_of_measured_triad": round(achieved / measured["triad"], 4) if measured else None,
		"alg_bytes_per_launch": int(alg_bytes_per_launch), "avg_launch_ms": round(avg_ms, 4), "launches": int(launches),
	}  # fmt: skip
	## whole-loop view: SURVEY §8(d) contract bytes per probe-matvec / wall time of the step
	contract = sum(contract_bytes_per_probe_matvec(n, nnz, s, P, j, orth) for j in range(deg)) / deg
	kernels = {
		k: {
			"ms_per_step": round(prof[k]["ms"] / prof_steps, 3),
			"launches_per_step": prof[k]["launches"] / prof_steps,
			**({"alg_GBps": round(kb[k] * prof_steps / (prof[k]["ms"] * 1e-3) / 1e9, 1)} if k in kb and prof[k]["ms"] > 0 else {}),
		}
		for k in prof
		if prof[k]["launches"] > 0 and (k not in kb or kl.get(k, 1) > 0)
	}

	line = {
		"metric": "probe-matvecs/sec", "value": round(value, 1), "unit": "probe-matvecs/s", "n_gpus": world,
		"steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
		"scaling": scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
		"config": {
			"workload": f"{'configs[1]: logdet via SLQ' if workload == 'lap2d_1000' and dtype == 'f64' else 'SLQ trace'}, {'2D 5-point' if kind == 'lap2d' else '3D 7-point'} Laplacian CSR n={n} nnz={nnz}, k={deg}, {P} Rademacher probes per GPU (device Philox), f={fun}"
			+ (f" [strong scaling: {P_global} probes per step over {world} GPUs]" if scaling == "strong" else ""),
			"n": n, "nnz": int(nnz), "deg": deg, "orth": orth, "probes_per_gpu": P, "probes_per_step_global": P_global, "resident_probes_b": P,
			"panel_width": info["panel_width"], "create_s": round(create_s, 4), "plan_s": round(plan_s, 4),
			"parallelism": f"probe-sharded x{world}, operator replicated", "fused_passes": bool(fused), "plan": info,
			"kernel_events_in_timed_region": bool(profiled),
			"update_sweep_columns": {"read": cols_read, "offered": cols_offered},
		},
		"trace_estimates_per_s": round(P_global * steps / elapsed, 1),
		"estimate": float(estimate),
		"roofline": roofline,
		"loop": {
			"contract_bytes_per_probe_matvec": int(contract),
			"contract_GBps_per_gpu": round(value / world * contract / 1e9, 1),
			"contract_frac_of_peak": round(value / world * contract / 1e9 / HBM_PEAK_GBS, 4),
			## the implementation's OWN algorithmic bytes (kernel_bytes: every launch of the step) over the step's wall time
			"own_bytes_per_step": int(sum(kb.values())),
			"own_bytes_GBps": round(sum(kb.values()) / (elapsed / steps) / 1e9, 1),
			"own_bytes_frac_of_peak": round(sum(kb.values()) / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, 4),
		},
		"kernels": kernels,
	}  # fmt: skip

	del plan, op
	return {"line": line, "A": A, "n": n, "deg": deg, "orth": orth}


SPAWN_TIMEOUT_S = 3000.0  # overall limit of a self-launched N-rank run (--timeout)


def spawn_ranks(n, argv, cmd=None, timeout=None):
	"""Start `n` ranks of this script as CHILD processes (one per GPU: LOCAL_RANK = RANK), relay rank 0's stdout, return the
	worst exit code. The parent never imports torch and never initialises a GPU, and nothing is re-executed in place: a
	process that holds a HIP context must not exec (the box refuses it), so the launcher stays a plain parent. `cmd`
	replaces `[python, bench.py]` (the CPU test of this function runs a stub)."""
	import socket
	import subprocess

	with socket.socket() as sk:  # a free rendezvous port, unless the caller fixed one
		sk.bind(("127.0.0.1", 0))
		port = sk.getsockname()[1]
	base = dict(os.environ)
	base.setdefault("MASTER_ADDR", "127.0.0.1")
	base.setdefault("MASTER_PORT", str(port))
	base["WORLD_SIZE"] = str(n)
	base["LOCAL_WORLD_SIZE"] = str(n)
	cmd = list(cmd) if cmd is not None else [sys.executable, str(Path(__file__).resolve())]
	procs = []
	for r in range(n):
		env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
		## rank 0 owns the JSON line; whatever other ranks print goes to stderr so that stdout stays ONE line
		procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
	## Watch EVERY child, not only rank 0: a rank that dies before or inside a collective leaves the others blocked in it (rank 0
	## included), so the first non-zero exit ends the rest - the children started here, by handle - and the run reports failure
	## instead of holding the GPUs until a watchdog fires. Rank 0's stdout is drained by a thread meanwhile (a full pipe would
	## block it).
	import threading

	chunks = []
	rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
	rd.start()
	t_end = None if timeout is None else time.monotonic() + timeout
	timed_out = False
	while True:
		codes = [p.poll() for p in procs]
		if all(c is not None for c in codes) or any(c not in (None, 0) for c in codes):
			break
		if t_end is not None and time.monotonic() > t_end:
			timed_out = True
			break
		time.sleep(0.05)
	for p in procs:  # exactly the children started above, nothing by pattern
		if p.poll() is None:
			p.terminate()
	for p in procs:
		try:
			p.wait(timeout=20)
		except subprocess.TimeoutExpired:
			p.kill()
			p.wait()
	rd.join(timeout=20)
	out0 = b"".join(c for c in chunks if c)
	if timed_out:
		print(f"bench.py: ranks still running after {timeout:.0f} s were ended", file=sys.stderr)
	## stdout stays ONE line: rank 0's JSON record. Anything else a rank-0 library printed there (gloo announces its peers on
	## stdout) is passed on through stderr
	for ln in out0.decode(errors="replace").splitlines():
		print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr, flush=True)
	codes = [p.returncode for p in procs]
	bad = [c for c in codes if c != 0]
	if bad:
		print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
		return max(abs(c) for c in bad) or 1
	return 0


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=10)
	ap.add_argument("--warmup", type=int, default=2)
	ap.add_argument("--orth", type=int, default=3, help="reorthogonalisation depth (MatrixFunction default: 3)")
	ap.add_argument("--deg", type=int, default=30)
	ap.add_argument("--probes", type=int, default=256, help="probes per GPU")
	ap.add_argument("--workload", default="lap2d_1000", help="lap2d_<m> | lap3d_<m>")
	ap.add_argument("--fun", default="log")
	ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="arithmetic type of the operator and the Lanczos vectors")
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--no-extra", action="store_true", help="skip the extra operators appended to the default line")
	ap.add_argument("--cpu-seconds", type=float, default=15.0)
	ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
	                help="weak: --probes per GPU (default); strong: --probes is the GLOBAL count, rank r takes its contiguous shard")
	ap.add_argument("--timeout", type=float, default=SPAWN_TIMEOUT_S, help="overall limit (s) of a self-launched N-rank run")
	args = ap.parse_args()

	N = args.gpus
	if N > 1 and "WORLD_SIZE" not in os.environ:
		sys.exit(spawn_ranks(N, sys.argv[1:], timeout=args.timeout))  # a bare `python bench.py --gpus N`: be the launcher (torch is not imported yet)
	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	assert world == N or (N == 1 and world == 1), f"--gpus {N} but WORLD_SIZE={world}"

	import torch

	## Rehearsal knobs (one-GPU box): BENCH_BACKEND=gloo keeps the collective on the CPU and
	## BENCH_DEVICE=0 puts every rank on one card. The driver's runs use neither: nccl (= RCCL over
	## xGMI), one rank per GPU.
	backend = os.environ.get("BENCH_BACKEND", "nccl")
	if "BENCH_DEVICE" in os.environ:
		local_rank = int(os.environ["BENCH_DEVICE"])
	red_dev = "cuda" if backend == "nccl" else "cpu"
	dist = None
	torch.cuda.set_device(local_rank)
	if world > 1:
		import torch.distributed as dist

		os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
		if backend == "nccl":
			dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
		else:
			dist.init_process_group(backend, rank=rank, world_size=world)

	from primate_amd.engine import Context

	ctx = Context(device=local_rank)
	res = measure(ctx, args.workload, args.dtype, args.probes, args.deg, args.orth, args.steps, args.warmup, args.fun,
				  rank, world, dist, red_dev, profiled=not os.environ.get("BENCH_NO_PROFILE"), scaling=args.scaling)
	line, A, n, deg, orth = res["line"], res["A"], res["n"], res["deg"], res["orth"]
	line["steps"], line["warmup"] = args.steps, args.warmup

	## ---- what a reader needs to verify an N-rank run: backend, world size, one distinct device per rank
	if dist is not None:
		prop = torch.cuda.get_device_properties(local_rank)
		mine = {
			"rank": rank, "local_rank": local_rank, "name": prop.name,
			"pci": f"{getattr(prop, 'pci_domain_id', 0):04x}:{getattr(prop, 'pci_bus_id', -1):02x}:{getattr(prop, 'pci_device_id', 0):02x}",
			"uuid": str(getattr(prop, "uuid", "")),
		}  # fmt: skip
		devs = [None] * world
		dist.all_gather_object(devs, mine)
		line["rccl"] = {"backend": dist.get_backend(), "world": world, "devices": devs}
		if dist.get_backend() == "nccl" and "BENCH_DEVICE" not in os.environ:
			ids = {(d["pci"], d["uuid"]) for d in devs}
			assert len(ids) == world, f"ranks share a GPU: {devs}"

	## ---- the other operators of BASELINE.md §2 in the same record (N = 1, default invocation only): the north_star's
	## 3-D 100^3 operator (nnz = 6.94 M) at the default and at no reorthogonalisation, and configs[1] at orth = 0
	default_run = args.workload == "lap2d_1000" and args.dtype == "f64" and args.orth == 3 and args.probes == 256 and args.deg == 30
	if rank == 0 and world == 1 and default_run and not args.no_extra:
		extra = {}
		## ... and the shapes the reference's drivers submit next to the 256-probe batch: a 64-probe panel (an 8-GPU shard of 512 probes;
		## hutch's batches are 32, src/primate/trace.py:36) and a six-column reorthogonalisation window
		cases = {
			"lap3d_100_orth3": ("lap3d_100", 3, 256), "lap3d_100_orth0": ("lap3d_100", 0, 256), "lap2d_1000_orth0": ("lap2d_1000", 0, 256),
			"lap3d_100_orth3_p64": ("lap3d_100", 3, 64), "lap2d_1000_orth3_p64": ("lap2d_1000", 3, 64), "lap2d_1000_orth6": ("lap2d_1000", 6, 256),
		}  # fmt: skip
		## ... and the north_star's wording of the inner loop, "with full reorthogonalization" (orth = k: lanczos.h:133-136; operators.py:77,80
		## clamp out-of-range orth to deg): 2 timed steps each, ~0.4 s per step on a 64 GB ring
		cases.update({"lap2d_1000_orth30": ("lap2d_1000", 30, 256, 2), "lap3d_100_orth30": ("lap3d_100", 30, 256, 2)})
		for key, (w, o, pr, *st) in cases.items():
			nst = st[0] if st else 3
			r = measure(ctx, w, "f64", pr, 30, o, nst, 1, args.fun, 0, 1, None, red_dev, profiled=True, stream_rates=False)["line"]
			extra[key] = {
				"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": nst, "warmup": 1,
				"create_s": r["config"]["create_s"], "own_bytes_frac_of_peak": r["loop"]["own_bytes_frac_of_peak"], "sequence": r["config"]["plan"]["sequence"],
				"workload": r["config"]["workload"], "nnz": r["config"]["nnz"], "estimate": r["estimate"],
				"roofline": {k: r["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "alg_bytes_per_launch", "avg_launch_ms", "launches")},
				"kernels": r["kernels"],
			}  # fmt: skip
		line["extra"] = extra

	## ---- CPU baseline: the oracle (C restatement of the reference kernel), rank 0, N = 1 only ---
	if rank == 0 and world == 1 and not args.no_cpu_baseline:
		from oracle import oracle

		rng = np.random.default_rng(1234)
		mk = lambda c: np.asfortranarray(np.floor(rng.random((n, c)) * 2) * 2 - 1)  # noqa: E731
		oracle.quad_batch(A, mk(1), deg, orth, fun=args.fun, fresh_q=False, nthreads=1)  # builds the CSC copy, warms up
		t = time.perf_counter()
		oracle.quad_batch(A, mk(2), deg, orth, fun=args.fun, fresh_q=False, nthreads=1)
		one = (time.perf_counter() - t) / 2
		cnt = int(max(2, min(128, args.cpu_seconds / max(one, 1e-3))))
		X = mk(cnt)
		t = time.perf_counter()
		qc = oracle.quad_batch(A, X, deg, orth, fun=args.fun, fresh_q=False, nthreads=1)
		tc = time.perf_counter() - t
		line["cpu_baseline"] = {
			"value": round(cnt * deg / tc, 2), "unit": "probe-matvecs/s", "cores": 1, "kind": "port",
			"sample": f"{cnt} Rademacher probes x k={deg}, orth={orth}, same operator, 1 thread (the reference's execution model; "
			f"oracle/slq_oracle.c restating lanczos.h:43-149 + CSC SpMV + QL quadrature), {tc:.1f} s",
			"s_per_probe": round(tc / cnt, 3), "estimate": float(np.mean(qc)),
		}  # fmt: skip
		## generous upper bound the reference does not have: all host cores, OpenMP over probes
		ncore = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))  # the box's CPU share for one GPU
		if ncore > 1 and not os.environ.get("BENCH_SKIP_CPU_PARALLEL"):
			cntp = 2 * ncore
			Xp = mk(cntp)
			t = time.perf_counter()
			oracle.quad_batch(A, Xp, deg, orth, fun=args.fun, fresh_q=True, nthreads=ncore)
			tp = time.perf_counter() - t
			line["cpu_baseline_all_cores"] = {
				"value": round(cntp * deg / tp, 2), "unit": "probe-matvecs/s", "cores": ncore, "kind": "port",
				"sample": f"{cntp} probes, OpenMP over probes ({ncore} threads), {tp:.1f} s; an upper bound: the reference is single-threaded",
			}  # fmt: skip

	if rank == 0:
		print(json.dumps(line))
	if dist is not None:
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
