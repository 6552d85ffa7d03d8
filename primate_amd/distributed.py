"""Multi-GPU sharding of the probe set: one process per GPU, operator replicated, probes split, no
data-path collective; ONE reduction at the end.

Backend-agnostic on purpose: `torch.distributed` with backend "nccl" is RCCL over xGMI on the
MI355X node; the same code runs under "gloo" in the CPU tests.
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from .estimators import Covariance


def _collective_device(group, device: Optional[str]) -> str:
	"""Tensors of a collective live on this rank's GPU for RCCL ("nccl": LOCAL_RANK, as `engine.Context` picks it)
	and on the host for gloo."""
	import os

	import torch.distributed as dist

	if device:
		return device
	if dist.get_backend(group) != "nccl":
		return "cpu"
	import torch

	return f"cuda:{int(os.environ['LOCAL_RANK']) if 'LOCAL_RANK' in os.environ else torch.cuda.current_device()}"


def shard_range(nprobes: int, rank: int, world: int) -> tuple:
	"""Contiguous block [lo, hi) of global probe ids owned by `rank` (first ranks take the remainder)."""
	base, rem = divmod(int(nprobes), int(world))
	lo = rank * base + min(rank, rem)
	return lo, lo + base + (1 if rank < rem else 0)


def local_statistics(samples: np.ndarray) -> np.ndarray:
	"""(count, mean, M2) of the local per-probe values: the sufficient statistics of the trace
	estimator (src/primate/stats.py:47-49,77-86)."""
	x = np.asarray(samples, dtype=np.float64).ravel()
	if x.size == 0:
		return np.zeros(3)
	mu = x.mean()
	return np.array([x.size, mu, np.sum((x - mu) ** 2)])


def merge_statistics(stats: np.ndarray) -> tuple:
	"""Fold per-rank (count, mean, M2) rows in rank order with the batch-Welford formula of
	`Covariance.update` (stats.py:77-86). Returns (count, mean, sample variance)."""
	acc = Covariance(dim=1)
	for n, mu, m2 in np.asarray(stats, dtype=np.float64).reshape(-1, 3):
		acc.merge(int(n), np.array([mu]), np.array([[m2]]))
	var = acc.S.item() / (acc.n - 1) if acc.n > 1 else float("inf")
	return acc.n, acc.mu.item(), var


def allreduce_trace(samples: np.ndarray, group=None, device: Optional[str] = None, gather_samples: bool = False):
	"""The single collective of a sharded hutch(): all-gather each rank's (count, mean, M2) — 3 doubles
	per rank — and merge in rank order, so every rank ends with the same, order-deterministic result.
	With gather_samples=True the per-probe values themselves are gathered (<= a few thousand doubles)
	so the host can replay `MeanEstimator.update` in the reference's exact order (bit-identical to a
	1-GPU run)."""
	import torch
	import torch.distributed as dist

	world = dist.get_world_size(group)
	dev = _collective_device(group, device)
	if gather_samples:
		x = torch.as_tensor(np.asarray(samples, dtype=np.float64).ravel(), device=dev)
		sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
		dist.all_gather(sizes, torch.tensor([x.numel()], dtype=torch.int64, device=dev), group=group)
		mx = int(max(int(s.item()) for s in sizes))
		pad = torch.zeros(mx, dtype=torch.float64, device=dev)
		pad[: x.numel()] = x
		bufs = [torch.zeros(mx, dtype=torch.float64, device=dev) for _ in range(world)]
		dist.all_gather(bufs, pad, group=group)
		allx = np.concatenate([b[: int(s.item())].cpu().numpy() for b, s in zip(bufs, sizes)])
		n, mu, var = merge_statistics(local_statistics(allx))
		return n, mu, var, allx
	st = torch.as_tensor(local_statistics(samples), device=dev)
	bufs = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in range(world)]
	dist.all_gather(bufs, st, group=group)
	return merge_statistics(np.stack([b.cpu().numpy() for b in bufs]))


def allreduce_sum(x: np.ndarray, group=None, device: Optional[str] = None) -> np.ndarray:
	"""Sum an n-vector over ranks (the diag estimator's numer/denom)."""
	import torch
	import torch.distributed as dist

	dev = _collective_device(group, device)
	t = torch.as_tensor(np.ascontiguousarray(x), device=dev)
	dist.all_reduce(t, group=group)
	return t.cpu().numpy()


def sharded_hutch(evaluate, converge="default", batch: int = 32, group=None, full: bool = False, record: bool = False, callback=None, **kwargs):
	"""hutch() with an adaptive stopping rule over probe-sharded ranks (SURVEY.md §8e "Early stopping"): batch-synchronous.
	Every GLOBAL batch of `batch` probe ids [done, done + batch) is cut into contiguous per-rank blocks;
	`evaluate(lo, hi)` returns this rank's per-probe values for global ids [lo, hi); ONE all-gather per batch brings the
	values of all ranks together in id order, every rank folds them into its own copy of the estimator with
	`MeanEstimator.update` - the same call, on the same numbers, in the same order as the single-process
	`hutch(..., full=True)` makes per batch (src/primate/trace.py:104-110) - and evaluates the criterion on it, so all
	ranks stop at the same batch with the same estimate. `converge` / kwargs as in `hutch` (default: 200 samples or a
	95 % confidence interval of half-width 1, trace.py:89-92)."""
	import torch.distributed as dist

	from .estimators import ConfidenceCriterion, CountCriterion, EstimatorResult, MeanEstimator, convergence_criterion

	rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
	estimator = MeanEstimator(covariance=True, record=record)
	if isinstance(converge, str) and converge == "default":
		crit = CountCriterion(count=200) | ConfidenceCriterion(confidence=0.95, atol=1.0, rtol=0.0)
	else:
		crit = convergence_criterion(converge, **kwargs)
	result = EstimatorResult(estimator, crit)
	done = 0
	while not crit(estimator):
		lo, hi = shard_range(batch, rank, world)
		q = np.asarray(evaluate(done + lo, done + hi), dtype=np.float64).ravel() if hi > lo else np.zeros(0)
		assert q.size == hi - lo
		allq = q if world == 1 else allreduce_trace(q, group=group, gather_samples=True)[3]
		estimator.update(allq)
		done += batch
		if callback is not None:
			callback(result)
	result.message = crit.message(estimator)
	result.estimate, result.nit = estimator.estimate, len(estimator)
	return (estimator.estimate, result) if full else estimator.estimate


def sharded_hutch_device(op, nprobes: Optional[int] = None, deg: int = 20, orth: int = 3, fun="identity", pdf: str = "rademacher", seed: int = 0, rtol: float = 1e-8,
						 group=None, converge=None, batch: int = 32, full: bool = False, converge_kwargs: Optional[dict] = None, **fun_kwargs):  # fmt: skip
	"""hutch() with probes drawn on the device and sharded over the ranks of `group`. Global probe ids make the
	per-probe values independent of the number of GPUs.
	  * converge=None: fixed budget of `nprobes`, one all-gather of (count, mean, M2) at the end; returns
	    (count, mean, sample variance);
	  * converge="default" | "count" | "confidence" | "tolerance" | "knee" | a criterion: adaptive stopping,
	    evaluated per global batch of `batch` probes on the merged estimator (`sharded_hutch`); returns what
	    `hutch` returns (the estimate, or (estimate, EstimatorResult) with full=True). `converge_kwargs` are the
	    criterion's arguments (count, confidence, atol, rtol, ...)."""
	import torch.distributed as dist

	from .engine import LanczosPlan

	rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
	plans = {}

	def evaluate(lo: int, hi: int) -> np.ndarray:
		m = hi - lo
		if m not in plans:
			plans[m] = LanczosPlan(op, m, deg, orth)
		plan = plans[m]
		plan.generate_probes(pdf, seed=seed, probe_offset=lo)
		plan.run(rtol)
		return plan.quadrature(fun, **fun_kwargs)

	try:
		if converge is not None:
			return sharded_hutch(evaluate, converge=converge, batch=batch, group=group, full=full, **(converge_kwargs or {}))
		assert nprobes is not None, "a fixed budget needs nprobes"
		lo, hi = shard_range(nprobes, rank, world)
		q = evaluate(lo, hi) if hi > lo else np.zeros(0)
	finally:
		for pl in plans.values():
			pl.close()
	if world == 1:
		return merge_statistics(local_statistics(q))
	return allreduce_trace(q, group=group)


def sharded_diag_device(op, nprobes: int, deg: int, orth: int = 3, fun="identity", pdf: str = "rademacher", seed: int = 0, rtol: float = 1e-8, batch: int = 256, group=None, **fun_kwargs):
	"""diag f(A) with a fixed probe budget sharded over the ranks of `group` (BASELINE.json configs[3]):
	every rank accumulates numer += f(A)v * v and denom += v * v on its own GPU for its probe ids, then
	ONE all-reduce of the 2n-vector (numer, denom) combines them (16 MB at n = 2e6 fp32).
	Returns (numer / denom, numer, denom, count) — the Hutchinson diagonal estimate of the pooled probes.
	(The reference's running mean of successive ratios, diagonal.py:79, is order-dependent and is only
	reproduced by the single-process `primate_amd.diagonal.diag`.)"""
	import torch.distributed as dist

	from .engine import DiagAccumulator, LanczosPlan

	rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
	lo, hi = shard_range(nprobes, rank, world)
	n = op.shape[0]
	acc = DiagAccumulator(n, ctx=op.ctx)
	done = lo
	while done < hi:
		m = min(batch, hi - done)
		plan = LanczosPlan(op, m, deg, orth, keep_basis=True)
		plan.generate_probes(pdf, seed=seed, probe_offset=done)
		plan.run(rtol)
		acc.update(plan, fun, **fun_kwargs)
		plan.close()
		done += m
	numer, denom, _, cnt = acc.get()
	acc.close()
	if world > 1:
		both = allreduce_sum(np.concatenate([numer, denom, [float(cnt)]]), group=group)
		numer, denom, cnt = both[:n], both[n : 2 * n], int(round(both[-1]))
	with np.errstate(divide="ignore", invalid="ignore"):
		est = numer / denom
	return est, numer, denom, cnt


def allgather_columns(src, ncols: int, dst, group=None):
	"""dst[:, r*ncols:(r+1)*ncols] = rank r's src[:, :ncols] for every rank r: the exchange step of a
	column-sharded f(A)-product (xtrace). Backend "nccl" (RCCL over xGMI) gathers the device
	buffers in place through zero-copy torch views; any other backend (gloo in the tests) stages through
	the host."""
	import torch
	import torch.distributed as dist

	world = dist.get_world_size(group)
	if dist.get_backend(group) == "nccl":
		dev = f"cuda:{src.ctx.device}"  # the Context's own (resolved) ordinal, not torch's current device
		tin = torch.as_tensor(src.cuda_array(0, ncols), device=dev)
		tout = torch.as_tensor(dst.cuda_array(0, ncols * world), device=dev)
		src.ctx.synchronize()  # libslq writes on its own stream
		dist.all_gather_into_tensor(tout, tin, group=group)
		torch.cuda.synchronize(dev)
		return
	loc = torch.from_numpy(np.ascontiguousarray(src.get(0, ncols).T))  # (ncols, n): rows = columns
	parts = [torch.empty_like(loc) for _ in range(world)]
	dist.all_gather(parts, loc, group=group)
	dst.set(0, np.concatenate([p.numpy().T for p in parts], axis=1))


def alltoall_blocks(src, dst, group=None):
	"""dst block q = rank q's src block of this rank's number: src and dst are device matrices of `world` equal column blocks
	(column-major, so a block is one contiguous run). Backend "nccl" (RCCL over xGMI) exchanges the device buffers in place
	through zero-copy torch views; any other backend (gloo in the tests) stages through the host."""
	import torch
	import torch.distributed as dist

	if dist.get_backend(group) == "nccl":
		dev = f"cuda:{src.ctx.device}"
		tin = torch.as_tensor(src.cuda_array(0, src.cols), device=dev)
		tout = torch.as_tensor(dst.cuda_array(0, dst.cols), device=dev)
		src.ctx.synchronize()  # libslq writes on its own stream
		dist.all_to_all_single(tout, tin, group=group)
		torch.cuda.synchronize(dev)
		return
	loc = torch.from_numpy(np.ascontiguousarray(src.get().T))  # (cols, rows) row-major = the column-major matrix, flat
	out = torch.empty_like(loc)
	dist.all_to_all_single(out, loc, group=group)
	dst.set(0, out.numpy().T)


def _xtrace_row_sharded(M, count: int, batch: int, pdf: str, seed: int, group, full: bool):
	"""XTrace with ROW-SHARDED sketches (the estimator of src/primate/trace.py:233-315, whose n x m matrices
	W, Q, Z at :296-302 are what grows): rank r keeps rows [r nr, (r+1) nr) of W, Q and Z - n m / world doubles each instead of
	n m - and every tall-skinny product of the block Gram-Schmidt, of CholeskyQR2 and of the three m x m summaries is a LOCAL
	product over those rows followed by ONE all-reduce of an m x ns or m x m matrix. The f(A)-products stay column-sharded
	(a Lanczos run needs whole vectors): an all-to-all turns this rank's full columns into every rank's rows of them
	(n ns / world doubles sent per rank and block, against n ns received in the replicated form's all-gather), and the
	reverse all-to-all hands the new Q columns to the ranks that apply f(A) to them. Same estimator, same probe stream
	(global probe ids) as `xtrace(..., device_rng=True)`; the sums run in a different order: rounding only."""
	import torch.distributed as dist
	from scipy.linalg import cholesky, solve_triangular

	from . import engine
	from .estimators import CountCriterion, EstimatorResult, MeanEstimator
	from .trace import _leave_one_out_estimates

	rank, world = dist.get_rank(group), dist.get_world_size(group)
	n = M.shape[0]
	name, kw = M._builtin
	ctx = M._op.ctx
	P = min(int(count), n)
	nr = -(-n // world)  # rows per rank; the last ranks' shards are shorter (zero rows behind them: nothing to any Gram matrix)
	rows_of = lambda q: max(0, min(nr, n - q * nr))  # noqa: E731
	row0 = lambda q: min(q * nr, n)  # noqa: E731  (more ranks than rows: the shards past the end are empty and start AT the end, not beyond it)
	cmax = -(-int(batch) // world)  # columns per rank and block, padded to equal shards for the all-to-all
	Wd, Qd, Zd = (engine.DeviceMatrix(nr, P, ctx=ctx) for _ in range(3))
	Yd, Td = engine.DeviceMatrix(nr, batch, ctx=ctx), engine.DeviceMatrix(nr, batch, ctx=ctx)
	Wfull = engine.DeviceMatrix(n, batch, ctx=ctx)  # a block's probes, whole columns (every rank draws all of them: same ids, same values)
	Xfull, Yfull = engine.DeviceMatrix(n, cmax, ctx=ctx), engine.DeviceMatrix(n, cmax, ctx=ctx)  # this rank's columns, whole: in and out of f(A)
	SBf, SBb, RB = (engine.DeviceMatrix(nr, cmax * world, ctx=ctx) for _ in range(3))
	mats = (Wd, Qd, Zd, Yd, Td, Wfull, Xfull, Yfull, SBf, SBb, RB)
	allsum = lambda X: allreduce_sum(X, group=group)  # noqa: E731

	def shard(ns: int, q: int):
		base, rem = divmod(ns, world)
		return q * base + min(q, rem), base + (1 if q < rem else 0)

	## the two plan shapes of a block - all ns probes (drawn by every rank) and this rank's nloc columns of them - live side by side for the
	## whole call: M._plan keeps ONE plan per shape class, so asking it for both in turn would destroy and re-create two kept-basis plans
	## (a ring of deg + 1 panels each) in every block
	own_plans = {}

	def plan_for(k: int):
		if k not in own_plans:
			if len(own_plans) >= 3:  # (full blocks: ns and nloc; the last, shorter block: two more - drop the oldest)
				own_plans.pop(next(iter(own_plans))).close()
			own_plans[k] = engine.LanczosPlan(M._op, k, M._deg, M._orth, keep_basis=True)
		return own_plans[k]

	def apply_fun(src, c0: int, nloc: int):
		"""Yfull[:, :nloc] = f(A) src[:, c0:c0+nloc] (whole columns, this rank's share of the block)"""
		if nloc > 0:
			plan = plan_for(nloc)
			plan.set_probes_device(src.col_ptr(c0))
			plan.run(M._rtol)
			plan.fun_action_into(Yfull, 0, name, **kw)

	def rows_from_columns(dst, o0: int, ns: int):
		"""dst[:, o0:o0+ns] (local rows) = the block whose column shards sit in every rank's Yfull"""
		for q in range(world):
			SBf.copy_rows_from(q * cmax, 0, Yfull, 0, row0(q), rows_of(q), cmax)
		alltoall_blocks(SBf, RB, group)
		for q in range(world):
			lo, nq = shard(ns, q)
			if nq > 0:
				dst.copy_from(o0 + lo, RB, q * cmax, nq)

	def columns_from_rows(src, c0: int, ns: int):
		"""Xfull[:, :nloc] (whole columns) = this rank's column shard of src[:, c0:c0+ns], whose rows are spread over the ranks"""
		for q in range(world):
			lo, nq = shard(ns, q)
			if nq > 0:
				SBb.copy_from(q * cmax, src, c0 + lo, nq)
		alltoall_blocks(SBb, RB, group)
		for q in range(world):
			Xfull.copy_rows_from(0, row0(q), RB, q * cmax, 0, rows_of(q), cmax)

	R, R_inv = np.zeros((0, 0)), np.zeros((0, 0))
	result = EstimatorResult()
	m = 0
	try:
		while m < P:
			ns = min(int(batch), P - m)
			lo, nloc = shard(ns, rank)
			plan = plan_for(ns)
			plan.generate_probes(pdf, seed=seed, probe_offset=m)
			plan.get_probes_into(Wfull, 0)
			Wd.copy_rows_from(m, 0, Wfull, 0, row0(rank), rows_of(rank), ns)
			apply_fun(Wfull, lo, nloc)
			rows_from_columns(Yd, 0, ns)
			Cm = np.zeros((m, ns))
			if m > 0:
				for _ in range(2):  # block Gram-Schmidt against the existing Q, twice
					Ci = allsum(Qd.tn(0, m, Yd, 0, ns))
					Yd.add_product(0, Qd, 0, Ci, alpha=-1.0, beta=1.0)
					Cm += Ci
			## CholeskyQR2 on the row shards: the two Gram matrices are the only things exchanged
			try:
				R1 = cholesky(allsum(Yd.tn(0, ns, Yd, 0, ns)), lower=False)
				Td.add_product(0, Yd, 0, solve_triangular(R1, np.eye(ns)), alpha=1.0, beta=0.0)
				R2 = cholesky(allsum(Td.tn(0, ns, Td, 0, ns)), lower=False)
				Qd.add_product(m, Td, 0, solve_triangular(R2, np.eye(ns)), alpha=1.0, beta=0.0)
				Rn = R2 @ R1
			except np.linalg.LinAlgError:
				## numerically rank-deficient block (every rank sees the same Gram matrix, so every rank lands here together):
				## Householder QR of the whole block on the host, every rank keeps its rows
				Yall = np.zeros((world * nr, ns))
				Yall[rank * nr : rank * nr + nr] = Yd.get(0, ns)
				Qn, Rn = np.linalg.qr(allsum(Yall)[:n], mode="reduced")
				mine = np.zeros((nr, ns))
				mine[: rows_of(rank)] = Qn[rank * nr : rank * nr + rows_of(rank)]
				Qd.set(m, mine)
			R_new = np.zeros((m + ns, m + ns))
			R_new[:m, :m], R_new[:m, m:], R_new[m:, m:] = R, Cm, Rn
			Rn_inv = solve_triangular(Rn, np.eye(ns))
			Ri_new = np.zeros((m + ns, m + ns))
			Ri_new[:m, :m], Ri_new[m:, m:] = R_inv, Rn_inv
			Ri_new[:m, m:] = -R_inv @ Cm @ Rn_inv
			R, R_inv = R_new, Ri_new
			columns_from_rows(Qd, m, ns)
			apply_fun(Xfull, 0, nloc)
			rows_from_columns(Zd, m, ns)
			m += ns
			## the three m x m summaries in one all-reduce
			S3 = allsum(np.stack([Qd.tn(0, m, Wd, 0, m), Qd.tn(0, m, Zd, 0, m), Zd.tn(0, m, Wd, 0, m)]))
			## (pdf = None: no sphere rescaling - the reference rebinds `pdf` to its sampler before the test `pdf == "sphere"`, src/primate/trace.py:295,
			## 305 against :207, so that branch never runs there either; primate_amd/trace.py says so at its own call)
			t_samples = _leave_one_out_estimates(n, S3[0], S3[1], S3[2], R, R_inv, None)
			estimator = MeanEstimator()
			estimator.update(t_samples.ravel())
			result.estimator, result.estimate, result.nit = estimator, estimator.estimate, m
	finally:
		for d in mats:
			d.close()
		for pl in own_plans.values():
			pl.close()
	result.criterion = CountCriterion(count=P)
	return (result.estimate, result) if full else result.estimate


def sharded_xtrace(M, count: int, batch: int = 128, pdf: str = "sphere", seed: int = 0, group=None, full: bool = False, device_rng: bool = True, sketches: str = "replicated"):
	"""XTrace of a device `MatrixFunction` with the f(A)-products of every block split by column over the
	ranks of `group` (BASELINE.json configs[2]: "512 probes, 1 -> 8 GPUs probe-sharded"). Every rank must
	call it with the same arguments; every rank returns the same estimate. Two all-gathers of n x batch
	doubles per block (Y = f(A) W and Z = f(A) Q_new) are the only collectives; the sample matrix W is
	replicated without communication because all ranks draw it from the same (seed, probe id) stream.
	sketches="rows": the n x m sketches W, Q, Z are ROW-sharded instead (`_xtrace_row_sharded`: 1/world of the memory and of the
	Gram-matrix work per rank, one m x m all-reduce per Gram matrix, all-to-alls instead of all-gathers) - for n m beyond one
	GPU's HBM or blocks wide enough for the dense algebra to matter; needs the device probe stream."""
	import torch.distributed as dist

	from .trace import xtrace

	assert sketches in ("replicated", "rows"), "sketches is 'replicated' or 'rows'"
	if sketches == "rows" and dist.is_initialized():  # (also for a world of one: the same code path, its collectives included)
		assert device_rng and isinstance(pdf, str), "row-sharded sketches draw their probes on the device"
		return _xtrace_row_sharded(M, count, batch, pdf, int(seed), group, full)
	if not dist.is_initialized() or dist.get_world_size(group) == 1:
		return xtrace(M, batch=batch, pdf=pdf, seed=seed, count=count, full=full, device_rng=device_rng)
	rank, world = dist.get_rank(group), dist.get_world_size(group)
	shard = (rank, world, lambda src, ncols, dst: allgather_columns(src, ncols, dst, group))
	return xtrace(M, batch=batch, pdf=pdf, seed=seed, count=count, full=full, device_rng=device_rng, _shard=shard)
