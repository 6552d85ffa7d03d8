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


def sharded_xtrace(M, count: int, batch: int = 128, pdf: str = "sphere", seed: int = 0, group=None, full: bool = False, device_rng: bool = True):
	"""XTrace of a device `MatrixFunction` with the f(A)-products of every block split by column over the
	ranks of `group` (BASELINE.json configs[2]: "512 probes, 1 -> 8 GPUs probe-sharded"). Every rank must
	call it with the same arguments; every rank returns the same estimate. Two all-gathers of n x batch
	doubles per block (Y = f(A) W and Z = f(A) Q_new) are the only collectives; the sample matrix W is
	replicated without communication because all ranks draw it from the same (seed, probe id) stream."""
	import torch.distributed as dist

	from .trace import xtrace

	if not dist.is_initialized() or dist.get_world_size(group) == 1:
		return xtrace(M, batch=batch, pdf=pdf, seed=seed, count=count, full=full, device_rng=device_rng)
	rank, world = dist.get_rank(group), dist.get_world_size(group)
	shard = (rank, world, lambda src, ncols, dst: allgather_columns(src, ncols, dst, group))
	return xtrace(M, batch=batch, pdf=pdf, seed=seed, count=count, full=full, device_rng=device_rng, _shard=shard)
