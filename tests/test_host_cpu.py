"""CPU-side tests (no GPU): the C-ABI library loads and exports what include/slq.h declares, host
logic (function registry, probe stream, estimators, hutch bookkeeping, byte model), and the
world_size-2 gloo rehearsal of the probe-sharded reduction."""

import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
	import __graft_entry__ as g
	from primate_amd import _capi

	g.build_libslq()
	hdr = (ROOT / "include" / "slq.h").read_text()
	hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
	declared = set(re.findall(r"\b(slq_[A-Za-z0-9_]+)\s*\(", hdr)) - {"slq_matvec_fn"}
	assert len(declared) >= 30
	L = _capi.lib()
	missing = [s for s in declared if not hasattr(L, s)]
	assert not missing, missing
	assert declared == set(_capi.EXPORTED_SYMBOLS), declared ^ set(_capi.EXPORTED_SYMBOLS)
	assert L.slq_version() == 100


def test_ring_pass_bail_out_is_reported_not_swallowed():
	"""The ring-fed tile pass bounds every wait; a workgroup that gives up raises a device word, and the accessors turn
	it into SLQ_EHIP through ONE translation (slq.hip: ring_flag_status). That translation needs no device: a clear flag
	is SLQ_OK, a raised one SLQ_EHIP with a message that names the pass. (The device side of the report - each accessor
	reading the word - is tests/test_gpu_api.py::test_ring_bail_out_flag_reaches_every_accessor.)"""
	from primate_amd import _capi

	L = _capi.lib()
	assert L.slq_debug_ring_flag_status(0) == _capi.SLQ_OK
	for raised in (1, 7, -1):
		assert L.slq_debug_ring_flag_status(raised) == _capi.SLQ_EHIP
		msg = L.slq_last_error().decode()
		assert "ring-fed tile pass" in msg and "invalid" in msg
	with pytest.raises(_capi.SlqError) as ei:
		_capi.check(L.slq_debug_ring_flag_status(1))
	assert ei.value.code == _capi.SLQ_EHIP


def test_no_gpu_fails_loudly_not_silently():
	import torch

	if torch.cuda.is_available():
		pytest.skip("a GPU is visible")
	from primate_amd import _capi, engine

	with pytest.raises(_capi.SlqError) as ei:
		engine.Context(device=0)
	assert ei.value.code == _capi.SLQ_ENODEV and "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
	for p in (ROOT / "primate_amd").rglob("*"):
		if p.suffix in {".py", ".hip", ".hpp", ".h"}:
			txt = p.read_text()
			assert "import oracle" not in txt and "from oracle" not in txt and "slq_oracle" not in txt, p


def test_function_registry_matches_golden(golden):
	from primate_amd.engine import fun_spec
	from primate_amd.special import _BUILTIN_MATRIX_FUNCTIONS, builtin_spec, param_callable

	x = golden["fun_x"]
	with np.errstate(all="ignore"):
		for name, kw in [("identity", {}), ("log", {}), ("exp", {}), ("sqrt", {}), ("inv", {}), ("abs", {}), ("smoothstep", {"a": 0.5, "b": 6.0}), ("numrank", {}), ("softsign", {"q": 10})]:
			f = param_callable(name, **dict(kw))
			r, ref = np.asarray(f(x), dtype=float), golden[f"fun_{name}"]
			ok = np.isfinite(ref)
			assert np.array_equal(r[ok], ref[ok]), name
			assert builtin_spec(f)[0] == name
		assert np.array_equal(param_callable("exp", t=-0.1)(x), golden["fun_exp_t"])
	assert _BUILTIN_MATRIX_FUNCTIONS == ["identity", "abs", "sqrt", "log", "inv", "exp", "smoothstep", "numrank"]
	assert fun_spec("numrank")[0] == 7 and list(fun_spec("numrank")[1][:2]) == [1e-6, 1.0]
	assert fun_spec("exp", t=-2.0)[1][0] == -2.0 and fun_spec(np.log) == (None, None)
	with pytest.raises(AssertionError):
		param_callable("nope")


def test_probe_stream_order_contract(golden):
	from primate_amd.random import isotropic

	for pdf in ["rademacher", "normal", "sphere"]:
		a = isotropic((37, 5), pdf=pdf, seed=1234)
		assert np.array_equal(a, golden[f"iso_{pdf}_37x5_s1234"]) and a.flags["F_CONTIGUOUS"] and a.dtype == np.float64
		g = isotropic(pdf=pdf, seed=99)
		b = np.column_stack([g(size=(11, 1)), g(size=(11, 2)), g(size=(11, 1))])
		assert np.array_equal(b, golden[f"iso_{pdf}_seq_s99"])
	assert set(np.unique(isotropic((50, 4), pdf="signs", seed=1))) == {-1.0, 1.0}
	assert np.allclose(np.linalg.norm(isotropic((50, 4), pdf="sphere", seed=1), axis=0), np.sqrt(50))
	with pytest.raises(AssertionError):
		isotropic((3, 3), pdf="cauchy")


def test_streaming_estimator_and_merge(golden):
	from primate_amd.estimators import ConfidenceCriterion, CountCriterion, Covariance, MeanEstimator, ToleranceCriterion, convergence_criterion

	xs, tr = golden["est_samples"], golden["est_trail"]
	cov, est = Covariance(1), MeanEstimator(covariance=True)
	for k, (lo, hi) in enumerate([(0, 1), (1, 9), (9, 41), (41, 57)]):
		cov.update(xs[lo:hi])
		est.update(xs[lo:hi])
		assert np.array_equal([cov.n, cov.mu.item(), cov.S.item(), est.estimate, np.ravel(est.delta)[0]], tr[k])
	assert len(est) == 57 and np.isclose(cov.covariance(), np.var(xs, ddof=1))
	a, b, c = Covariance(1), Covariance(1), Covariance(1)
	a.update(xs[:20]); b.update(xs[20:]); c.update(xs)  # noqa: E702
	a.merge(b.n, b.mu, b.S)
	assert a.n == c.n and np.allclose(a.mu, c.mu, rtol=1e-14) and np.allclose(a.S, c.S, rtol=1e-12)
	cc = CountCriterion(10) | ConfidenceCriterion(0.95, atol=1e-9, rtol=0.0)
	e2 = MeanEstimator(covariance=True)
	assert not cc(e2)
	e2.update(np.arange(10.0))
	assert cc(e2) and not (~cc)(e2) and not (cc & ToleranceCriterion(rtol=0, atol=0))(e2)
	assert isinstance(convergence_criterion("count", count=3, junk=1), CountCriterion)


def test_hutch_bookkeeping_without_a_gpu(golden):
	"""hutch() on plain arrays never touches the device: checks the sample-at-a-time semantics
	against the reference's own outputs ("pure" golden)."""
	from primate_amd.trace import hutch

	assert hutch(golden["dense_A"], converge="count", count=64, seed=1234) == pytest.approx(float(golden["dense_hutch_c64"]), rel=1e-13)
	assert hutch(golden["sym_A"], converge="count", count=150, seed=1234) == pytest.approx(float(golden["sym_hutch_c150"]), rel=1e-13)

	class FakeQuad:  # an operator exposing .quad, as trace.py:97 prefers
		shape, dtype = (20, 20), np.dtype(np.float64)
		seen = []

		def __matmul__(self, v):
			return v

		def quad(self, V):
			FakeQuad.seen.append(V.shape[1])
			return np.einsum("ij,ij->j", V, V)

	est, info = hutch(FakeQuad(), converge="count", count=70, seed=0, full=True, batch=32)
	assert est == 20.0 and info.nit == 96 and FakeQuad.seen == [32, 32, 32]
	FakeQuad.seen.clear()
	assert hutch(FakeQuad(), converge="count", count=70, seed=0) == 20.0 and sum(FakeQuad.seen) == 70


def test_bench_byte_model():
	sys.path.insert(0, str(ROOT))
	import bench

	n, nnz, s, b = 1_000_000, 4_996_000, 8, 256
	## SURVEY.md §8(d) figures: 64.25 MB per probe-matvec at orth = 0, b = 256
	B0 = bench.contract_bytes_per_probe_matvec(n, nnz, s, b, 5, 0)
	assert abs(B0 - (64e6 + (12 * nnz + 4 * (n + 1)) / 256)) < 1 and abs(B0 / 1e6 - 64.25) < 0.01
	assert bench.contract_bytes_per_probe_matvec(n, nnz, s, b, 10, 3) == pytest.approx(112e6 + 0.25e6, rel=1e-3)
	kb, kl = bench.kernel_bytes(n, nnz, s, b, 128, 30, 0, fused=False)
	assert kl == {"spmm_3term": 30, "axpy_norm": 31, "reorth_dot": 0, "reorth_update": 0}
	vec = s * n * b
	csr = 2 * (12 * nnz + 4 * (n + 1))
	assert kb["spmm_3term"] == 30 * csr + (2 + 29 * 3) * vec
	kb, kl = bench.kernel_bytes(n, nnz, s, b, 128, 30, 30, fused=False)
	assert kl["reorth_dot"] == 30 + 14 and kl["reorth_update"] == 30
	assert kb["reorth_update"] == sum(min(j + 1, 30) + 2 for j in range(30)) * vec
	## fused passes (r <= 4): alpha pass reads 1 panel over the upper triangle, update pass reads 2 (+r-2) and writes 1
	csr_u = 2 * (12 * ((nnz + n) // 2) + 4 * (n + 1))
	kb, kl = bench.kernel_bytes(n, nnz, s, b, 128, 30, 0, fused=True, last_nostore=False)
	assert kl == {"spmm_3term": 30, "axpy_norm": 31, "reorth_dot": 0, "reorth_update": 0}
	assert kb["spmm_3term"] == 30 * csr_u + 30 * vec and kb["axpy_norm"] == vec + 30 * csr + (2 + 29 * 3) * vec
	## r04: the last step's update pass stores nothing (W_deg is never read), and Rademacher probes drawn on the device need no norm sweep
	kb2, kl2 = bench.kernel_bytes(n, nnz, s, b, 128, 30, 0, fused=True)
	assert kb2["axpy_norm"] == kb["axpy_norm"] - vec and kl2 == kl
	kb3, kl3 = bench.kernel_bytes(n, nnz, s, b, 128, 30, 0, fused=True, norm_sweep=False)
	assert kb3["axpy_norm"] == kb2["axpy_norm"] - vec and kl3["axpy_norm"] == 30
	kb, kl = bench.kernel_bytes(n, nnz, s, b, 128, 30, 3, fused=True)
	assert kl["reorth_dot"] == 30 and kb["reorth_dot"] == 30 * csr + (1 + 2 + 28 * 3) * vec
	## deeper reorthogonalisation falls back to the store-and-revisit sweeps once r_j > 4
	kb5, kl5 = bench.kernel_bytes(n, nnz, s, b, 128, 30, 30, fused=True)
	assert kl5["reorth_dot"] == 30 + 14


def test_shard_ranges_cover_and_are_disjoint():
	from primate_amd.distributed import shard_range

	for P in [1, 7, 256, 2048]:
		for W in [1, 2, 3, 8]:
			r = [shard_range(P, k, W) for k in range(W)]
			assert r[0][0] == 0 and r[-1][1] == P and all(r[k][1] == r[k + 1][0] for k in range(W - 1))
			assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


_GLOO_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from primate_amd.distributed import allreduce_trace, allreduce_sum, shard_range, local_statistics, merge_statistics
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(42)
allq = rng.standard_normal(257) * 5 + 100          # stands in for the per-probe quad values
lo, hi = shard_range(len(allq), rank, world)
n, mu, var = allreduce_trace(allq[lo:hi])
n2, mu2, var2, xs = allreduce_trace(allq[lo:hi], gather_samples=True)
tot = allreduce_sum(np.full(5, rank + 1.0))
ok = (n == 257 and abs(mu - allq.mean()) < 1e-12 and abs(var - allq.var(ddof=1)) < 1e-10
      and np.array_equal(xs, allq) and mu2 == merge_statistics(local_statistics(allq))[1]
      and np.array_equal(tot, np.full(5, world * (world + 1) / 2)))
print("RANK", rank, "OK" if ok else "FAIL", n, mu, var)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def test_sharded_reduce_world2_gloo(tmp_path):
	script = tmp_path / "worker.py"
	script.write_text(_GLOO_WORKER)
	env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2")
	procs = [subprocess.Popen([sys.executable, str(script), str(ROOT)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
	outs = [p.communicate(timeout=300)[0].decode() for p in procs]
	assert all(p.returncode == 0 for p in procs), outs
	assert all("OK" in o for o in outs), outs


_GLOO_ADAPTIVE_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from primate_amd.distributed import sharded_hutch
from primate_amd.trace import hutch
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank = dist.get_rank()
vals = np.random.default_rng(7).standard_normal(4096) * 3.0 + 50.0   # per-probe values by GLOBAL probe id

class Fake:   # what hutch needs of an operator with device-drawn probes
	shape, dtype = (64, 64), np.dtype(np.float64)
	def matvec(self, x): return x
	def quad_generated(self, m, pdf, seed, offset): return vals[offset:offset + m]

ok = True
for conv, kw in (("confidence", dict(confidence=0.95, atol=0.4, rtol=0.0)), ("default", {}), ("count", dict(count=70)), ("tolerance", dict(atol=0.0, rtol=2e-3))):
	ref, rinfo = hutch(Fake(), pdf="device:rademacher", converge=conv, batch=24, seed=1, full=True, **kw)
	got, ginfo = sharded_hutch(lambda lo, hi: vals[lo:hi], converge=conv, batch=24, full=True, **kw)
	same = (got == ref and ginfo.nit == rinfo.nit and ginfo.message == rinfo.message)
	print("RANK", rank, conv, "nit", ginfo.nit, rinfo.nit, got, ref, "same" if same else "DIFFERENT")
	ok = ok and same and ginfo.nit % 24 == 0 and ginfo.nit < 4096
print("RANK", rank, "OK" if ok else "FAIL")
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def test_sharded_adaptive_stopping_world2_gloo(tmp_path):
	"""SURVEY.md §8(e) "Early stopping": the batch-synchronous sharded hutch stops at the same batch, with the same
	estimate and message, as the single-process hutch(full=True) on the same per-probe values, for every criterion."""
	script = tmp_path / "worker_adaptive.py"
	script.write_text(_GLOO_ADAPTIVE_WORKER)
	env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", WORLD_SIZE="2")
	procs = [subprocess.Popen([sys.executable, str(script), str(ROOT)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
	outs = [p.communicate(timeout=300)[0].decode() for p in procs]
	assert all(p.returncode == 0 for p in procs), outs
	assert all("OK" in o for o in outs), outs


def test_toeplitz_plugin_operator_on_the_host():
	"""The FFT-applied Toeplitz plugin equals the dense Toeplitz product (symmetric and non-symmetric)."""
	from scipy.linalg import toeplitz

	from primate_amd.operators import Toeplitz, is_linear_op, is_valid_operator

	rng = np.random.default_rng(3)
	c, r = rng.standard_normal(17), rng.standard_normal(17)
	r[0] = c[0]
	x = rng.standard_normal(17)
	np.testing.assert_allclose(Toeplitz(c) @ x, toeplitz(c) @ x, rtol=1e-12, atol=1e-12)
	np.testing.assert_allclose(Toeplitz(c, r) @ x, toeplitz(c, r) @ x, rtol=1e-12, atol=1e-12)
	assert is_linear_op(Toeplitz(c)) and is_valid_operator(Toeplitz(c)) == np.float64


def test_parallel_rademacher_fill_is_the_same_stream(monkeypatch):
	"""Large Rademacher draws are filled by several threads from PCG64 copies advanced to their offsets: the
	values, and the state the caller's generator is left in, are those of the one-thread draw."""
	import primate_amd.random as R

	outs, tails = [], []
	for threshold in (1 << 62, 1 << 10):
		monkeypatch.setattr(R, "_PARALLEL_MIN", threshold)
		rng = np.random.default_rng(99)
		W = R.isotropic((5003, 37), pdf="rademacher", seed=rng)
		V = R.isotropic((5003, 3), pdf="rademacher", seed=rng)  # continues the same generator
		outs.append(np.c_[W, V])
		tails.append(rng.random(4))
	assert np.array_equal(outs[0], outs[1]) and np.array_equal(tails[0], tails[1])
	assert set(np.unique(outs[1])) == {-1.0, 1.0}


def test_bench_starts_its_own_ranks(capfd):
	"""`python bench.py --gpus N` without a launcher: the parent starts N child ranks with RANK / LOCAL_RANK / WORLD_SIZE /
	MASTER_* set, relays rank 0's stdout only, and returns the worst exit code (one process per GPU). The
	children here are stubs - no torch, no GPU."""
	import bench

	stub = "import os, sys; r = int(os.environ['RANK']); print('[lib] chatter on stdout'); print('{line', r, os.environ['LOCAL_RANK'], os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'], bool(int(os.environ['MASTER_PORT']))); sys.exit(int(sys.argv[1]) if r == int(sys.argv[2]) else 0)"
	rc = bench.spawn_ranks(3, ["0", "0"], cmd=[sys.executable, "-c", stub])
	out, err = capfd.readouterr()
	assert rc == 0 and out == "{line 0 0 3 127.0.0.1 True\n"  # rank 0's record and nothing else on stdout
	assert "line 1 1 3" in err and "line 2 2 3" in err and "[lib] chatter" in err
	rc = bench.spawn_ranks(2, ["5", "1"], cmd=[sys.executable, "-c", stub])  # rank 1 fails with 5
	out, err = capfd.readouterr()
	assert rc == 5 and out.startswith("{line 0 0 2") and "exit codes [0, 5]" in err
	## a rank that hangs is ended by the parent (its own child, by handle), the others' result is kept
	hang = "import os, sys, time; r = int(os.environ['RANK']); print('{up', r, flush=True); time.sleep(600 if r == 1 else 0)"
	rc = bench.spawn_ranks(2, [], cmd=[sys.executable, "-c", hang], timeout=3)
	out, err = capfd.readouterr()
	assert rc != 0 and out == "{up 0\n"
	## a rank that DIES while rank 0 is blocked (in a collective, on the real thing): the parent must not sit in rank 0's
	## pipe until a watchdog fires - it ends rank 0 and reports the failure at once
	import time

	die = "import os, sys, time; r = int(os.environ['RANK']); print('{up', r, flush=True); time.sleep(600) if r == 0 else sys.exit(7)"
	t0 = time.monotonic()
	rc = bench.spawn_ranks(2, [], cmd=[sys.executable, "-c", die], timeout=300)
	out, err = capfd.readouterr()
	assert rc != 0 and time.monotonic() - t0 < 60 and "exit codes" in err


def test_bench_strong_scaling_shards():
	"""`bench.py --scaling strong`: --probes is the global batch of a step and rank r advances shard_range(P, r, N) of it
	(the reduction the shards feed: src/primate/stats.py:77-86). The bench's split is the library's, covers
	every probe id once, and 256 probes over 8 GPUs are the 32-probe panels hutch() itself draws (src/primate/trace.py:36)."""
	import bench
	from primate_amd.distributed import shard_range

	for P, N in ((256, 8), (256, 3), (7, 8), (512, 4)):
		blocks = [bench.shard_range(P, r, N) for r in range(N)]
		assert blocks == [shard_range(P, r, N) for r in range(N)]
		assert blocks[0][0] == 0 and blocks[-1][1] == P and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
	assert bench.shard_range(256, 5, 8) == (160, 192)
	## argument plumbing: the flag exists, defaults to weak, and the launcher passes it on untouched
	import subprocess

	h = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True).stdout
	assert "--scaling" in h and "strong" in h and "--timeout" in h
