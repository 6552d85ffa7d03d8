"""Multi-process runs of the sharded drivers on the GPU box. The box has ONE GPU, so the 2-rank case is a
rehearsal: both ranks use device 0 and the collective goes through gloo (host staging); the nccl (RCCL)
code path of the same function is exercised with a world of one rank."""

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import laplacian_2d

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _launch(nproc: int, backend: str, out: Path, port: int, share_gpu0: bool):
	env = dict(os.environ, MASTER_ADDR="127.0.0.1")
	if share_gpu0:
		env["DIST_TEST_SHARE_GPU0"] = "1"
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
	       "--master-port", str(port), str(ROOT / "tests" / "_dist_xtrace_worker.py"), backend, str(out)]  # fmt: skip
	r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
	assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
	return [json.load(open(f"{out}.rank{k}.json")) for k in range(nproc)]


@pytest.mark.parametrize("world,port", [(2, 29531), (3, 29533)])
def test_sharded_xtrace_ranks_match_single_process(tmp_path, world, port):
	"""world = 3 makes the column shards ragged (blocks of 20 -> 7, 7, 6 and the last block of 10 -> 4, 3, 3)."""
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import xtrace

	L = laplacian_2d(40)
	M = MatrixFunction(L, fun="exp", deg=20, orth=3, t=-0.5)
	single = xtrace(M, batch=20, pdf="sphere", seed=7, count=50, device_rng=True)
	res = _launch(world, "gloo", tmp_path / f"g{world}", port, share_gpu0=True)
	assert all(r["gather_ok"] and r["nit"] == 50 for r in res)
	assert len({r["estimate"] for r in res}) == 1  # replicated algebra on identical inputs
	## row-sharded sketches (world = 3: 1600 rows -> 534, 534, 532): every rank the same number, equal to one process to rounding
	assert all(r["nit_rows"] == 50 for r in res) and len({r["estimate_rows"] for r in res}) == 1
	assert res[0]["estimate_rows"] == pytest.approx(single, rel=1e-9)
	## probe-sharded hutch / diag: global probe ids, so the pooled statistics equal the single-process ones
	from primate_amd.distributed import sharded_diag_device, sharded_hutch_device

	cnt, mean, var = sharded_hutch_device(M._op, 45, 20, 3, fun="exp", seed=13, t=-0.5)
	est, numer, denom, c = sharded_diag_device(M._op, 35, 20, 3, fun="exp", seed=13, batch=8, t=-0.5)
	for r in res:
		assert r["hutch"][0] == cnt == 45 and r["hutch"][1] == pytest.approx(mean, rel=1e-12) and r["hutch"][2] == pytest.approx(var, rel=1e-9)
		assert r["diag"][0] == c == 35 and r["diag"][3] == pytest.approx(float(np.sum(denom)), rel=1e-13)
		assert r["diag"][2] == pytest.approx(float(np.sum(numer)), rel=1e-10) and r["diag"][1] == pytest.approx(float(np.sum(est)), rel=1e-10)
	## adaptive stopping: same stopping batch and (to rounding: narrower panels per rank) the same estimate as one process
	e1, i1 = sharded_hutch_device(M._op, None, 20, 3, fun="exp", seed=13, converge="confidence", batch=12, full=True,
								  converge_kwargs=dict(confidence=0.95, atol=0.0, rtol=0.02), t=-0.5)  # fmt: skip
	assert 12 <= i1.nit < 600 and i1.nit % 12 == 0
	for r in res:
		assert r["adaptive"][1] == i1.nit and r["adaptive"][0] == pytest.approx(e1, rel=1e-11)
	## shards of 10 columns run in a narrower panel geometry than the single 20-column batch: rounding only
	assert res[0]["estimate"] == pytest.approx(single, rel=1e-9)
	exact = np.sum(np.exp(-0.5 * np.linalg.eigvalsh(L.toarray())))
	assert abs(single - exact) / exact < 2e-2


def test_rccl_allgather_code_path_world_of_one(tmp_path):
	res = _launch(1, "nccl", tmp_path / "n1", 29532, share_gpu0=False)
	assert res[0]["gather_ok"] and res[0]["nit"] == 50 and res[0]["nit_rows"] == 50  # (all_gather_into_tensor and all_to_all_single on libslq's buffers)
	assert res[0]["estimate_rows"] == pytest.approx(res[0]["estimate"], rel=1e-9)


def test_rccl_two_ranks_two_gpus(tmp_path):
	"""The RCCL path with a world larger than one: one GPU per rank, `all_gather_into_tensor` straight on the libslq
	device buffers, ragged column shards. Needs two GPUs: skipped on the one-GPU box (the multi-rank RCCL path is
	otherwise exercised only by the driver's multi-GPU bench; DESIGN.md §7 says so)."""
	import torch

	if torch.cuda.device_count() < 2:
		pytest.skip("needs two GPUs")
	from primate_amd.operators import MatrixFunction
	from primate_amd.trace import xtrace

	L = laplacian_2d(40)
	M = MatrixFunction(L, fun="exp", deg=20, orth=3, t=-0.5)
	single = xtrace(M, batch=20, pdf="sphere", seed=7, count=50, device_rng=True)
	res = _launch(2, "nccl", tmp_path / "n2", 29535, share_gpu0=False)
	assert all(r["gather_ok"] and r["nit"] == 50 for r in res)
	assert len({r["estimate"] for r in res}) == 1 and res[0]["estimate"] == pytest.approx(single, rel=1e-9)
	assert len({r["estimate_rows"] for r in res}) == 1 and res[0]["estimate_rows"] == pytest.approx(single, rel=1e-9)
