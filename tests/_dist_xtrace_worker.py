"""Worker of tests/test_gpu_distributed.py: one rank of a column-sharded xtrace (all ranks share GPU 0 in
the rehearsal; on a real node each rank has its own GPU and the backend is nccl = RCCL)."""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
	import torch.distributed as dist

	backend, out = sys.argv[1], sys.argv[2]
	if os.environ.get("DIST_TEST_SHARE_GPU0"):
		os.environ["LOCAL_RANK"] = "0"  # rehearsal on a 1-GPU box: every rank on device 0
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	if backend == "nccl":
		## one GPU per rank, bound BEFORE any GPU call: RCCL's communicator and barrier then use this rank's own device
		import torch

		torch.cuda.set_device(local_rank)
		dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
	else:
		dist.init_process_group(backend)
	rank, world = dist.get_rank(), dist.get_world_size()
	from conftest import laplacian_2d
	from primate_amd.distributed import allgather_columns, sharded_diag_device, sharded_hutch_device, sharded_xtrace
	from primate_amd.engine import DeviceMatrix
	from primate_amd.operators import MatrixFunction

	L = laplacian_2d(40)
	M = MatrixFunction(L, fun="exp", deg=20, orth=3, t=-0.5)
	res = {}
	## ragged on purpose: 50 probes in blocks of 20 (20, 20, 10) over `world` ranks
	est, info = sharded_xtrace(M, count=50, batch=20, pdf="sphere", seed=7, full=True)
	res["estimate"], res["nit"] = float(est), int(info.nit)
	## the same estimator with the sketches row-sharded (all-to-alls + m x m all-reduces instead of all-gathers)
	est, info = sharded_xtrace(M, count=50, batch=20, pdf="sphere", seed=7, full=True, sketches="rows")
	res["estimate_rows"], res["nit_rows"] = float(est), int(info.nit)
	## probe-sharded hutch and diag on the same operator: one reduction each
	cnt, mean, var = sharded_hutch_device(M._op, 45, 20, 3, fun="exp", seed=13, t=-0.5)
	res["hutch"] = [int(cnt), float(mean), float(var)]
	## adaptive stopping, batch-synchronous over the ranks (global batches of 12 probes)
	e, info = sharded_hutch_device(M._op, None, 20, 3, fun="exp", seed=13, converge="confidence", batch=12, full=True,
								   converge_kwargs=dict(confidence=0.95, atol=0.0, rtol=0.02), t=-0.5)  # fmt: skip
	res["adaptive"] = [float(e), int(info.nit)]
	est, numer, denom, c = sharded_diag_device(M._op, 35, 20, 3, fun="exp", seed=13, batch=8, t=-0.5)
	res["diag"] = [int(c), float(np.sum(est)), float(np.sum(numer)), float(np.sum(denom))]
	## the collective itself, on known data
	n = L.shape[0]
	S, G = DeviceMatrix(n, 3, ctx=M._op.ctx), DeviceMatrix(n, 3 * world, ctx=M._op.ctx)
	S.set(0, np.arange(n * 3, dtype=np.float64).reshape(n, 3, order="F") + 1000.0 * rank)
	allgather_columns(S, 3, G)
	got = G.get()
	ok = all(np.array_equal(got[:, 3 * r : 3 * r + 3], np.arange(n * 3, dtype=np.float64).reshape(n, 3, order="F") + 1000.0 * r) for r in range(world))
	res["gather_ok"] = bool(ok)
	json.dump(res, open(f"{out}.rank{rank}.json", "w"))
	dist.barrier()
	dist.destroy_process_group()


if __name__ == "__main__":
	main()
