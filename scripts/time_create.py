"""What creating an operator costs on the host (reordering, tile clustering, uploads), for the operators of BASELINE.json's
configs: the grids take the tiles, the random graph and the scattered band are turned away by the 256-cluster sample.

    SLQ_DEBUG=1 python scripts/time_create.py        (SLQ_DEVICE_BUILD=0: everything on the host, as before r04)
"""
import sys, time
from pathlib import Path
import numpy as np, scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d, laplacian_3d
from primate_amd import engine as eng

def random_graph(n=500000, deg=16, seed=1234):
	rng = np.random.default_rng(seed)
	m = int(n * deg / 2)
	i, j = rng.integers(0, n, m), rng.integers(0, n, m)
	keep = i != j
	W = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(n, n)).tocsr()
	W = ((W + W.T) > 0).astype(np.float64).tocsr()
	W.sort_indices()
	return W

def scattered_band(n=2_000_000, seed=1234):  # configs[4]'s operator at a fifth of its size
	rng = np.random.default_rng(seed)
	offs = np.unique(np.concatenate([[1, 2, 3], rng.integers(4, 2000, 4)]))[:7]
	S = sp.diags([rng.uniform(-1, 0, n - o) for o in offs], offs, shape=(n, n))
	S = (S + S.T).tocsr()
	A = (S + sp.diags(np.asarray(abs(S).sum(axis=1)).ravel() * 1.1)).tocsr()
	A.sort_indices()
	return A

ctx = eng.default_context()
for name, A in [("lap2d_1000", laplacian_2d(1000)), ("lap3d_100", laplacian_3d(100)), ("lap3d_126_f32", laplacian_3d(126, np.float32)),
                ("random graph n=5e5 deg 16", random_graph()), ("scattered band n=2e6, 15 per row", scattered_band())]:
	t = time.perf_counter(); M = sp.csr_matrix(A); M.has_sorted_indices; [np.ascontiguousarray(x) for x in (M.indptr, M.indices, M.data)]; prep = time.perf_counter() - t
	t = time.perf_counter(); op = eng.DeviceOperator(A); ctx.synchronize(); dt = time.perf_counter() - t
	t = time.perf_counter(); plan = eng.LanczosPlan(op, 256, 30, 3); ctx.synchronize(); dp = time.perf_counter() - t
	print(f"{name}: operator create {dt:.3f} s (host-side checks of the scipy matrix alone: {prep*1e3:.1f} ms), plan create {dp:.3f} s, tiles {plan.describe()['tiles']}", flush=True)
	plan.close()
	## narrow panels: the first plan of 64 probes has the 2-merged tiles' streams built (r04: on the device, from the operator's own arrays)
	t = time.perf_counter(); plan = eng.LanczosPlan(op, 64, 30, 3); ctx.synchronize(); dp = time.perf_counter() - t
	print(f"    first plan of 64 probes: {dp:.3f} s, tiles {plan.describe()['tiles']}", flush=True)
	plan.close(); op.close()
	## what the drivers pay: MatrixFunction over the same matrix twice (r04: operators are cached by content of the sparse matrix)
	from primate_amd.operators import MatrixFunction
	t = time.perf_counter(); M1 = MatrixFunction(A, fun="log", deg=30); ctx.synchronize(); d1 = time.perf_counter() - t
	t = time.perf_counter(); M2 = MatrixFunction(sp.csr_matrix(A.copy()), fun="log", deg=30); ctx.synchronize(); d2 = time.perf_counter() - t
	print(f"    MatrixFunction(A): first {d1:.3f} s, again over an equal matrix {d2:.3f} s (same operator: {M1._op is M2._op})", flush=True)
	del M1, M2
