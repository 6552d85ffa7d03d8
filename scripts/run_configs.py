"""Full-size runs of BASELINE.json configs 3-5 on one GPU (timing + sanity), one JSON line each."""
import json, os, sys, time
from pathlib import Path
import numpy as np, scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_3d
from primate_amd import engine
from primate_amd.engine import DeviceOperator, LanczosPlan, DiagAccumulator

which = sys.argv[1:] or ["c4", "c3", "c5"]
out = {}

if "c4" in which:
    # config 4: heat-kernel signature diag(exp(-t L)), 3D 7-point Laplacian 126^3, k=50, fp32, P probes
    m, t, k, P, B = 126, 0.1, 50, int(os.environ.get("C4_PROBES", 1024)), 256
    A = laplacian_3d(m, dtype=np.float32)
    op = DeviceOperator(A)
    n = A.shape[0]
    T = (sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))).toarray()
    w, U = np.linalg.eigh(T); d1 = ((U * np.exp(-t * w)) @ U.T).diagonal()
    exact = np.einsum("i,j,k->ijk", d1, d1, d1).ravel()
    acc = DiagAccumulator(n, ctx=op.ctx)
    plan = LanczosPlan(op, B, k, 3, keep_basis=True)
    print("c4 workspace GB", plan.workspace_bytes / 1e9, flush=True)
    op.ctx.synchronize(); t0 = time.time()
    for c in range(0, P, B):
        plan.generate_probes("rademacher", seed=1234, probe_offset=c)
        plan.run(); acc.update(plan, "exp", t=-t)
    numer, denom, rmean, cnt = acc.get()
    dt = time.time() - t0
    est = numer / denom
    out["c4"] = dict(n=n, nnz=int(A.nnz), k=k, probes=P, dtype="f32", seconds=dt, probe_matvecs_per_s=P * k / dt,
                     rel_l2_err=float(np.linalg.norm(est - exact) / np.linalg.norm(exact)), max_abs_err=float(np.max(np.abs(est - exact))),
                     expected_stat_err=float(1 / np.sqrt(P)))
    print(json.dumps({"c4": out["c4"]}), flush=True)
    plan.close(); acc.close(); op.close()

if "c3" in which:
    # config 3: Estrada index tr(exp(A)) of a G(n, 16/n) graph, n = 5e5, k = 40, 512 probes
    n, k, P = 500000, 40, int(os.environ.get("C3_PROBES", 512))
    rng = np.random.default_rng(1234)
    mm = int(n * 16 / 2)
    i, j = rng.integers(0, n, mm), rng.integers(0, n, mm)
    keep = i != j
    W = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(n, n)).tocsr()
    W = ((W + W.T) > 0).astype(np.float64).tocsr(); W.sort_indices()
    op = DeviceOperator(W)
    res = {}
    for orth in [0, 3]:
        plan = LanczosPlan(op, 256, k, orth)
        qs = []
        op.ctx.synchronize(); t0 = time.time()
        for c in range(0, P, 256):
            plan.generate_probes("rademacher", seed=1234, probe_offset=c)
            plan.run(); qs.append(plan.quadrature("exp"))
        dt = time.time() - t0
        q = np.concatenate(qs)
        res[f"hutch_orth{orth}"] = dict(seconds=dt, probe_matvecs_per_s=P * k / dt, estimate=float(q.mean()), stderr=float(q.std(ddof=1) / np.sqrt(P)))
        plan.close()
    # f(A) Omega products as xtrace needs them: batched action with the basis kept
    plan = LanczosPlan(op, 128, k, 3, keep_basis=True)
    print("c3 action workspace GB", plan.workspace_bytes / 1e9, flush=True)
    op.ctx.synchronize(); t0 = time.time()
    plan.generate_probes("sphere", seed=1234)
    plan.run(); Y = plan.fun_action("exp")
    dt = time.time() - t0
    res["fun_action_128"] = dict(seconds=dt, probe_matvecs_per_s=128 * k / dt, ynorm=float(np.linalg.norm(Y)))
    out["c3"] = dict(n=n, nnz=int(W.nnz), k=k, probes=P, **res)
    print(json.dumps({"c3": out["c3"]}), flush=True)
    plan.close(); op.close()

if "c5" in which:
    # config 5 (one GPU's share, reduced): n = 1e7, 15 nnz/row banded-random SPD, k = 80, full reorth, step f
    n, k, P = 10_000_000, 80, int(os.environ.get("C5_PROBES", 32))
    rng = np.random.default_rng(1234)
    offs = np.unique(np.concatenate([[1, 2, 3], rng.integers(4, 2000, 4)]))[:7]
    diags = [rng.uniform(-1, 0, n - o) for o in offs]
    U = sp.diags(diags, offs, shape=(n, n))
    S = (U + U.T).tocsr()
    d = np.asarray(abs(S).sum(axis=1)).ravel() * rng.uniform(0.2, 1.2, n)  # not all rows dominant: indefinite tail
    A = (S + sp.diags(d)).tocsr(); A.sort_indices()
    op = DeviceOperator(A)
    plan = LanczosPlan(op, P, k, k)
    print("c5 workspace GB", plan.workspace_bytes / 1e9, "nnz", A.nnz, flush=True)
    op.ctx.synchronize(); t0 = time.time()
    plan.generate_probes("rademacher", seed=1234)
    plan.run(); q = plan.quadrature("numrank", threshold=1e-6)
    dt = time.time() - t0
    out["c5"] = dict(n=n, nnz=int(A.nnz), k=k, probes=P, orth=k, seconds=dt, probe_matvecs_per_s=P * k / dt, eigencount=float(q.mean()), stderr=float(q.std(ddof=1) / np.sqrt(P)))
    print(json.dumps({"c5": out["c5"]}), flush=True)

if "c3x" in which:
    # config 3 as written: xtrace of exp(A), G(n, 16/n), n = 5e5, k = 40, 512 probes in batches of 128
    from primate_amd.operators import MatrixFunction
    from primate_amd.trace import xtrace, hutch
    n, k, P = 500000, 40, int(os.environ.get("C3_PROBES", 512))
    rng = np.random.default_rng(1234)
    mm = int(n * 16 / 2)
    i, j = rng.integers(0, n, mm), rng.integers(0, n, mm)
    keep = i != j
    W = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(n, n)).tocsr()
    W = ((W + W.T) > 0).astype(np.float64).tocsr(); W.sort_indices()
    M = MatrixFunction(W, fun="exp", deg=k, orth=3)
    t0 = time.time(); marks = []
    est, info = xtrace(M, batch=128, seed=1234, count=P, full=True, device_rng=bool(int(os.environ.get("C3_DEVICE_RNG", "1"))), callback=lambda r: marks.append((r.nit, float(r.estimate), time.time() - t0)))
    dt = time.time() - t0
    t0 = time.time(); h = hutch(M, converge="count", count=P, seed=1234); dth = time.time() - t0
    t0 = time.time(); hd = hutch(M, pdf="device:rademacher", converge="count", count=P, seed=1234); dthd = time.time() - t0
    out["c3x"] = dict(n=n, nnz=int(W.nnz), k=k, probes=P, xtrace_seconds=dt, xtrace_estimate=float(est), progress=marks, hutch_seconds=dth, hutch_estimate=float(h), hutch_device_rng_seconds=dthd, hutch_device_rng_estimate=float(hd))
    print(json.dumps({"c3x": out["c3x"]}), flush=True)

(ROOT / "gpurun_out").mkdir(exist_ok=True)
json.dump(out, open(ROOT / "gpurun_out" / ("configs_" + "_".join(which) + ".json"), "w"), indent=1)
