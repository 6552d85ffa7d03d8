"""First-light check on the GPU box: parity against golden vectors + a first timing of C2."""
import sys, time, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d
from primate_amd.engine import Context, DeviceOperator, LanczosPlan, quad_batch

G = np.load(ROOT / "tests/golden/slq_golden.npz")
L = laplacian_2d(int(G["lap_m"])); V = G["lap_probes"]
op = DeviceOperator(L)
for orth in [0, 3, 20]:
    for fun, key, kw in [("log", "log", {}), ("exp", "exp", {}), ("exp", "exp_t", {"t": -0.1}), ("smoothstep", "smoothstep", {"a": .5, "b": 6.}), ("numrank", "numrank", {})]:
        q = quad_batch(op, V, 20, orth, fun=fun, **kw)
        print("lap", orth, key, float(np.max(np.abs(q / G[f"lap_quad_{key}_o{orth}"] - 1))))
    plan = LanczosPlan(op, V.shape[1], 20, orth)
    plan.set_probes(V); plan.run()
    a, b, s = plan.tridiag()
    print("  alpha err", float(np.max(np.abs(a[:, :20] - G[f"lap_alpha_o{orth}"]))), "beta err", float(np.max(np.abs(b[:, :20] - G[f"lap_beta_o{orth}"]))), "steps", s[:3])
    q, nd, wt = plan.quadrature("log", return_rule=True)
    print("  nodes err", float(np.max(np.abs(nd - G[f"lap_nodes_o{orth}"]))), "weights err", float(np.max(np.abs(wt - G[f"lap_weights_o{orth}"]))))

# dense KAT full reorth
A = G["kat_A"]; v0 = G["kat_v0"]
opd = DeviceOperator(A)
plan = LanczosPlan(opd, 1, 50, 50, keep_basis=True)
plan.set_probes(v0); plan.run()
a, b, s = plan.tridiag()
print("kat alpha", float(np.max(np.abs(a[0, :50] - G["kat_alpha_o50_c50"]) / np.abs(a).max())), "beta", float(np.max(np.abs(b[0, :50] - G["kat_beta_o50_c50"]) / np.abs(b).max())), s)
Q = plan.basis(0)
print("kat |Q^T Qref| - I", float(np.max(np.abs(np.abs(Q.T @ G["kat_Q_o50_c50"]) - np.eye(50)))))
# early stop
ops = DeviceOperator(G["stop_A"])
plan = LanczosPlan(ops, 1, 20, 20); plan.set_probes(G["stop_v"]); plan.run()
a, b, s = plan.tridiag(); print("stop steps", s, b[0, :7], G["stop_beta"][:7])

# C2 anchor + timing
L2 = laplacian_2d(1000)
t0 = time.time(); op2 = DeviceOperator(L2); print("upload s", time.time() - t0)
rng = np.random.default_rng(1234)
v = np.floor(rng.random((L2.shape[0], 1)) * 2) * 2 - 1
for orth in [0, 3]:
    q = quad_batch(op2, v, 30, orth, fun="log")
    print("c2 anchor orth", orth, q[0], G["c2_quad_log_seed1234_o0_o3"], float(q[0] / G["c2_quad_log_seed1234_o0_o3"][min(orth, 1) if orth == 0 else 1] - 1))
res = {}
for orth in [0, 3, 30]:
    P = 256
    plan = LanczosPlan(op2, P, 30, orth)
    print("orth", orth, "workspace GB", plan.workspace_bytes / 1e9)
    for it in range(3):
        plan.generate_probes("rademacher", seed=1234)
        if it == 2: plan.profile_enable(True)
        op2.ctx.synchronize(); t0 = time.time()
        plan.run(); q = plan.quadrature("log")
        dt = time.time() - t0
    prof = plan.profile_read()
    print(f"orth={orth} run+quad {dt*1e3:.1f} ms  probe-matvec/s {P*30/dt:.0f}  est {np.mean(q):.4f}")
    print("   ", {k: (round(v['ms'], 3), v['launches']) for k, v in prof.items()})
    res[orth] = {"ms": dt * 1e3, "prof": prof, "est": float(np.mean(q))}
    plan.close()
Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
json.dump(res, open(ROOT / "gpurun_out/first_light.json", "w"), indent=1)
