"""Diagnostic: configs[3]'s operator (126^3 7-point Laplacian, fp32, k = 50, orth 3) - how far apart are two correct fp32
implementations of the ACTION exp(-tA)v (oracle recurrence with the whole basis kept vs the device), per launch sequence?
Prints alpha/beta divergence by step, the action's relative difference and v^T f(A) v against the quadrature.
    python scripts/check_c4_action.py [m]"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_3d  # noqa: E402
from oracle import oracle  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 126
t, k, P = 0.1, 50, 128
A = laplacian_3d(m, dtype=np.float32)
n = A.shape[0]
for env in ({}, {"SLQ_RING_NARROW": "0"}, {"SLQ_TILES": "0"}):
	for kk, vv in env.items():
		os.environ[kk] = vv
	from primate_amd import engine as eng

	op = eng.DeviceOperator(A)
	plan = eng.LanczosPlan(op, P, k, 3, keep_basis=True)
	plan.generate_probes("rademacher", seed=1234)
	V = plan.get_probes()
	plan.run()
	a, b, steps = plan.tridiag()
	Y = plan.fun_action("exp", t=-t)
	q = plan.quadrature("exp", t=-t)
	c = P - 1
	v = np.ascontiguousarray(V[:, c])
	al, be, Q = np.zeros(k + 1, dtype=np.float32), np.zeros(k + 1, dtype=np.float32), np.zeros((n, k), dtype=np.float32, order="F")
	oracle.lanczos(A, v.copy(), k, 1e-8, 3, al, be, Q)
	T = np.diag(al[:k].astype(np.float64)) + np.diag(be[1:k].astype(np.float64), 1) + np.diag(be[1:k].astype(np.float64), -1)
	th, Yv = np.linalg.eigh(T)
	ref_y = np.linalg.norm(v.astype(np.float64)) * (Q.astype(np.float64) @ (Yv @ (np.exp(-t * th) * Yv[0, :])))
	ref_q = oracle.quad_batch(A, v.reshape(-1, 1), k, 3, fun="exp", t=-t, fresh_q=True)[0]
	da = np.abs(a[c, :k] - al[:k]) / np.abs(al[:k]).max()
	print(env, plan.describe()["tiles"], "alpha rel diff at steps 0,5,10,20,30,40,49:", [f"{da[i]:.1e}" for i in (0, 5, 10, 20, 30, 40, 49)])
	print("   action rel diff (max-norm):", np.abs(Y[:, c] - ref_y).max() / np.abs(ref_y).max(), " v.Y:", float(v.astype(np.float64) @ Y[:, c].astype(np.float64)), " quadrature:", q[c], " oracle quad:", ref_q)
	G = Q[:, :].astype(np.float64).T @ Q[:, :].astype(np.float64)
	print("   oracle basis loss of orthogonality ||Q^T Q - I||_max:", np.abs(G - np.eye(k)).max())
	Qd = plan.basis(c).astype(np.float64)
	print("   device basis loss of orthogonality:", np.abs(Qd.T @ Qd - np.eye(k)).max())
	plan.close()
	op.close()
	for kk in env:
		del os.environ[kk]
