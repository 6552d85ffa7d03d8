import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d
from oracle import oracle
from primate_amd.lanczos import _native_lanczos
from primate_amd.operators import MatrixFunction
from primate_amd.integrate import quadrature
G = np.load(ROOT / "tests/golden/slq_golden.npz")
L, V = laplacian_2d(int(G["lap_m"])), G["lap_probes"]
M = MatrixFunction(L, fun="log", deg=20, orth=3, stale_ring=True)
print("compat     ", M.quad(V[:, :3].copy()))
print("golden stale", G["mf_quad_log_o3"][:3])
print("golden clean", G["lap_quad_log_o3"][:3])
print("oracle stale", oracle.quad_batch(L, V[:, :3], 20, 3, fun="log", fresh_q=False))
n = L.shape[0]
al, be, Q = np.zeros(21), np.zeros(21), np.zeros((n, 20), order="F")
for j in range(3):
    _native_lanczos(M._op, V[:, j], 20, 1e-8, 3, al, be, Q)
    nd, wt = quadrature(al[:20], be[:20], deg=20)
    nd2, wt2 = oracle.np_quadrature(al[:20], be[:20])
    print(j, np.sum(np.log(nd) * wt) * np.linalg.norm(V[:, j]) ** 2, np.sum(np.log(nd2) * wt2) * np.linalg.norm(V[:, j]) ** 2, "Mnodes", M._nodes[:2])
