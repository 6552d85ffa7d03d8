import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d
from oracle import oracle
from primate_amd.lanczos import _native_lanczos
G = np.load(ROOT / "tests/golden/slq_golden.npz")
L, V = laplacian_2d(int(G["lap_m"])), G["lap_probes"]
rng = np.random.default_rng(3)
for orth, ncv in [(3, 20), (2, 20), (5, 7)]:
    Q0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((L.shape[0], ncv)))[0])
    al, be, Q = np.zeros(21), np.zeros(21), Q0.copy(order="F")
    al2, be2, Q2 = np.zeros(21), np.zeros(21), Q0.copy(order="F")
    s1 = _native_lanczos(L, V[:, 1], 20, 1e-8, orth, al, be, Q)
    s2 = oracle.lanczos(L, V[:, 1], 20, 1e-8, orth, al2, be2, Q2)
    al3, be3, Q3 = np.zeros(21), np.zeros(21), np.zeros_like(Q0)
    oracle.lanczos(L, V[:, 1], 20, 1e-8, orth, al3, be3, Q3)
    print("orth", orth, "ncv", ncv, "steps", s1, s2)
    print("  alpha gpu-oracle(stale)", np.abs(al - al2)[:6], "\n  alpha oracle stale-fresh", np.abs(al2 - al3)[:6])
    print("  beta  gpu-oracle(stale)", np.abs(be - be2)[:6], "\n  beta  oracle stale-fresh", np.abs(be2 - be3)[:6])
