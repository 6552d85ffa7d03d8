import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d
from oracle import oracle
from primate_amd.lanczos import _native_lanczos
from primate_amd.engine import DeviceOperator
G = np.load(ROOT / "tests/golden/slq_golden.npz")
L, V = laplacian_2d(int(G["lap_m"])), G["lap_probes"]
op = DeviceOperator(L)
n = L.shape[0]
al, be, Q = np.zeros(21), np.zeros(21), np.zeros((n, 20), order="F")
al2, be2, Q2 = np.zeros(21), np.zeros(21), np.zeros((n, 20), order="F")
for j in range(3):
    _native_lanczos(op, V[:, j], 20, 1e-8, 3, al, be, Q)
    oracle.lanczos(L, V[:, j], 20, 1e-8, 3, al2, be2, Q2)
    print("probe", j, "alpha diff", np.max(np.abs(al - al2)), "beta diff", np.max(np.abs(be - be2)), "Q diff", np.max(np.abs(Q - Q2)), "col-wise", np.max(np.abs(Q - Q2), axis=0)[[0, 1, 17, 18, 19]])
