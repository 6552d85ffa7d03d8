"""Fingerprint of what an operator's creation decides (order, tiles, streams) through what it produces: sha256 of alpha, beta and the
quadrature values of one seeded 256 x 30 run per operator. Two libraries that print the same lines build the same operators
(r04: used to show that the device-side build, the stamp-based regrouping and the cached first-level Cuthill-McKee leave every result
bitwise unchanged).   PRIMATE_AMD_LIBSLQ=<other library> python scripts/op_fingerprint.py"""
import hashlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import laplacian_2d, laplacian_3d
from primate_amd import engine as eng

cases = [("lap2d_1000", laplacian_2d(1000), 256, 3), ("lap2d_1000", laplacian_2d(1000), 64, 3), ("lap3d_100", laplacian_3d(100), 256, 3), ("lap3d_100", laplacian_3d(100), 64, 6),
         ("lap3d_100", laplacian_3d(100), 32, 0), ("lap3d_80_f32", laplacian_3d(80, np.float32), 256, 3)]
ops = {}
for name, A, P, orth in cases:
	if name not in ops:
		t = time.perf_counter()
		ops[name] = eng.DeviceOperator(A)
		print(f"{name}: created in {time.perf_counter() - t:.3f} s", flush=True)
	plan = eng.LanczosPlan(ops[name], P, 30, orth)
	plan.generate_probes("rademacher", seed=77)
	plan.run()
	q = plan.quadrature("log")
	a, b, st = plan.tridiag()
	h = hashlib.sha256()
	for x in (a, b, q):
		h.update(np.ascontiguousarray(x).tobytes())
	info = plan.describe()
	print(f"{name} P={P} orth={orth} tiles={info['tiles']} seq={info['sequence']} sum={float(np.sum(q)):.12e} sha={h.hexdigest()[:16]}", flush=True)
	plan.close()
