"""Diagnostic (needs a -DSLQ_DEBUG_TIMES build of libslq): how far apart do the waves of one
XCD's row chunk run inside the merged dots pass? Every wave stamps s_memrealtime (100 MHz) at 0, 1/4, 1/2, 3/4 and the
end of its statically assigned rows; the spread between waves at each milestone is the drift of the sweep front.

    PRIMATE_AMD_LIBSLQ=primate_amd/_libslq_dbg.so python scripts/wave_drift.py [--workload lap3d_100] [--orth 3]
"""

import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from primate_amd import _capi  # noqa: E402
from primate_amd.engine import Context, DeviceOperator, LanczosPlan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="lap3d_100")
ap.add_argument("--orth", type=int, default=3)
ap.add_argument("--probes", type=int, default=256)
args = ap.parse_args()
kind, m = args.workload.split("_")
A = bench.laplacian_2d(int(m)) if kind == "lap2d" else bench.laplacian_3d(int(m))
ctx = Context(device=0)
op = DeviceOperator(A, ctx=ctx)
plan = LanczosPlan(op, args.probes, 6, args.orth)
L = _capi.lib()
nw = 4096 * 8  # waves: generous
L.slq_debug_times_begin.argtypes = [C.c_size_t]
L.slq_debug_times_read.argtypes = [C.c_void_p, C.c_size_t]
assert L.slq_debug_times_begin(nw * 64) == 0
for it in range(3):
	plan.generate_probes("rademacher", seed=1, probe_offset=0)
	plan.run(1e-8)
	plan.quadrature("log")
buf = np.zeros((nw, 8), dtype=np.uint64)
assert L.slq_debug_times_read(buf.ctypes.data, buf.nbytes) == 0
buf = buf[buf[:, 4] > 0]
t0 = buf[:, 0].min()
T = (buf[:, :5].astype(np.int64) - int(t0)) / 100.0  # microseconds
bx = (buf[:, 7] & 0xFFFFFFFF).astype(np.int64)
by = (buf[:, 7] >> 32).astype(np.int64)
print(f"{len(buf)} waves, kernel span {T[:, 4].max():.1f} us, rows per wave {int(buf[:, 6].min())}..{int(buf[:, 6].max())}")
for panel in sorted(set(by)):
	for grp in range(8):
		sel = (by == panel) & ((bx & 7) == grp)
		if not sel.any():
			continue
		t = T[sel]
		xcc = sorted(set(buf[sel, 5].astype(int)))
		msg = " | ".join(f"{np.min(t[:, k]):7.1f} {np.median(t[:, k]):7.1f} {np.max(t[:, k]):7.1f}" for k in range(5))
		print(f"panel {panel} group {grp} xcc {xcc} waves {sel.sum():4d}  (min med max us at 0, 1/4, 1/2, 3/4, end)  {msg}")
# per-iteration time of a wave, and the spread in ITERATIONS at the half-way milestone
per_it = (T[:, 4] - T[:, 0]) / np.maximum(buf[:, 6].astype(float), 1)
print(f"time per row of a wave: median {np.median(per_it):.2f} us (min {per_it.min():.2f}, max {per_it.max():.2f})")
for panel in sorted(set(by)):
	sel = by == panel
	half = T[sel, 2]
	print(f"panel {panel}: spread of the half-way stamp {half.max() - half.min():.1f} us = {(half.max() - half.min()) / np.median(per_it):.1f} rows per wave")
