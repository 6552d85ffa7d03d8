"""Diagnostic (needs a -DSLQ_DEBUG_TIMES build of libslq): the time line of ONE workgroup of the ring-fed merged dots pass
(k_csr_ring_pass, workgroup 0 of panel 0). Loader 0 stamps, per tile k: start of its iteration, end of the counted wait
(tile k - lag has landed), end of the wait for the slot, end of its DMA issue; consumer 0 stamps: start of its poll,
tile seen ready, row computed and slot released. s_memrealtime ticks are 10 ns.

    PRIMATE_AMD_LIBSLQ=build_ab/ring_dbg.so python scripts/ring_timeline.py [--workload lap2d_1000] [--orth 3]
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
ap.add_argument("--workload", default="lap2d_1000")
ap.add_argument("--orth", type=int, default=3)
ap.add_argument("--probes", type=int, default=256)
args = ap.parse_args()
kind, m = args.workload.split("_")
A = bench.laplacian_2d(int(m)) if kind == "lap2d" else bench.laplacian_3d(int(m))
ctx = Context(device=0)
op = DeviceOperator(A, ctx=ctx)
plan = LanczosPlan(op, args.probes, 6, args.orth)
print(plan.describe())
L = _capi.lib()
L.slq_debug_times_begin.argtypes = [C.c_size_t]
L.slq_debug_times_read.argtypes = [C.c_void_p, C.c_size_t]
assert L.slq_debug_times_begin(256 * 64) == 0
for it in range(2):
	plan.generate_probes("rademacher", seed=1, probe_offset=0)
	plan.run(1e-8)
	plan.quadrature("log")
buf = np.zeros((256, 8), dtype=np.uint64)
assert L.slq_debug_times_read(buf.ctypes.data, buf.nbytes) == 0
buf = buf[buf[:, 0] > 0]
t0 = int(buf[0, 0])
T = (buf[:, :7].astype(np.int64) - t0) / 100.0  # microseconds (the stamps of the LAST launch of the pass)
print(f"{len(buf)} tiles stamped; per-tile period {np.diff(T[8:, 0]).mean():.2f} us (loader iteration to iteration)")
print(" tile  L:start  landed(k-lag)  slot free   issued | C:poll   ready    released | DMAs")
for k in list(range(0, 12)) + list(range(100, 112)):
	if k < len(buf):
		print(f"{k:5d} {T[k,0]:8.2f} {T[k,1]:10.2f} {T[k,2]:12.2f} {T[k,3]:9.2f} | {T[k,4]:7.2f} {T[k,5]:8.2f} {T[k,6]:9.2f} | {int(buf[k,7])}")
S = T[16:]
print("loader per tile (us): counted wait %.2f, slot wait %.2f, issue %.2f" % ((S[:, 1] - S[:, 0]).mean(), (S[:, 2] - S[:, 1]).mean(), (S[:, 3] - S[:, 2]).mean()))
Cs = S[(np.arange(len(S)) + 16) % 2 == 0]  # consumer 0 belongs to group 0: it stamps the even tiles only (16 waves: two groups)
print("consumer 0, per tile OF ITS GROUP (us): loop top %.2f (prev release -> poll), poll %.2f, compute %.2f; group period %.2f" % (
	(Cs[1:, 4] - Cs[:-1, 6]).mean(), (Cs[:, 5] - Cs[:, 4]).mean(), (Cs[:, 6] - Cs[:, 5]).mean(), np.diff(Cs[:, 4]).mean()))
lag = 2
land = S[lag:, 1] - S[:-lag, 3]
print("issue end of tile k -> seen landed (at the loader's next-but-one iteration): %.2f us mean, %.2f min" % (land.mean(), land.min()))
print("published -> consumer sees it: %.2f us (even tiles)" % (S[:-lag, 5] - S[lag:, 1])[(np.arange(len(S) - lag) + 16) % 2 == 0].mean())
