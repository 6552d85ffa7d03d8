"""Where does the device xtrace of configs[2] spend its time? Wraps the device-matrix entry points and the
plan calls with synchronising timers (scripts only; the product path has no such hooks)."""
import sys, time, json
from collections import defaultdict
from pathlib import Path
import numpy as np, scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from primate_amd import engine, trace
from primate_amd.operators import MatrixFunction

acc = defaultdict(float); cnt = defaultdict(int)
ctx_holder = {}

def timed(name, fn):
	def w(*a, **k):
		c = ctx_holder.get("ctx")
		if c: c.synchronize()
		t = time.perf_counter(); r = fn(*a, **k)
		if c: c.synchronize()
		acc[name] += time.perf_counter() - t; cnt[name] += 1
		return r
	return w

for nm in ("tn", "add_product", "set", "get"):
	setattr(engine.DeviceMatrix, nm, timed("dmat." + nm, getattr(engine.DeviceMatrix, nm)))
for nm in ("run", "fun_action_into", "set_probes_device"):
	setattr(engine.LanczosPlan, nm, timed("plan." + nm, getattr(engine.LanczosPlan, nm)))
trace._leave_one_out_estimates = timed("leave_one_out_estimates(host)", trace._leave_one_out_estimates)

n, k, P = 500000, 40, 512
rng = np.random.default_rng(1234)
mm = int(n * 16 / 2)
i, j = rng.integers(0, n, mm), rng.integers(0, n, mm)
keep = i != j
W = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(n, n)).tocsr()
W = ((W + W.T) > 0).astype(np.float64).tocsr(); W.sort_indices()
M = MatrixFunction(W, fun="exp", deg=k, orth=3)
ctx_holder["ctx"] = M._op.ctx
import primate_amd.random as R
R.isotropic = timed("isotropic(host draw)", R.isotropic)
trace.isotropic = R.isotropic if hasattr(trace, "isotropic") else None
t0 = time.perf_counter()
est, info = trace.xtrace(M, batch=128, seed=1234, count=P, full=True)
tot = time.perf_counter() - t0
print(json.dumps({"total_s": tot, "estimate": float(est), "parts": {k: [round(v, 4), cnt[k]] for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}}, indent=1))
