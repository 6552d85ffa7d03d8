import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from primate_amd.engine import DeviceOperator, LanczosPlan
rng = np.random.default_rng(1234)
import os
dt = np.float32 if os.environ.get("DENSE_DTYPE", "f64") == "f32" else np.float64  # DENSE_DTYPE=f32: k_dense_mfma32_lds (SLQ_DENSE_MFMA=0: the VALU kernel)
for n, P in [(5000, 64), (5000, 128), (5000, 256), (2000, 64), (5001, 64)]:
    B = rng.standard_normal((n, n)); A = (B @ B.T / n + np.eye(n)).astype(dt)
    op = DeviceOperator(A)
    for orth in [0, 3]:
        plan = LanczosPlan(op, P, 20, orth)
        ts = []
        for it in range(4):
            plan.generate_probes("rademacher", seed=1)
            if it == 1: plan.profile_enable(True); plan.profile_read()
            op.ctx.synchronize(); t0 = time.time(); plan.run(); q = plan.quadrature("identity"); ts.append(time.time() - t0)
        prof = plan.profile_read()
        ks = {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in prof.items() if v["launches"]}
        print(f"dense {np.dtype(dt).name} n={n} P={P} orth={orth}: {min(ts[1:])*1e3:.2f} ms per 20-step run; est {q.mean():.4f} vs trace {np.trace(A):.4f}; {ks}", flush=True)
        plan.close()
    op.close()
