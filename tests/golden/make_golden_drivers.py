"""Second golden fixture: the reference's estimator DRIVERS (diag, hutchpp, xtrace, fttr rule,
Knee criterion), run in the dev container exactly as tests/golden/make_golden.py describes
(synthetic `primate` package over /root/reference/src/primate; "pure" = reference code only,
"injected" = reference drivers over oracle.lanczos standing in for primate._lanczos).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_drivers.py
"""

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.dont_write_bytecode = True
from make_golden import import_reference, laplacian_2d  # noqa: E402


def main():
	import_reference()
	from primate.diagonal import diag, xdiag
	from primate.estimators import KneeCriterion, MeanEstimator
	from primate.integrate import quadrature
	from primate.lanczos import _lanczos_recurrence
	from primate.operators import MatrixFunction
	from primate.trace import hutchpp, xtrace

	out, prov = {}, {}
	rng = np.random.default_rng(1234)
	n = 60
	B = rng.standard_normal((n, n))
	A = B @ B.T / n + np.eye(n) * 0.5
	A = (A + A.T) / 2
	out["A"] = A

	## diag on a plain matrix: pure reference (no Lanczos)
	out["diag_c50"] = diag(A, converge="count", count=50, seed=1234)
	d, info = diag(A, converge="count", count=20, seed=7, full=True, pdf="normal")
	out["diag_c20_normal_full"] = d
	out["diag_tol"] = diag(A, converge="tolerance", atol=0.0, rtol=0.01, seed=3)
	prov["diag_*"] = "pure"

	## hutch++ on a plain matrix: pure
	out["hutchpp_m30"] = np.float64(hutchpp(A, m=30, seed=1234))
	out["hutchpp_m30_full"] = np.float64(hutchpp(A, m=30, seed=1234, mode="full"))
	est, res = hutchpp(A, m=20, seed=5, full=True, pdf="normal")
	out["hutchpp_m20_normal"], out["hutchpp_m20_samples"] = np.float64(est), np.ravel(res.samples)
	prov["hutchpp_*"] = "pure"

	## xtrace on a plain matrix: pure (always runs to n probes, trace.py:271-275)
	for pdf in ["sphere", "rademacher", "normal"]:
		for nb in [7, 20]:
			out[f"xtrace_{pdf}_b{nb}"] = np.float64(xtrace(A, pdf=pdf, batch=nb, seed=1234))
	est, info = xtrace(A, batch=16, seed=99, full=True)
	out["xtrace_full_b16"] = np.float64(est)
	prov["xtrace_*"] = "pure"
	out["xdiag_m40"] = xdiag(A, m=40, seed=1234)
	prov["xdiag_m40"] = "pure"

	## the same drivers over a MatrixFunction (injected)
	L = laplacian_2d(12)
	out["lap_m"] = np.int64(12)
	M = MatrixFunction(L, fun="exp", deg=20, orth=20, t=-0.1)
	out["mf_diag_c30"] = diag(M, converge="count", count=30, seed=1234)
	## hutchpp's default mode calls M.quad per column (trace.py:170), which with orth > 0 carries the
	## reference's stale-ring quirk: capture mode="full" (only _matvec, which clears Q)
	## and an orth = 0 operator (orth_vector never runs) so the vectors are quirk-free
	out["mf_hutchpp_m24_full"] = np.float64(hutchpp(M, m=24, seed=1234, mode="full"))
	M0 = MatrixFunction(L, fun="exp", deg=20, orth=0, t=-0.1)
	out["mf0_hutchpp_m24"] = np.float64(hutchpp(M0, m=24, seed=1234))
	out["mf_xtrace_b12"] = np.float64(xtrace(M, batch=12, seed=1234))
	prov["mf_*"] = "injected"
	ew = np.linalg.eigvalsh(L.toarray())
	out["mf_exact_trace"], out["mf_exact_diag"] = np.float64(np.sum(np.exp(-0.1 * ew))), np.diag(
		(lambda w, U: (U * np.exp(-0.1 * w)) @ U.T)(*np.linalg.eigh(L.toarray()))
	)

	## FTTR rule through quadrature(quad="fttr"): pure
	v = rng.uniform(size=L.shape[0])
	a, b, _ = _lanczos_recurrence(L, v.copy(), 15, 1e-8, 15, None, 15)
	th, w = quadrature(a, b, deg=15, quad="fttr")
	out["fttr_alpha"], out["fttr_beta"], out["fttr_nodes"], out["fttr_weights"] = a, b, th, w
	th2, w2 = quadrature(a, b, deg=15, quad="gw")
	out["fttr_gw_weights"] = w2
	prov["fttr_*"] = "pure"

	## Knee criterion decisions along a stream: pure
	xs = np.concatenate([rng.standard_normal(25) * 5 + 50, rng.standard_normal(40) * 0.05 + 50])
	kc = KneeCriterion(S=1.0)
	est = MeanEstimator(record=True)
	dec = []
	for x in xs:
		est.update(x)
		dec.append(bool(kc(est)))
	out["knee_samples"], out["knee_decisions"] = xs, np.array(dec)
	prov["knee_*"] = "pure"

	## helper modules the drivers lean on (pure reference): test-matrix generators, the threaded batch filler,
	## confidence intervals, the tridiagonal eigensolver front end, the control-variate estimator
	from primate.estimators import ControlVariableEstimator
	from primate.random import Isotropic, haar, symmetric
	from primate.stats import confidence_interval
	from primate.tridiag import eigh_tridiag, eigvalsh_tridiag

	out["sym_n12_s3"] = symmetric(12, seed=3)
	out["sym_n9_uniform_pd_s9"] = symmetric(9, dist="uniform", pd=True, seed=9)
	out["haar_n8_s4"] = haar(8, seed=4)
	for th in (1, 3):
		for pdf in ("signs", "sphere"):
			iso = Isotropic((20, 5), pdf=pdf, seed=11, threads=th)
			iso.fill()
			iso.fill()
			out[f"iso_{pdf}_t{th}"] = iso.values.copy()
	xs_ci = np.random.default_rng(0).normal(size=30)
	out["ci_samples"] = xs_ci
	out["ci_t95"], out["ci_n90"] = np.array(confidence_interval(xs_ci)), np.array(confidence_interval(xs_ci, 0.9, "normal"))
	rng_t = np.random.default_rng(5)
	td, te = rng_t.uniform(1, 3, 24), np.r_[0.0, rng_t.uniform(0.2, 1.0, 23)]
	tw, tZ = eigh_tridiag(td, te)
	out["tri_d"], out["tri_e"], out["tri_w"], out["tri_absZ"] = td, te, tw, np.abs(tZ)
	out["tri_w_only"] = eigvalsh_tridiag(td, te[1:])
	rng_cv = np.random.default_rng(1235)
	U = rng_cv.uniform(size=(250, 5)) * np.array([1, 2, 3, 1, 2])
	y = np.min(np.c_[U[:, 0] + U[:, 3], U[:, 0] + U[:, 2] + U[:, 4], U[:, 1] + U[:, 2] + U[:, 3], U[:, 1] + U[:, 4]], axis=1)
	ycv = np.minimum(U[:, 0] + U[:, 3], U[:, 1] + U[:, 4])
	cve = ControlVariableEstimator(15 / 16)
	cve.update(np.c_[y[:100], ycv[:100]])
	cve.update(np.c_[y[100:], ycv[100:]])
	out["cv_samples"], out["cv_estimate"], out["cv_alpha"] = np.c_[y, ycv], np.float64(cve.estimate), np.atleast_1d(cve.alpha)
	prov["sym_*|haar_*|iso_*|ci_*|tri_*|cv_*"] = "pure"

	out["provenance"] = np.array([f"{k}={v}" for k, v in sorted(prov.items())])
	np.savez_compressed(HERE / "slq_golden_drivers.npz", **out)
	print("wrote", HERE / "slq_golden_drivers.npz", (HERE / "slq_golden_drivers.npz").stat().st_size, "bytes")
	for k in ["hutchpp_m30", "xtrace_sphere_b7", "mf_hutchpp_m24_full", "mf0_hutchpp_m24", "mf_xtrace_b12", "mf_exact_trace"]:
		print(k, out[k])
	print("mf diag err", np.linalg.norm(out["mf_diag_c30"] - out["mf_exact_diag"]))
	print("knee decisions", out["knee_decisions"].sum(), "of", len(xs))


if __name__ == "__main__":
	main()
