"""Estimator drivers on plain matrices (host path, no GPU): exact agreement with vectors captured
from the reference's own hutchpp / xtrace / diag / KneeCriterion ("pure" golden)."""

from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def gd():
	return np.load(ROOT / "tests" / "golden" / "slq_golden_drivers.npz")


def test_hutchpp_matches_reference(gd):
	from primate_amd.estimators import EstimatorResult
	from primate_amd.trace import hutchpp

	A = gd["A"]
	assert hutchpp(A, m=30, seed=1234) == pytest.approx(float(gd["hutchpp_m30"]), rel=1e-13)
	assert hutchpp(A, m=30, seed=1234, mode="full") == pytest.approx(float(gd["hutchpp_m30_full"]), rel=1e-13)
	est, res = hutchpp(A, m=20, seed=5, full=True, pdf="normal")
	assert isinstance(res, EstimatorResult) and res.nit == 2 * 22  # nb += nb % 3 (trace.py:151): 20 -> 22
	assert est == pytest.approx(float(gd["hutchpp_m20_normal"]), rel=1e-13)
	np.testing.assert_allclose(res.samples, gd["hutchpp_m20_samples"], rtol=1e-12)
	assert abs(hutchpp(A, m=60, seed=1) - A.trace()) < 1e-8  # m = n: the sketch is exact


def test_xtrace_matches_reference_including_its_quirks(gd):
	from primate_amd.trace import xtrace

	A = gd["A"]
	for pdf in ["sphere", "rademacher", "normal"]:
		for nb in [7, 20]:
			assert xtrace(A, pdf=pdf, batch=nb, seed=1234) == pytest.approx(float(gd[f"xtrace_{pdf}_b{nb}"]), rel=1e-12)
	est, info = xtrace(A, batch=16, seed=99, full=True)
	assert est == pytest.approx(float(gd["xtrace_full_b16"]), rel=1e-12) and info.nit == A.shape[0]
	assert abs(xtrace(A, batch=13, seed=3) - A.trace()) < 0.02 * A.trace()
	## explicit probe budget (the reference cannot do this, trace.py:271-275)
	ests = []
	est, info = xtrace(A, batch=8, seed=3, count=24, full=True, callback=lambda r: ests.append(r.nit))
	assert info.nit == 24 and ests == [8, 16, 24] and abs(est - A.trace()) < 0.15 * A.trace()


def test_diag_matches_reference(gd):
	from primate_amd.diagonal import diag

	A = gd["A"]
	np.testing.assert_allclose(diag(A, converge="count", count=50, seed=1234), gd["diag_c50"], rtol=1e-13)
	d, info = diag(A, converge="count", count=20, seed=7, full=True, pdf="normal")
	np.testing.assert_allclose(d, gd["diag_c20_normal_full"], rtol=1e-13)
	assert info.nit == 20 and info.criterion(info.estimator)
	np.testing.assert_allclose(diag(A, converge="tolerance", atol=0.0, rtol=0.01, seed=3), gd["diag_tol"], rtol=1e-13)


def test_xdiag_matches_reference(gd):
	"""The exchangeable diagonal estimator (src/primate/diagonal.py:99-138), re-derived as an average of leave-one-out estimates: the reference's output for
	the same seed, the budget rounding (odd m rounds up, m is capped at 2 n), and - as an estimator - far closer to the diagonal than plain Girard-Hutchinson at
	the same number of products when the spectrum decays."""
	from primate_amd.diagonal import diag, xdiag

	A = gd["A"]
	n = A.shape[0]
	np.testing.assert_allclose(xdiag(A, m=40, seed=1234), gd["xdiag_m40"], rtol=1e-12, atol=1e-13)
	np.testing.assert_array_equal(xdiag(A, m=39, seed=5), xdiag(A, m=40, seed=5))
	np.testing.assert_array_equal(xdiag(A, m=10 * n, seed=5), xdiag(A, seed=5))
	for pdf in ("rademacher", "normal"):
		assert xdiag(A, m=40, pdf=pdf, seed=2).shape == (n,)
	## where the estimator earns its products: a spectrum that decays (the sketch captures most of the matrix, the probes only see the rest)
	rng = np.random.default_rng(0)
	U, _ = np.linalg.qr(rng.standard_normal((n, n)))
	B = (U * 0.7 ** np.arange(n)) @ U.T
	err_x = np.linalg.norm(xdiag(B, m=40, seed=8) - B.diagonal())
	err_g = np.linalg.norm(diag(B, converge="count", count=40, seed=8) - B.diagonal())
	assert err_x < 0.05 * err_g, (err_x, err_g)


def test_knee_criterion_and_update_trinv(gd):
	from primate_amd.estimators import KneeCriterion, MeanEstimator, convergence_criterion
	from primate_amd.linalg import update_trinv

	kc, est, dec = convergence_criterion("knee", S=1.0), MeanEstimator(record=True), []
	assert isinstance(kc, KneeCriterion)
	for x in gd["knee_samples"]:
		est.update(x)
		dec.append(bool(kc(est)))
	assert np.array_equal(dec, gd["knee_decisions"])
	rng = np.random.default_rng(0)
	R = np.triu(rng.standard_normal((6, 6))) + 3 * np.eye(6)
	Rinv = np.zeros((0, 0))
	for j in range(6):
		Rinv = update_trinv(Rinv, R[: j + 1, j])
	np.testing.assert_allclose(Rinv, np.linalg.inv(R), atol=1e-12)


def test_helper_modules_match_reference_vectors(gd):
	"""Host-side helpers that ship with the reference's driver modules (random.symmetric/haar/Isotropic,
	stats.confidence_interval, estimators.ControlVariableEstimator, typing helpers) against vectors captured
	from the reference itself (tests/golden/make_golden_drivers.py)."""
	from primate_amd.estimators import ControlVariableEstimator, arr_summary
	from primate_amd.random import Isotropic, haar, symmetric
	from primate_amd.stats import Covariance, Mean, confidence_interval
	from primate_amd.typing import restrict_kwargs, setdiff_kwargs

	np.testing.assert_array_equal(symmetric(12, seed=3), gd["sym_n12_s3"])
	np.testing.assert_array_equal(symmetric(9, dist="uniform", pd=True, seed=9), gd["sym_n9_uniform_pd_s9"])
	assert np.all(np.linalg.eigvalsh(gd["sym_n9_uniform_pd_s9"]) > 0)
	np.testing.assert_allclose(np.linalg.eigvalsh(symmetric(15, ew=np.linspace(1, 2, 15), seed=1)), np.linspace(1, 2, 15), rtol=1e-12)
	np.testing.assert_array_equal(haar(8, seed=4), gd["haar_n8_s4"])
	with pytest.raises(ValueError):
		symmetric(5, dist="cauchy")
	for th in (1, 3):
		for pdf in ("signs", "sphere"):
			iso = Isotropic((20, 5), pdf=pdf, seed=11, threads=th)
			iso.fill()
			iso.fill()
			np.testing.assert_array_equal(iso.values, gd[f"iso_{pdf}_t{th}"])
			assert iso.values.flags["F_CONTIGUOUS"]
	np.testing.assert_allclose(confidence_interval(gd["ci_samples"]), gd["ci_t95"], rtol=1e-13)
	np.testing.assert_allclose(confidence_interval(gd["ci_samples"], 0.9, "normal"), gd["ci_n90"], rtol=1e-13)
	with pytest.raises(ValueError):
		confidence_interval(gd["ci_samples"], sdist="cauchy")
	cv = ControlVariableEstimator(15 / 16)
	cv.update(gd["cv_samples"][:100])
	cv.update(gd["cv_samples"][100:])
	assert cv.estimate == pytest.approx(float(gd["cv_estimate"]), rel=1e-12) and len(cv) == 250
	np.testing.assert_allclose(cv.alpha, gd["cv_alpha"], rtol=1e-10)
	assert abs(cv.estimate - 1339 / 1440) < abs(gd["cv_samples"][:, 0].mean() - 1339 / 1440)  # the variance reduction at work
	assert Mean is not None and Covariance is not None
	f = lambda a, b=1: None  # noqa: E731
	assert restrict_kwargs(f, {"a": 1, "c": 2}) == {"a": 1} and setdiff_kwargs(f, {"a": 1, "c": 2}) == {"c": 2}
	assert arr_summary(None) == "None" and arr_summary(1.0) == "1.000"
