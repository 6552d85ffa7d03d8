"""ctypes binding of libslq (include/slq.h) — the thin layer between the Python host API and the
hand-written HIP kernels. There is NO fallback: if the shared library is missing or no MI355X is
visible, every entry raises.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
## PRIMATE_AMD_LIBSLQ: load another build of the same library (kernel experiments, scripts/); no other effect
LIB_PATH = Path(os.environ["PRIMATE_AMD_LIBSLQ"]) if os.environ.get("PRIMATE_AMD_LIBSLQ") else _HERE / "_libslq.so"

SLQ_OK, SLQ_EINVAL, SLQ_ENOMEM, SLQ_EHIP, SLQ_ENODEV, SLQ_ECALLBACK, SLQ_ENOTCONV = 0, -1, -2, -3, -4, -5, -6
SLQ_F32, SLQ_F64 = 0, 1
FUN_NONE = -1
FUN_IDS = {
	"identity": 0, "abs": 1, "sqrt": 2, "log": 3, "inv": 4, "exp": 5, "smoothstep": 6, "step": 7, "numrank": 7,
	"softsign": 8,
}  # fmt: skip
PDF_IDS = {"rademacher": 0, "signs": 0, "normal": 1, "gaussian": 1, "sphere": 2}
KERNEL_CLASSES = ["spmm_3term", "axpy_norm", "reorth_dot", "reorth_update", "finalize", "probes", "quadrature", "fun_combine"]

MATVEC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


class SlqProfile(C.Structure):
	_fields_ = [("ms", C.c_double * len(KERNEL_CLASSES)), ("launches", C.c_int64 * len(KERNEL_CLASSES))]


class SlqError(RuntimeError):
	def __init__(self, code: int, msg: str):
		super().__init__(f"libslq error {code}: {msg}")
		self.code = code


_lib = None

_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_SIGNATURES = {
	"slq_last_error": (C.c_char_p, []),
	"slq_version": (C.c_int, []),
	"slq_device_count": (C.c_int, [C.POINTER(C.c_int)]),
	"slq_context_create": (C.c_int, [C.c_int, _P, _PP]),
	"slq_context_destroy": (C.c_int, [_P]),
	"slq_context_synchronize": (C.c_int, [_P]),
	"slq_context_meminfo": (C.c_int, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
	"slq_context_device": (C.c_int, [_P, C.POINTER(C.c_int)]),
	"slq_csr_create": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, _P, _P, _P, _PP]),
	"slq_csr_create_device": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, _P, _P, _P, _PP]),
	"slq_csr_gram_create": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, _PP]),
	"slq_csr_affine_create": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, _PP]),
	"slq_operator_set_parameter": (C.c_int, [_P, C.c_double]),
	"slq_dense_create": (C.c_int, [_P, C.c_int, C.c_int64, _P, C.c_int64, _PP]),
	"slq_callback_create": (C.c_int, [_P, C.c_int, C.c_int64, MATVEC_FN, _P, _PP]),
	"slq_operator_destroy": (C.c_int, [_P]),
	"slq_operator_shape": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
	"slq_operator_matmat": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int]),
	"slq_plan_create": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _PP]),
	"slq_plan_destroy": (C.c_int, [_P]),
	"slq_plan_workspace_bytes": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
	"slq_plan_describe": (C.c_int, [_P, C.c_void_p]),
	"slq_plan_query_bytes": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
	"slq_plan_set_probes": (C.c_int, [_P, _P, C.c_int64]),
	"slq_plan_generate_probes": (C.c_int, [_P, C.c_int, C.c_uint64, C.c_uint64]),
	"slq_plan_get_probes": (C.c_int, [_P, _P, C.c_int64]),
	"slq_plan_run": (C.c_int, [_P, C.c_double]),
	"slq_plan_get_tridiag": (C.c_int, [_P, _P, _P, _P]),
	"slq_plan_quadrature": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
	"slq_plan_get_basis": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
	"slq_plan_fun_action": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64]),
	"slq_quadrature_batch": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P]),
	"slq_dmat_create": (C.c_int, [_P, C.c_int64, C.c_int, _PP]),
	"slq_dmat_destroy": (C.c_int, [_P]),
	"slq_dmat_set": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64]),
	"slq_dmat_get": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64]),
	"slq_dmat_ptr": (C.c_int, [_P, C.c_int, _PP]),
	"slq_dmat_gemm_tn": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
	"slq_dmat_gemm_nn": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, C.c_double, C.c_double]),
	"slq_plan_fun_action_dmat": (C.c_int, [_P, C.c_int, _P, _P, C.c_int]),
	"slq_plan_get_probes_dmat": (C.c_int, [_P, _P, C.c_int]),
	"slq_device_callback_create": (C.c_int, [_P, C.c_int, C.c_int64, _P, _P, _P]),
	"slq_dmat_generate": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64]),
	"slq_dmat_copy": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int]),
	"slq_dmat_copy_rows": (C.c_int, [_P, C.c_int, C.c_int64, _P, C.c_int, C.c_int64, C.c_int64, C.c_int]),
	"slq_measure_stream": (C.c_int, [_P, C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
	"slq_plan_set_probes_device": (C.c_int, [_P, _P, C.c_int64]),
	"slq_fttr_batch": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
	"slq_diag_create": (C.c_int, [_P, C.c_int64, _PP]),
	"slq_diag_destroy": (C.c_int, [_P]),
	"slq_diag_update": (C.c_int, [_P, _P, C.c_int, _P]),
	"slq_diag_get": (C.c_int, [_P, _P, _P, _P, C.POINTER(C.c_int64)]),
	"slq_plan_profile_enable": (C.c_int, [_P, C.c_int]),
	"slq_plan_sweep_columns": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
	"slq_plan_profile_read": (C.c_int, [_P, C.POINTER(SlqProfile), C.c_int]),
	"slq_quad_batch": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _P, _P, _P, _P]),
	"slq_eigh_tridiag_batch": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P]),
	"slq_fAv_batch": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _P, _P, C.c_int64]),
	"slq_lanczos_f64": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_int, _P, _P, _P, C.c_size_t]),
	"slq_lanczos_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_float, C.c_int, _P, _P, _P, C.c_size_t]),
	"slq_debug_ring_flag_status": (C.c_int, [C.c_int]),
	"slq_debug_plan_poke_ring_flag": (C.c_int, [_P, C.c_int]),
}  # fmt: skip
DEVICE_MATMAT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
class PlanInfo(C.Structure):
	"""slq_plan_info (include/slq.h)."""

	_fields_ = [("panel_width", C.c_int), ("panels", C.c_int), ("ring_slots", C.c_int), ("sequence", C.c_int), ("pipelined", C.c_int),
				("reordered", C.c_int), ("upper_alpha", C.c_int), ("far_per_row", C.c_double), ("tiles", C.c_int), ("fused_alpha", C.c_int)]  # fmt: skip


EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _preload_hip_runtime() -> None:
	"""Keep ONE HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64.so (soname
	libamdhip64.so.7, the same as /opt/rocm's) and link to it by the unversioned file name; libslq
	links to the soname. If libslq loads first, a later `import torch` pulls in a SECOND runtime and
	torch then reports "No HIP GPUs are available". Loading torch's copy first (without importing
	torch) makes libslq's NEEDED entry resolve to it, whichever order the user imports things in."""
	import importlib.util
	import sys

	if "torch" in sys.modules:
		return
	try:
		spec = importlib.util.find_spec("torch")
	except (ImportError, ValueError):
		spec = None
	if spec is None or not spec.origin:
		return
	cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
	if cand.exists():
		try:
			C.CDLL(str(cand), mode=getattr(os, "RTLD_NOW", 2) | getattr(os, "RTLD_GLOBAL", 0x100))
		except OSError:
			pass


def lib() -> C.CDLL:
	"""Load primate_amd/_libslq.so (built by __graft_entry__.build()). Fails loudly if absent."""
	global _lib
	if _lib is None:
		_preload_hip_runtime()
		if not LIB_PATH.exists():
			raise ImportError(
				f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
				"(hipcc --offload-arch=gfx950). primate_amd has no CPU fallback."
			)
		L = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
		for name, (res, args) in _SIGNATURES.items():
			fn = getattr(L, name)
			fn.restype, fn.argtypes = res, args
		_lib = L
	return _lib


def check(rc: int) -> int:
	if rc < 0:
		msg = lib().slq_last_error().decode(errors="replace")
		if rc == SLQ_EINVAL:
			## the reference raises AssertionError/ValueError for bad arguments (lanczos.py:81-106,
			## pylinop.h:24-25); both are caught by `except (AssertionError, ValueError)`
			raise ValueError(msg)
		if rc == SLQ_ENOMEM:
			raise MemoryError(msg)
		raise SlqError(rc, msg)
	return rc


def ptr(a) -> C.c_void_p:
	return None if a is None else a.ctypes.data_as(C.c_void_p)


def dtype_id(dt) -> int:
	dt = np.dtype(dt)
	if dt == np.float64:
		return SLQ_F64
	if dt == np.float32:
		return SLQ_F32
	raise AssertionError("Only 32- or 64-bit floats are supported.")
