#!/bin/bash
# Dev-container check (needs /root/reference; nothing here travels to the GPU box): run the REFERENCE's own test
# files for the host-only modules against primate_amd, by exposing it under the package name `primate`.
# The GPU-dependent reference tests (lanczos, operator, trace, diagonal, quadrature, tridiagonal) are mirrored in
# tests/test_gpu_*.py instead, since the reference cannot run where the GPU is.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SHIM=$(mktemp -d)
mkdir -p "$SHIM/primate"
cat > "$SHIM/primate/__init__.py" <<'PY'
import importlib, sys
import primate_amd
for m in ["estimators", "stats", "random", "linalg", "special", "typing"]:
    sys.modules[f"primate.{m}"] = importlib.import_module(f"primate_amd.{m}")
from primate_amd import get_include  # noqa: F401
PY
cd "$SHIM"
T=/root/reference/tests
PYTHONDONTWRITEBYTECODE=1 PYTHONPATH="$SHIM:$ROOT" python -m pytest -p no:cacheprovider -q \
  $T/test_estimators.py $T/test_stats.py $T/test_random.py $T/test_linalg.py $T/test_special.py
rm -rf "$SHIM"
