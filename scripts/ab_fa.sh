#!/bin/bash
# A/B builds of the fused update + alpha pass (slq_ring_fa.hpp): lag in rounds, loader read-ahead, store flavour.
# Build here (no GPU needed):  bash scripts/ab_fa.sh build      -> build_ab/libslq_fa<i>.so
# Run on the GPU box:          bash scripts/ab_fa.sh run [workload] [orth]
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VARIANTS=("-DSLQ_FA_LAG=1" "-DSLQ_FA_LAG=1 -DSLQ_FA_INFLIGHT=3" "-DSLQ_FA_LAG=2 -DSLQ_FA_INFLIGHT=3" "-DSLQ_FA_LAG=1 -DSLQ_FA_STORE=1" "-DSLQ_FA_LAG=3")
if [ "$1" = "build" ]; then
  mkdir -p $ROOT/build_ab
  for i in "${!VARIANTS[@]}"; do
    python3 - "${VARIANTS[$i]}" "$ROOT/build_ab/libslq_fa$i.so" <<'PY'
import sys, __graft_entry__ as g
from pathlib import Path
g.build_libslq(extra_flags=sys.argv[1].split(), out=Path(sys.argv[2]), ring_only=True)
PY
    echo "built $i: ${VARIANTS[$i]}"
  done
  exit 0
fi
W=${2:-lap2d_1000}; O=${3:-3}
for i in "${!VARIANTS[@]}"; do
  echo "== ${VARIANTS[$i]}"
  PRIMATE_AMD_LIBSLQ=$ROOT/build_ab/libslq_fa$i.so python3 $ROOT/scripts/fa_time.py $W $O 2>&1 | grep -v amdgpu.ids | grep fused
done
