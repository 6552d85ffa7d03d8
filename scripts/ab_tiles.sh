#!/bin/bash
# A/B of the tile modes (SLQ_TILES=0 against the default) over the operators and panel shapes that matter, and what
# building the tiles costs when an operator is created: scripts/ab_tiles.sh <tag>
set -eo pipefail
TAG=${1:-tiles_ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for mode in 0 default; do
  if [ "$mode" = "default" ]; then unset SLQ_TILES; else export SLQ_TILES=$mode; fi
  echo "== SLQ_TILES=$mode: f64, 256 probes"
  bash $ROOT/scripts/ab_bench.sh ${TAG}_t${mode} "default" "lap2d_1000:3 lap2d_1000:0 lap2d_1000:6 lap3d_100:3 lap3d_100:0 lap3d_100:6" --no-extra
  echo "== SLQ_TILES=$mode: f32, 512 probes, k = 50"
  bash $ROOT/scripts/ab_bench.sh ${TAG}_t${mode}_f32 "default" "lap2d_1400:3 lap3d_126:3" --no-extra --dtype f32 --probes 512 --deg 50
  echo "== SLQ_TILES=$mode: narrow panels, 64 and 32 probes (merged tiles through k_ring_pass since r03; scripts/ab_ring.sh switches the new forms one by one)"
  bash $ROOT/scripts/ab_bench.sh ${TAG}_t${mode}_p64 "default" "lap2d_1000:3 lap3d_100:3 lap2d_1000:6 lap3d_100:6" --no-extra --probes 64
  bash $ROOT/scripts/ab_bench.sh ${TAG}_t${mode}_p32 "default" "lap2d_1000:3 lap3d_100:3" --no-extra --probes 32
  python3 - <<'PY'
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd())); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "tests"))
from conftest import laplacian_2d, laplacian_3d
from primate_amd.engine import DeviceOperator
for name, A in (("lap2d_1000", laplacian_2d(1000)), ("lap3d_100", laplacian_3d(100))):
    DeviceOperator(laplacian_2d(100)).close()
    t0 = time.time(); op = DeviceOperator(A); dt = time.time() - t0
    print(f"operator creation {name}: {dt:.2f} s", flush=True)
    op.close()
PY
done
