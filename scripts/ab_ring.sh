#!/bin/bash
# A/B of the ring-fed tile passes' new shapes (slq_ring.hpp) against what those plans took before: narrow panels (64 and
# 32 probes; SLQ_RING_NARROW=0 = the generic passes on the tiles' row order), steps with 4..8 ring columns (orth 6;
# SLQ_RING_DEEP=0), and the wide default through k_ring_pass instead of k_csr_ring_pass (SLQ_RING_GEN=1).
#   scripts/ab_ring.sh <tag> [cases...]      cases: narrow deep gen   (default: all)
set -eo pipefail
TAG=${1:-ring_ab}; shift || true
CASES=${@:-narrow deep gen}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
AB="bash $ROOT/scripts/ab_bench.sh"
for c in $CASES; do
  case $c in
    narrow)
      for P in 64 32; do
        for mode in 0 1; do
          export SLQ_RING_NARROW=$mode
          echo "== f64, $P probes, SLQ_RING_NARROW=$mode"
          $AB ${TAG}_p${P}_n${mode} "default" "lap2d_1000:3 lap3d_100:3 lap2d_1000:0 lap3d_100:0" --no-extra --probes $P
        done
        unset SLQ_RING_NARROW
      done
      for mode in 0 1; do
        export SLQ_RING_NARROW=$mode
        echo "== f32, 128 probes (512-byte panel rows), k = 50, SLQ_RING_NARROW=$mode"
        $AB ${TAG}_f32_p128_n${mode} "default" "lap3d_126:3" --no-extra --dtype f32 --probes 128 --deg 50
      done
      unset SLQ_RING_NARROW;;
    deep)
      for mode in 0 1; do
        export SLQ_RING_DEEP=$mode
        echo "== f64, 256 probes, orth 6 / 8, SLQ_RING_DEEP=$mode"
        $AB ${TAG}_deep${mode} "default" "lap2d_1000:6 lap3d_100:6 lap2d_1000:8" --no-extra
        echo "== f64, 64 probes, orth 6, SLQ_RING_DEEP=$mode"
        $AB ${TAG}_deep${mode}_p64 "default" "lap2d_1000:6 lap3d_100:6" --no-extra --probes 64
      done
      unset SLQ_RING_DEEP;;
    gen)
      for mode in 0 1; do
        export SLQ_RING_GEN=$mode
        echo "== f64, 256 probes, SLQ_RING_GEN=$mode"
        $AB ${TAG}_gen${mode} "default" "lap2d_1000:3 lap3d_100:3 lap2d_1000:0 lap2d_1000:30" --no-extra
      done
      unset SLQ_RING_GEN;;
  esac
done
