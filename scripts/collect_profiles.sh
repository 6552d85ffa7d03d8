#!/bin/bash
# Round-end evidence run on the GPU box: bench lines, rocprofv3 kernel stats and the HBM-traffic PMC passes (each counter in its
# own pass, with --kernel-trace only, per the pool's rules), all on ONE kernel-source sha, which every record carries.
# usage: bash scripts/collect_profiles.sh <tag> [part]      -> gpurun_out/<tag>/...
#   part = lines | stats | pmc | configs | all (default): a box call has a time limit; `configs` runs LAST and refuses to write
#   <tag>_configs.json unless its kernel sha equals bench.kernel_sources_sha256() and the one pmc_summary.json was taken on.
set -eo pipefail
TAG=${1:-r04}; PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=$ROOT/bench.py
SHA=$(cd $ROOT && python3 -c "import bench; print(bench.kernel_sources_sha256())")
echo "kernel sources sha256 $SHA" | tee $OUT/kernel_sha256.txt
if [ "$PART" = "lines" ] || [ "$PART" = "all" ]; then
  echo "== bench lines"
  python3 $B > $OUT/bench_default_line.json
  python3 $B --orth 0 --no-cpu-baseline --no-extra > $OUT/bench_orth0_line.json
  python3 $B --orth 30 --no-cpu-baseline --no-extra --steps 4 > $OUT/bench_orth30_line.json
  python3 $B --orth 6 --no-cpu-baseline --no-extra > $OUT/bench_orth6_line.json
  python3 $B --workload lap3d_100 --no-cpu-baseline --no-extra > $OUT/bench_lap3d_100_line.json
  python3 $B --workload lap3d_100 --orth 0 --no-cpu-baseline --no-extra > $OUT/bench_lap3d_100_orth0_line.json
  python3 $B --workload lap3d_100 --orth 30 --no-cpu-baseline --no-extra --steps 4 > $OUT/bench_lap3d_100_orth30_line.json
  python3 $B --probes 64 --no-cpu-baseline --no-extra > $OUT/bench_p64_line.json
  python3 $B --workload lap3d_100 --probes 64 --no-cpu-baseline --no-extra > $OUT/bench_lap3d_100_p64_line.json
  python3 $B --probes 32 --no-cpu-baseline --no-extra > $OUT/bench_p32_line.json
  BENCH_NO_PROFILE=1 python3 $B --no-cpu-baseline --no-extra > $OUT/bench_default_line_no_events.json
fi
if [ "$PART" = "stats" ] || [ "$PART" = "all" ]; then
  for cfg in "lap2d_1000 3 256" "lap2d_1000 0 256" "lap2d_1000 30 256" "lap2d_1000 6 256" "lap3d_100 3 256" "lap3d_100 0 256" "lap3d_100 30 256" "lap3d_100 3 64"; do
    set -- $cfg; w=$1; o=$2; pr=$3
    name=${w}_orth${o}; [ "$pr" != "256" ] && name=${name}_p$pr
    echo "== kernel stats $name"
    rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_$name -o run -- python3 $B --workload $w --orth $o --probes $pr --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_${name}_line_under_rocprof.json
  done
  echo "== kernel stats dense (configs[0] operator: 5000^2, 64 / 128 probes)"
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_dense -o run -- python3 $ROOT/scripts/bench_dense.py > $OUT/bench_dense.log 2>&1
  find $OUT -name "*kernel_stats.csv" | while read f; do d=$(basename $(dirname $f)); case $d in stats_*) cp $f $OUT/${d#stats_}_kernel_stats.csv;; esac; done
fi
if [ "$PART" = "pmc" ] || [ "$PART" = "all" ]; then
  for cfg in "lap2d_1000 3 256" "lap2d_1000 0 256" "lap2d_1000 6 256" "lap2d_1000 30 256" "lap3d_100 3 256" "lap3d_100 0 256" "lap3d_100 30 256" "lap2d_1000 3 64" "lap3d_100 3 64"; do
    set -- $cfg; w=$1; o=$2; pr=$3
    sfx=""; [ "$pr" != "256" ] && sfx=_p$pr
    for c in FETCH_SIZE WRITE_SIZE; do
      echo "== pmc $c $w orth $o probes $pr"
      rocprofv3 --output-format csv --kernel-trace --pmc $c -d $OUT/pmc_${c}_${w}_orth$o$sfx -o run -- python3 $B --workload $w --orth $o --probes $pr --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null
    done
  done
  python3 $ROOT/scripts/summarise_pmc.py $OUT > $OUT/pmc_summary.json
fi
if [ "$PART" = "configs" ] || [ "$PART" = "all" ]; then
  echo "== configs[2]..[4] at full size (scripts/run_configs.py), then configs[3] / [4] under rocprofv3 (scripts/profile_configs.sh)"
  rm -f $ROOT/gpurun_out/${TAG}_configs.json
  RUN_TAG=$TAG python3 $ROOT/scripts/run_configs.py c3 c3x c4 > $OUT/run_configs.log 2>&1
  C5_RING32=both RUN_TAG=$TAG python3 $ROOT/scripts/run_configs.py c5 >> $OUT/run_configs.log 2>&1
  RUN_TAG=$TAG bash $ROOT/scripts/profile_configs.sh ${TAG}_configs
  python3 - "$ROOT/gpurun_out/${TAG}_configs.json" "$SHA" "$OUT/pmc_summary.json" <<'PY'
import json, sys
cfg = json.load(open(sys.argv[1]))
assert cfg["_meta"]["kernel_sha256"] == sys.argv[2], "configs record taken on other kernel sources than this checkout's"
try:
    pm = json.load(open(sys.argv[3]))
    assert pm["_meta"]["kernel_sha256"] == sys.argv[2], "pmc_summary.json taken on other kernel sources"
except FileNotFoundError:
    print("(no pmc_summary.json in this tag's directory: not compared)")
print("configs record and PMC summary are on", sys.argv[2][:12])
PY
fi
# the raw traces are large: keep summaries only
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
