#!/bin/bash
# Round-end evidence run on the GPU box (one gpurun call): bench lines, rocprofv3 kernel stats and the
# HBM-traffic PMC passes (each counter in its own pass, with --kernel-trace only, per the pool's rules).
# usage: bash scripts/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
set -eo pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=$ROOT/bench.py
echo "== bench lines"; 
python3 $B > $OUT/bench_default_line.json
python3 $B --orth 0 --no-cpu-baseline > $OUT/bench_orth0_line.json
python3 $B --orth 30 --no-cpu-baseline --steps 4 > $OUT/bench_orth30_line.json
python3 $B --workload lap3d_100 --no-cpu-baseline > $OUT/bench_lap3d_100_line.json
python3 $B --workload lap3d_100 --orth 0 --no-cpu-baseline > $OUT/bench_lap3d_100_orth0_line.json
for o in 3 0 30; do
  echo "== kernel stats orth $o"
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_orth$o -o run -- python3 $B --orth $o --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_orth${o}_line_under_rocprof.json
done
for o in 3 0; do
  echo "== kernel stats lap3d_100 orth $o"
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_lap3d_100_orth$o -o run -- python3 $B --workload lap3d_100 --orth $o --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_lap3d_100_orth${o}_line_under_rocprof.json
done
for cfg in "lap2d_1000 3" "lap2d_1000 0" "lap2d_1000 6" "lap2d_1000 30" "lap3d_100 3" "lap3d_100 0"; do
  set -- $cfg; w=$1; o=$2
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $c $w orth $o"
    rocprofv3 --output-format csv --kernel-trace --pmc $c -d $OUT/pmc_${c}_${w}_orth$o -o run -- python3 $B --workload $w --orth $o --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null
  done
done
echo "== kernel stats dense (configs[0] operator: 5000^2, 64 / 128 probes)"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_dense -o run -- python3 $ROOT/scripts/bench_dense.py > $OUT/bench_dense.log 2>&1
python3 $ROOT/scripts/summarise_pmc.py $OUT > $OUT/pmc_summary.json
echo "== narrow panels (64 probes) and a six-column window: bench lines + kernel stats"
python3 $B --probes 64 --no-cpu-baseline --no-extra > $OUT/bench_p64_line.json
python3 $B --workload lap3d_100 --probes 64 --no-cpu-baseline --no-extra > $OUT/bench_lap3d_100_p64_line.json
python3 $B --orth 6 --no-cpu-baseline --no-extra > $OUT/bench_orth6_line.json
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_lap3d_100_p64 -o run -- python3 $B --workload lap3d_100 --probes 64 --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_lap3d_100_p64_line_under_rocprof.json
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_orth6 -o run -- python3 $B --orth 6 --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_orth6_line_under_rocprof.json
if [ "${SKIP_CONFIGS:-0}" != "1" ]; then  # (a call of its own when the box's time limit is short: SKIP_CONFIGS=1 here, then the script itself)
  echo "== configs[3] and configs[4] under rocprofv3 (scripts/profile_configs.sh)"
  bash $ROOT/scripts/profile_configs.sh ${TAG}_configs
fi
find $OUT -name "*kernel_stats.csv" | while read f; do d=$(basename $(dirname $f)); cp $f $OUT/${d}_kernel_stats.csv 2>/dev/null || true; done
# the raw traces are large: keep summaries only
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
