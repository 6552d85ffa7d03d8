#!/bin/bash
# rocprofv3 kernel stats of the configs[3] (fp32 126^3 diag) and configs[4] (n = 1e7, full reorth; 2 batches of 32 probes)
# runs of scripts/run_configs.py:  bash scripts/profile_configs.sh <tag>  ->  gpurun_out/<tag>/{c4,c5}_kernel_stats.csv
set -eo pipefail
TAG=${1:-r03_configs}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN_TAG=${RUN_TAG:-r04}_prof rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/c4 -o run -- python3 $ROOT/scripts/run_configs.py c4 > $OUT/c4.log 2>&1
RUN_TAG=${RUN_TAG:-r04}_prof C5_PROBES=64 C5_RING32=0 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/c5 -o run -- python3 $ROOT/scripts/run_configs.py c5 > $OUT/c5.log 2>&1
for c in c4 c5; do f=$(find $OUT/$c -name "*kernel_stats.csv" | head -1); cp $f $OUT/${c}_kernel_stats.csv; done
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
head -8 $OUT/c4_kernel_stats.csv | cut -c1-200; head -8 $OUT/c5_kernel_stats.csv | cut -c1-200
