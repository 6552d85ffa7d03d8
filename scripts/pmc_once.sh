#!/bin/bash
# usage: bash scripts/pmc_once.sh <tag> <bench args...>   -> prints per-kernel HBM bytes per launch (FETCH x2 + WRITE)
set -eo pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --kernel-trace --pmc $c -d $OUT/pmc_$c -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/log_$c.txt
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c:
                acc[row["Kernel_Name"].split("(")[0]][c].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
    f, w = d.get("FETCH_SIZE", [0]), d.get("WRITE_SIZE", [0])
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    if len(f) >= 3:
        print(f"{k[:70]:70s} n={len(f):4d} fetch={2*fa*1024/1e9:7.3f} GB write={wa*1024/1e9:7.3f} GB total={(2*fa+wa)*1024/1e9:7.3f} GB")
PY
