#!/bin/bash
# usage: bash scripts/pmc_counters.sh <tag> "<pass1 counters>;<pass2 counters>;..." <bench args...>
# (PMC_SCRIPT="<script.py> <args>" profiles another python script instead of bench.py)
# one rocprofv3 --pmc pass per ';'-separated group (kernel-trace only, per the pool's rules); prints the
# per-kernel average of every counter over the launches of the run.
set -eo pipefail
TAG=$1; PASSES=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra GROUPS_ <<< "$PASSES"
i=0
for g in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --output-format csv --kernel-trace --pmc $g -d $OUT/pass$i -o run -- python3 ${PMC_SCRIPT:-$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline} "$@" > /dev/null 2> $OUT/log_pass$i.txt || { echo "pass $i ($g) failed"; tail -5 $OUT/log_pass$i.txt; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, d in acc.items():
    n = max(len(v) for v in d.values())
    if n < 3: continue
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    res[k]["launches"] = n
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("launches", 0)):
    if "k_csr_" in k or "k_ring" in k or "k_spmm" in k or "k_reorth" in k or "k_axpy" in k:
        print(k[:64], " ".join(f"{c}={v:.4g}" for c, v in d.items()))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -delete
