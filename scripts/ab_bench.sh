#!/bin/bash
# A/B of library builds on the GPU box: scripts/ab_bench.sh <tag> "<lib1> <lib2> ..." "<workload:orth> ..." [extra bench args]
# each lib is a path relative to the repo root ("default" = primate_amd/_libslq.so); one bench line per (lib, case)
set -eo pipefail
TAG=$1; LIBS=$2; CASES=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
for lib in $LIBS; do
  for c in $CASES; do
    w=${c%%:*}; o=${c##*:}
    name=$(basename $lib .so)
    if [ "$lib" = "default" ]; then unset PRIMATE_AMD_LIBSLQ; else export PRIMATE_AMD_LIBSLQ=$ROOT/$lib; fi
    python3 $ROOT/bench.py --workload $w --orth $o --no-cpu-baseline --steps 5 --warmup 2 "$@" > $OUT/${name}_${w}_o${o}.json 2> $OUT/${name}_${w}_o${o}.err || { echo "FAILED $lib $c"; tail -5 $OUT/${name}_${w}_o${o}.err; exit 1; }
    python3 - $OUT/${name}_${w}_o${o}.json $name $w $o <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print(sys.argv[2],sys.argv[3],"orth",sys.argv[4],"ms/step",d["ms_per_step"],"pmv/s",d["value"], " | ".join(f"{n}:{v['ms_per_step']/max(v['launches_per_step'],1):.3f}ms/{v.get('alg_GBps','-')}GB/s" for n,v in k.items() if n!="finalize" and n!="quadrature"), flush=True)
PY
  done
done
