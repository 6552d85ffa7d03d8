#!/bin/bash
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['estimate'])"; }
echo "o0 $(run --orth 0)"
echo "o3 $(run --orth 3)"
echo "o30 $(run --orth 30)"
echo "3d o3 $(run --orth 3 --workload lap3d_100)"
echo "3d o0 $(run --orth 0 --workload lap3d_100)"
echo "3d f32 k50 $(run --orth 3 --workload lap3d_126 --dtype f32 --deg 50)"
echo "2d f32 $(run --orth 3 --dtype f32)"
echo "2d p64 $(run --orth 3 --probes 64)"
