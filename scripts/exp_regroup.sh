# A/B: the upper-triangle stream on its own, longer tiles (SLQ_RING_UPPER_REGROUP, r04) against the base tiles
for w in lap3d_100 lap2d_1000; do for o in 3 0; do for rg in 0 1; do
  echo "== $w orth $o SLQ_RING_UPPER_REGROUP=$rg"
  SLQ_DEBUG=1 SLQ_RING_UPPER_REGROUP=$rg python bench.py --workload $w --orth $o --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2> /tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; u='reorth_update' if 'reorth_update' in k else 'axpy_norm'; print(d['value'], d['ms_per_step'], 'alpha ms/launch', round(k['spmm_3term']['ms_per_step']/k['spmm_3term']['launches_per_step'],4), 'update', round(k[u]['ms_per_step']/k[u]['launches_per_step'],4), 'est', d['estimate'], 'create', d['config']['create_s'])"; grep "upper-triangle stream on\|upper triangle:" /tmp/err.txt | head -2
done; done; done
