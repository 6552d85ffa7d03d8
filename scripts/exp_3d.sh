for env in "X=0" "SLQ_RCM_SUB=2" "SLQ_RCM_SUB=8" "SLQ_RCM_SUB=16" "SLQ_RING_ORDER=1"; do
  echo "== $env"
  env $env python bench.py --workload lap3d_100 --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['value'], d['ms_per_step'], 'alpha ms/launch', round(k['spmm_3term']['ms_per_step']/k['spmm_3term']['launches_per_step'],4), 'update', round(k['reorth_update']['ms_per_step']/k['reorth_update']['launches_per_step'],4), 'create', d['config']['create_s'])"
done
