for lib in "" build_ab/libslq_c48.so; do for w in lap3d_100 lap2d_1000; do for o in 3 0; do
  echo "== lib=${lib:-product} $w orth $o"
  PRIMATE_AMD_LIBSLQ=${lib:+$PWD/$lib} SLQ_FUSED_ALPHA=0 python bench.py --workload $w --orth $o --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; u='reorth_update' if 'reorth_update' in k else 'axpy_norm'; print(d['value'], d['ms_per_step'], 'alpha ms/launch', round(k['spmm_3term']['ms_per_step']/k['spmm_3term']['launches_per_step'],4), 'update', round(k[u]['ms_per_step']/k[u]['launches_per_step'],4), 'est', d['estimate'])"
done; done; done
