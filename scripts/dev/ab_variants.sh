# usage: ab_variants.sh <out tag> <rounds> base v1 v2 ...   -- alternating runs on ONE box; prints 20-step median, steady, alone C(t)
set -e
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out/$tag
for r in $(seq 1 $rounds); do
for v in "$@"; do
  if [ $v = base ]; then unset SPINRELAX_HIP_LIB; else export SPINRELAX_HIP_LIB=$PWD/_variants/lib_$v.so; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall $SR_AB_FLAGS > gpurun_out/$tag/$v.$r.json 2> gpurun_out/$tag/$v.$r.err
  python - <<P
import json
d=json.load(open('gpurun_out/$tag/$v.$r.json'))
a=d.get('stages_alone_ms',{})
k=d['kernels']
print('$v', 'step %.3f'%d['ms_per_step'], ['%.2f'%x for x in d['timed_region_samples_ms_per_step']], 'steady %.3f'%d['ms_per_step_steady'], 'ct alone %.3f'%a.get('ct',0), 'inpipe ct %.2f hist %.2f fit %.2f'%(k['k_ct_rfft']['in_pipeline_ms'], k['k_vechist']['in_pipeline_ms'], k['k_order_search']['in_pipeline_ms']), 'lat %.2f'%d['latency_ms']['min'], flush=True)
P
done
done
