# usage: flag_sweep.sh "<common flags>" "<flags A>" "<flags B>" ...  -> ms_per_step for each
mkdir -p gpurun_out/fv
common=$1; shift
for f in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-profile $common $f > gpurun_out/fv/fs.json 2> gpurun_out/fv/fs.err || { tail -5 gpurun_out/fv/fs.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/fv/fs.json'))
k=d['kernels']
print('%-28s %.3f ms/step   in-pipeline: ct %.2f hist %.2f fit %.2f'%('$f', d['ms_per_step'], [v for n,v in k.items() if n.startswith('k_ct_r') or n=='k_ct_fft'][0].get('in_pipeline_ms',0), k['k_vechist'].get('in_pipeline_ms',0), k['k_order_search'].get('in_pipeline_ms',0)))
"
done
