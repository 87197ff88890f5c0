cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp10
run() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall --no-kernel-profile --steady-steps 192 "$@" > gpurun_out/exp10/$tag.json 2> gpurun_out/exp10/$tag.err || tail -5 gpurun_out/exp10/$tag.err; python - <<P
import json
d=json.load(open('gpurun_out/exp10/$tag.json'))
k=d['kernels']
print('$tag', 'step %.3f'%d['ms_per_step'], ['%.2f'%x for x in d['timed_region_samples_ms_per_step']], 'steady %.3f'%d['ms_per_step_steady'], 'inpipe ct %.2f fit %.2f'%(k['k_ct_rfft']['in_pipeline_ms'], k['k_order_search']['in_pipeline_ms']), flush=True)
P
}
run gate
run nogate --no-gate
run gate2
run nogate2 --no-gate
run gate_g16 --group 16
run nogate_g16 --group 16 --no-gate
