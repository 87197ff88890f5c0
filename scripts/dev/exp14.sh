cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp14
run() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall --no-kernel-profile "$@" > gpurun_out/exp14/$tag.json 2> gpurun_out/exp14/$tag.err || tail -5 gpurun_out/exp14/$tag.err; python - <<P
import json
d=json.load(open('gpurun_out/exp14/$tag.json'))
k=d['kernels']
print('$tag', 'step %.3f'%d['ms_per_step'], ['%.2f'%x for x in d['timed_region_samples_ms_per_step']], 'steady %.3f'%d['ms_per_step_steady'], 'inpipe hist %.2f fit %.2f'%(k['k_vechist']['in_pipeline_ms'], k['k_order_search']['in_pipeline_ms']), flush=True)
P
}
run base
run lds40 --hist-lds-kb 40
run lds56 --hist-lds-kb 56
run lds80 --hist-lds-kb 80
run base2
run lds56b --hist-lds-kb 56
