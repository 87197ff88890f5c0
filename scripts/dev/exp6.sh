cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp6
run() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall --no-kernel-profile "$@" > gpurun_out/exp6/$tag.json 2> gpurun_out/exp6/$tag.err; python - <<P
import json
d=json.load(open('gpurun_out/exp6/$tag.json'))
k=d['kernels']
print('$tag', 'step %.3f'%d['ms_per_step'], ['%.2f'%x for x in d['timed_region_samples_ms_per_step']], 'steady %.3f'%d['ms_per_step_steady'], 'inpipe ct %.2f'%(k['k_ct_rfft']['in_pipeline_ms']), flush=True)
P
}
run skipfits --dev-skip-fits
SR_DEV_SKIP_HIST=1 run skipfits_nohist --dev-skip-fits
run full
SR_DEV_SKIP_HIST=1 run full_nohist
