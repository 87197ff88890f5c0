cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp5
run() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall --no-kernel-profile "$@" > gpurun_out/exp5/$tag.json 2> gpurun_out/exp5/$tag.err; python - <<P
import json
d=json.load(open('gpurun_out/exp5/$tag.json'))
k=d['kernels']
print('$tag', 'step %.3f'%d['ms_per_step'], ['%.2f'%x for x in d['timed_region_samples_ms_per_step']], 'steady %.3f'%d['ms_per_step_steady'], 'inpipe ct %.2f hist %.2f'%(k['k_ct_rfft']['in_pipeline_ms'], k['k_vechist']['in_pipeline_ms']), flush=True)
P
}
run skip_t1 --dev-skip-fits
SR_DEV_TAILS=3 run skip_t3 --dev-skip-fits
SR_DEV_TAILS=5 run skip_t5 --dev-skip-fits --psum-buffers 5
run skip_g1 --dev-skip-fits --group 1
SR_DEV_TAILS=3 run t3
run t1
