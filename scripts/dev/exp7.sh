cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp7
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -x -q -k "grouped or batched_order" > gpurun_out/exp7/pytest.log 2>&1 || { tail -30 gpurun_out/exp7/pytest.log; exit 1; }
tail -2 gpurun_out/exp7/pytest.log
run() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall --no-kernel-profile "$@" > gpurun_out/exp7/$tag.json 2> gpurun_out/exp7/$tag.err || tail -5 gpurun_out/exp7/$tag.err; python - <<P
import json
d=json.load(open('gpurun_out/exp7/$tag.json'))
k=d['kernels']
print('$tag', 'step %.3f'%d['ms_per_step'], ['%.2f'%x for x in d['timed_region_samples_ms_per_step']], 'steady %.3f'%d['ms_per_step_steady'], 'inpipe ct %.2f fit %.2f'%(k['k_ct_rfft']['in_pipeline_ms'], k['k_order_search']['in_pipeline_ms']), flush=True)
P
}
run base
run late --late-hist 1
run base2
run late2 --late-hist 1
