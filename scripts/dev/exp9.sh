cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp9
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/exp9/pytest.log 2>&1 || { tail -30 gpurun_out/exp9/pytest.log; exit 1; }
tail -2 gpurun_out/exp9/pytest.log
run() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall "$@" > gpurun_out/exp9/$tag.json 2> gpurun_out/exp9/$tag.err || tail -5 gpurun_out/exp9/$tag.err; python - <<P
import json
d=json.load(open('gpurun_out/exp9/$tag.json'))
k=d['kernels']; a=d.get('stages_alone_ms',{})
print('$tag', 'step %.3f'%d['ms_per_step'], ['%.2f'%x for x in d['timed_region_samples_ms_per_step']], 'steady %.3f'%d['ms_per_step_steady'], 'inpipe ct %.2f hist %.2f fit %.2f'%(k['k_ct_rfft']['in_pipeline_ms'], k['k_vechist']['in_pipeline_ms'], k['k_order_search']['in_pipeline_ms']), 'alone fit %.2f sat %.3f lat %.2f'%(a.get('fit',0), a.get('fit_saturated_per_batch',0), d['latency_ms']['min'] if d.get('latency_ms') else 0), flush=True)
P
}
run default
run g1 --group 1
run early --late-hist 0
run default2
