cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp4
run() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall --no-kernel-profile "$@" > gpurun_out/exp4/$tag.json 2> gpurun_out/exp4/$tag.err; python - <<P
import json
d=json.load(open('gpurun_out/exp4/$tag.json'))
k=d['kernels']
print('$tag', 'step %.3f'%d['ms_per_step'], ['%.2f'%x for x in d['timed_region_samples_ms_per_step']], 'steady %.3f'%d['ms_per_step_steady'], 'inpipe ct %.2f hist %.2f fit %.2f'%(k['k_ct_rfft']['in_pipeline_ms'], k['k_vechist']['in_pipeline_ms'], k['k_order_search']['in_pipeline_ms']), flush=True)
P
}
run base
run tr --ct-traceless 1
run base2
run tr2 --ct-traceless 1
run pb4 --plane-buffers 4
run pb2 --plane-buffers 2
