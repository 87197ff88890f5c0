# usage: run_variants.sh "<bench flags>" base u ...   (variants = _variants/lib_<name>.so; base = the in-tree library)
set -e
mkdir -p gpurun_out/fv
flags=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset SPINRELAX_HIP_LIB; else export SPINRELAX_HIP_LIB=$PWD/_variants/lib_$v.so; fi
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $flags > gpurun_out/fv/$v.json 2> gpurun_out/fv/$v.err
  python - <<P
import json
d=json.load(open('gpurun_out/fv/$v.json'))
a=d['stages_alone_ms']
print('$v', 'step %.3f'%d['ms_per_step'], 'fit alone %.2f sat %.3f'%(a.get('fit',0), a.get('fit_saturated_per_batch',0)), 'lat %.2f'%d['latency_ms']['min'], 'ct %.3f'%a.get('ct',0), d['fit']['evaluations_per_batch'])
P
done
