# usage: alone_times.sh base v1 v2 ...: stages_alone_ms of each library variant
mkdir -p gpurun_out/fv
for v in "$@"; do
  if [ $v = base ]; then unset SPINRELAX_HIP_LIB; else export SPINRELAX_HIP_LIB=$PWD/_variants/lib_$v.so; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/fv/al.json 2> gpurun_out/fv/al.err || { tail -5 gpurun_out/fv/al.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/fv/al.json'))
a=d['stages_alone_ms']
print('%-10s step %.3f  pack %.4f ct %.4f hist %.4f fin %.4f'%('$v', d['ms_per_step'], a['pack'], a['ct'], a['hist'], a['ct_finalize']))
"
done
