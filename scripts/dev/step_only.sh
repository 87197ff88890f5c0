# usage: step_only.sh "<flags A>" "<flags B>" ... -> ms_per_step only
mkdir -p gpurun_out/fv
for f in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 5 $f > gpurun_out/fv/so.json 2> gpurun_out/fv/so.err || { tail -5 gpurun_out/fv/so.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/fv/so.json'))
print('%-40s %.3f ms/step'%('$f', d['ms_per_step']))
"
done
