# ms_per_step as a function of --steps: T(K) = drain + K * steady
mkdir -p gpurun_out/fv
for k in 10 20 40 80 160; do
  timeout -k 10 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-kernel-profile "$@" > gpurun_out/fv/sw.json 2> gpurun_out/fv/sw.err || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/fv/sw.json'))
print($k, '%.3f ms/step  total %.2f ms'%(d['ms_per_step'], d['ms_per_step']*$k))
"
done
