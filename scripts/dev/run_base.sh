mkdir -p gpurun_out/fv
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/fv/base2.json 2>gpurun_out/fv/base2.err || exit 1
python - <<P
import json
d=json.load(open('gpurun_out/fv/base2.json'))
print(d['ms_per_step'], d['stages_alone_ms'])
P
