run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --steady-steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.3f'%d['ms_per_step'], ['%.3f'%x for x in d['timed_region_samples_ms_per_step']], 'steady %.3f'%d['ms_per_step_steady'], d['fit']['evaluations_by_order']['9'])"; }
(cd _r2 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r2 20-step ms/step %.3f'%d['ms_per_step'])")
(cd _r2 && python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r2 200-step ms/step %.3f'%d['ms_per_step'])")
run new
SR_DEV_OLD_FINALIZE=1 run new_oldfinalize
SPINRELAX_HIP_LIB=$PWD/_variants/lib_notr.so run notr
SPINRELAX_HIP_LIB=$PWD/_variants/lib_notr.so SR_DEV_OLD_FINALIZE=1 run notr_oldfinalize
