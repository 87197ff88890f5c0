# developer sweep (round 3): pipeline flags on one box, short runs
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-cli-wall --steady-steps 200 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'ms/step %.3f'%d['ms_per_step'], ['%.3f'%x for x in d['timed_region_samples_ms_per_step']], 'steady %.3f'%d['ms_per_step_steady'])"; }
run
run --depth 4
run --depth 6
run --depth 7
run --plane-buffers 2
run --plane-buffers 4
run --fit-priority -1
run --hist-on-main
run
