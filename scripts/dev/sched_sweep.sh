#!/bin/bash
# Developer sweep of the grouped schedule's knobs on one box: every line = one bench.py run (20 steps x 3, 120 steady).
# usage (GPU box, repo root): bash scripts/dev/sched_sweep.sh "<flags A>" "<flags B>" ...
for f in "$@"; do
    echo "== $f"
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steady-steps 120 --no-cpu-baseline --no-kernel-profile --no-cli-wall --no-h2d-stream $f 2>/dev/null |
        python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f steady' % j['ms_per_step'], j['ms_per_step_steady'], ['%.3f' % x for x in j['timed_region_samples_ms_per_step']], 'random', j['ms_per_step_random_dispatch'])" || exit 1
done
