#!/bin/bash
# Per-rank step times of the strong-scaling shards, one rank at a time on ONE GPU (bench.py --shard r/N): the slowest rank of
# every N sets the pace of the N-GPU run.  usage (GPU box, repo root): bash scripts/dev/shard_table.sh
short="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-cli-wall --no-h2d-stream"
for N in 1 2 4 8; do
    for ((r = 0; r < N; ++r)); do
        timeout -k 10 200 python bench.py $short --shard $r/$N 2>/dev/null |
            python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 shard $r/$N: %.3f ms per step (20 steps), %.3f steady, random dispatch %.3f' % (j['ms_per_step'], j['ms_per_step_steady'], j['ms_per_step_random_dispatch']))" || exit 1
    done
done
timeout -k 10 300 python bench.py $short --workload cfg4 --shard 0/8 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 shard 0/8 (256 vectors): %.3f ms per step (20 steps), %.3f steady' % (j['ms_per_step'], j['ms_per_step_steady']))"
