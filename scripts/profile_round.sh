#!/bin/bash
# Collect the rocprofv3 material behind profiles/<tag>_*: run ON the GPU box from the repo root, e.g.
#   gpurun --timeout 1100 -- 'bash scripts/profile_round.sh r02'
# then, back in the container:  python scripts/make_profiles.py r02 gpurun_out/r02prof/{stats,FETCH_SIZE,WRITE_SIZE,FP64}
# One pass per counter set (MI355X_MICROARCH.md, HBM / rocprofv3 section); --pmc is never combined with any trace domain
# other than the kernel trace.  The program itself follows `--` (no env / bash -c hop: the profiler's preloaded
# library has initialised the GPU by then).
set -e
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/${tag}prof
mkdir -p $out
cd /tmp
export TMPDIR=/tmp
# (1) kernel statistics of the benchmark command the driver runs (the per-kernel alone / saturated launches after the
#     timed region are left out so that the averages are the in-pipeline durations)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- \
    python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > $out/stats.log 2>&1
# (2-4) counters, one batch at a time (--depth 1): the counters are device-wide, kernels of overlapping batches would
#     be charged to each other
for pass in "FETCH_SIZE:FETCH_SIZE" "WRITE_SIZE:WRITE_SIZE" \
            "FP64:SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$name -- \
        python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile --depth 1 > $out/$name.log 2>&1
    echo "pass $name done"
done
tail -c 300 $out/stats.log
