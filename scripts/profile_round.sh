#!/bin/bash
# Collect the rocprofv3 material behind profiles/<tag>_*: run ON the GPU box from the repo root, e.g.
#   gpurun --timeout 1100 -- 'bash scripts/profile_round.sh r03'
# then, back in the container:  python scripts/make_profiles.py r03 gpurun_out/r03prof
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
# which library the figures belong to (spinrelax_amd/build.py:build_id, compiled into the library): stored in every JSON summary
python3 -c "import sys; sys.path.insert(0, '$root'); from spinrelax_amd import _lib; print(_lib.load().sr_build_id().decode())" > $out/build_id.txt
short="--no-cpu-baseline --no-kernel-profile --no-cli-wall --spinup-s 0 --repeats 1 --steady-steps 0"
# (1) kernel statistics of the benchmark command the driver runs (the per-kernel alone / saturated launches after the
#     timed region, the CPU baseline and the CLI chain are left out so that the averages are the in-pipeline durations)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- \
    python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-cli-wall > $out/stats.log 2>&1
echo "stats done"
# (1b) the same kernels ONE BATCH AT A TIME (--depth 1: nothing overlaps): every average is the kernel's duration alone on
#      the chip -- the "alone" fractions can be recomputed from this file without bench.py
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_alone -- \
    python3 $root/bench.py --steps 6 --warmup 2 --depth 1 $short > $out/stats_alone.log 2>&1
echo "stats_alone done"
# (2-4) counters, one batch at a time (--depth 1): the counters are device-wide, kernels of overlapping batches would
#     be charged to each other
for pass in "FETCH_SIZE:FETCH_SIZE" "WRITE_SIZE:WRITE_SIZE" \
            "FP64:SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
            "FP32:SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FLOPS_FP32" \
            "RDREQ:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$name -- \
        python3 $root/bench.py --steps 3 --warmup 1 --depth 1 $short > $out/$name.log 2>&1 || echo "pass $name FAILED"
    echo "pass $name done"
done
# (4b) issue-side counters of the two compute kernels (SQ block: 8 counters per pass), one batch at a time: how busy the VALU
#      is, what the waves wait for, LDS activity -- the evidence behind "x % of the issue-limited ceiling" 
for pass in "ISSUE1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
            "ISSUE2:SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM" \
            "ISSUE3:SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
            "ISSUE4:MeanOccupancyPerCU" "ISSUE5:LdsLatency" "ISSUE6:VmemLatency"; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$name -- \
        python3 $root/bench.py --steps 3 --warmup 1 --depth 1 $short > $out/$name.log 2>&1 || echo "pass $name FAILED"
    echo "pass $name done"
done
# (5) the north-star's named kernel (direct shifted products, k_ct_palmer) alone: HBM traffic and FP32 work
for pass in "palmer_FETCH:FETCH_SIZE" "palmer_WRITE:WRITE_SIZE" "palmer_VALU:SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"; do
    name=${pass%%:*}; ctr=${pass#*:}
    CT_FFT=0 REPS=3 timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$name -- \
        python3 $root/scripts/dev/ct_time.py > $out/$name.log 2>&1 || echo "pass $name FAILED"
    echo "pass $name done"
done
# (6) BASELINE cfg2's size (F + L = 1536 -> k_ct_fft<8>): statistics and HBM counters
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg2_stats -- \
    python3 $root/bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-cli-wall > $out/cfg2_stats.log 2>&1 || echo "cfg2 stats FAILED"
for pass in "cfg2_FETCH:FETCH_SIZE" "cfg2_WRITE:WRITE_SIZE" "cfg2_FP64:SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$name -- \
        python3 $root/bench.py --workload cfg2 --steps 3 --warmup 1 --depth 1 $short > $out/$name.log 2>&1 || echo "pass $name FAILED"
    echo "pass $name done"
done
tail -c 300 $out/stats.log
