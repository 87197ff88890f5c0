#!/bin/bash
# round-4 experiment batch A (run on the GPU box from the repo root): compiler-scheduler variants of kernel 1 and the fit,
# 8 waves per residue, table-driven exp; per-rank step of the strong-scaling shards
cd ${GRAFT_REPO_ROOT:-$PWD}
out=gpurun_out/exp_r4a; mkdir -p $out
for v in base ct_maxilp ct_memclause ct_trackers ct_bias0; do
  lib=""; [ $v != base ] && lib=$PWD/_variants/lib_$v.so
  echo "== ct $v" | tee -a $out/ct.log
  SPINRELAX_HIP_LIB=$lib REPS=9 timeout -k 10 120 python scripts/dev/ct_time.py 2>&1 | tail -1 | tee -a $out/ct.log
done
for v in fit_fast:4 fit_fast:8 fit_tab:4 fit_tab:8 fit_maxilp:4 fit_trackers:4; do
  n=${v%%:*}; w=${v#*:}
  echo "== fit $n W=$w" | tee -a $out/fit.log
  SPINRELAX_HIP_LIB=$PWD/_variants/lib_$n.so timeout -k 10 200 python scripts/dev/fit_probe.py 16 fit_waves=$w 2>&1 | tail -2 | tee -a $out/fit.log
done
for nv in 256 128 64; do
  echo "== strong shard $nv vectors" | tee -a $out/shard.log
  timeout -k 10 200 python bench.py --vectors $nv --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall --no-kernel-profile --steady-steps 60 2>/dev/null \
    | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['config']['vectors_per_gpu'], 'ms_per_step', j['ms_per_step'], j['timed_region_samples_ms_per_step'], 'steady', j['ms_per_step_steady'], 'fit merged ms', j['kernels']['k_order_search']['in_pipeline_ms'], 'ct ms', j['kernels']['k_ct_rfft']['in_pipeline_ms'])" | tee -a $out/shard.log
done
