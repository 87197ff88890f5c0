#!/bin/bash
# Developer A/B on one box: time the C(t) sums kernel (scripts/dev/ct_time.py, CT_FFT=3) with each variant library of _variants/
# named on the command line, round-robin, ROUNDS times.   usage: ct32_ab.sh ROUNDS name [name ...]   ("base" = the regular build)
rounds=$1; shift
for r in $(seq $rounds); do
  for n in "$@"; do
    if [ "$n" = base ]; then lib="spinrelax_amd/libspinrelax_hip.so"; else lib="_variants/lib_$n.so"; fi
    printf '%-10s ' $n
    SPINRELAX_HIP_LIB=$PWD/$lib CT_FFT=${CT_FFT:-3} REPS=${REPS:-15} timeout -k 10 200 python3 scripts/dev/ct_time.py 2>&1 | tail -1
  done
done
