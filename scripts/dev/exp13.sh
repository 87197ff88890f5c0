cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in base ct1 ct2 ct3; do
  if [ $v = base ]; then unset SPINRELAX_HIP_LIB; else export SPINRELAX_HIP_LIB=$PWD/_variants/lib_$v.so; fi
  echo "$v: $(CT_FFT=2 timeout -k 10 100 python scripts/dev/ct_time.py 2>/dev/null | tail -1)  | palmer $(CT_FFT=0 REPS=3 timeout -k 10 100 python scripts/dev/ct_time.py 2>/dev/null | tail -1)"
done
done
