cd $GRAFT_REPO_ROOT
echo "== base, LDS"; timeout -k 10 200 python scripts/dev/fit_by_order.py 2>/dev/null | grep orders | head -3
echo "== base, fit_lds=0"; timeout -k 10 200 python scripts/dev/fit_by_order.py fit_lds=0 2>/dev/null | grep orders
for v in low3 low4; do export SPINRELAX_HIP_LIB=$PWD/_variants/lib_$v.so
echo "== $v LDS"; timeout -k 10 200 python scripts/dev/fit_by_order.py fit_lds=1 2>/dev/null | grep orders
echo "== $v fit_lds=0"; timeout -k 10 200 python scripts/dev/fit_by_order.py fit_lds=0 2>/dev/null | grep orders
done
