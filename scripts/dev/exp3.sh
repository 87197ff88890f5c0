cd $GRAFT_REPO_ROOT
for st in 0 1; do
if [ $st = 1 ]; then export NOSTAGGER=1; fi
echo "== nostagger=$NOSTAGGER"; timeout -k 10 200 python scripts/dev/group_timeline.py 20 32 1 2>/dev/null | grep -v "^$"
done
