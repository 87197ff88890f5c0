cd $GRAFT_REPO_ROOT
export BRIEF=1
for gate in "" 1; do
export GATE=$gate
for sz in 20 16,4 14,6 12,8 12,5,3 10,6,4; do echo "== gate=$gate sizes $sz"; timeout -k 10 200 python scripts/dev/group_timeline.py 20 32 1 $sz 2>/dev/null | grep -v "^$"; done
done
