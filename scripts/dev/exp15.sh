cd $GRAFT_REPO_ROOT
for st in 0 4352 17664; do echo "== stagger $st"; SR_DEV_STAGGER=$st timeout -k 10 200 python scripts/dev/first_region.py 2>/dev/null | grep "20ev" | head -6 | cut -c1-95; done
