#!/bin/bash
# usage (on the GPU box): pmc.sh <out tag> "<counters of pass 1>" "<counters of pass 2>" ... -- python3 script args
# one rocprofv3 --pmc pass per counter group; prints per-kernel sums of every counter
tag=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
  timeout -k 10 400 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; exit 1; }
  i=$((i+1))
done
python3 - <<P
import glob, pandas as pd, re
for f in sorted(glob.glob('$out/p*/**/*counter_collection.csv', recursive=True)):
    t = pd.read_csv(f)
    t['k'] = t.Kernel_Name.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace(r'^void ', '', regex=True).str.replace(r'\(.*$', '', regex=True)
    per = t.groupby(['k', 'Dispatch_Id', 'Counter_Name']).Counter_Value.sum().reset_index()
    last = per.groupby(['k', 'Counter_Name']).Counter_Value.agg(['mean', 'max', 'count']).reset_index()
    for _, r in last.iterrows():
        if r['k'].startswith('__amd') or r['k'].startswith('at::'): continue
        print('%-34s %-28s mean %.4g  max %.4g  (%d launches)' % (r['k'][:34], r['Counter_Name'], r['mean'], r['max'], r['count']))
P
