#!/usr/bin/env python3
"""Developer tool: main-stream gaps between consecutive C(t) launches, and what ended just before each start.
usage: trace_gaps.py <dir with *_kernel_trace.csv>"""
import glob, sys
import pandas as pd
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
t = pd.read_csv(f)
t['k'] = t.Kernel_Name.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace(r'^void ', '', regex=True).str.replace(r'[<(].*$', '', regex=True)
t = t.sort_values('Start_Timestamp').reset_index(drop=True)
r = t[t.k == 'k_ct_rfft'].reset_index(drop=True)
t0 = r.Start_Timestamp.iloc[-20]
for i in range(len(r) - 20, len(r)):
    st = r.Start_Timestamp.iloc[i]
    prev_end = r.End_Timestamp.iloc[i - 1]
    # kernels that ended in (prev_end, st]
    w = t[(t.End_Timestamp > prev_end) & (t.End_Timestamp <= st)]
    pk = t[(t.k == 'k_pack_soa') & (t.End_Timestamp <= st)].iloc[-1]
    print('rfft %2d start %7.2f dur %.2f gap %.3f | last pack: start %7.2f end %7.2f | ended in gap: %s' % (
        i, (st - t0) / 1e6, (r.End_Timestamp.iloc[i] - st) / 1e6, (st - prev_end) / 1e6,
        (pk.Start_Timestamp - t0) / 1e6, (pk.End_Timestamp - t0) / 1e6, ','.join('%s@%.2f' % (a, (b - t0) / 1e6) for a, b in zip(w.k, w.End_Timestamp))))
print('end %.2f' % ((t.End_Timestamp.max() - t0) / 1e6))
