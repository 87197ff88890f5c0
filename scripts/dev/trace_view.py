#!/usr/bin/env python3
"""Developer tool: print the kernel timeline of the last pipeline iterations from a rocprofv3 kernel-trace csv."""
import sys
import glob
import pandas as pd

path = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
nper = int(sys.argv[2]) if len(sys.argv) > 2 else 2
d = pd.read_csv(path)
d['name'] = (d.Kernel_Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '', regex=False)
             .str.replace(r'\(.*', '', regex=True).str.slice(0, 30))
t0 = d.Start_Timestamp.min()
d['s'] = (d.Start_Timestamp - t0) / 1e6
d['e'] = (d.End_Timestamp - t0) / 1e6
d['dur'] = d.e - d.s
ct = d[d.name.str.contains('k_ct_palmer')]
a = ct.iloc[-2 - nper].s
b = ct.iloc[-2].e
w = d[(d.e >= a) & (d.s <= b) & ~d.name.str.contains('rocclr|at::native')].sort_values('s')
pd.set_option('display.width', 250)
pd.set_option('display.max_rows', 500)
print(w[['name', 'Queue_Id', 'Stream_Id', 's', 'e', 'dur']].to_string())
