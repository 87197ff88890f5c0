#!/usr/bin/env python3
"""
Turn raw rocprofv3 output (under gpurun_out/) into the summaries committed under profiles/.

    python scripts/make_profiles.py <round tag> <stats dir> <FETCH_SIZE dir> <WRITE_SIZE dir> [<FP64 instruction dir>]

  stats dir       rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python bench.py ...
  FETCH/WRITE dir rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace --output-format csv -d <dir> -- python bench.py ...
                  (separate passes, as MI355X_MICROARCH.md prescribes; never combined with other trace domains)
  FP64 dir        rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
                  --kernel-trace ... (optional): executed float64 flop per launch = 64 lanes x (ADD + MUL + 2 FMA + TRANS)
                  wave-instructions, stored as fp64_flop_per_launch (bench.py prices the fit kernel with it)
Writes profiles/<tag>_bench_cfg3_kernel_stats.csv and profiles/<tag>_bench_cfg3_hbm_counters.json.
"""
import glob
import json
import os
import re
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)


def counters(d, counter):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    t = pd.read_csv(f)
    t = t[t.Counter_Name == counter]
    t['k'] = t.Kernel_Name.map(short)
    per_launch = t.groupby(['k', 'Dispatch_Id']).Counter_Value.sum().reset_index()
    return per_launch.groupby('k').Counter_Value.agg(['mean', 'count'])


def main():
    tag, dstats, dfetch, dwrite = sys.argv[1:5]
    dfp64 = sys.argv[5] if len(sys.argv) > 5 else None
    f = glob.glob(os.path.join(dstats, '**', '*kernel_stats.csv'), recursive=True)[0]
    st = pd.read_csv(f)
    st.to_csv(os.path.join(ROOT, 'profiles', '%s_bench_cfg3_kernel_stats.csv' % tag), index=False)
    fe, wr = counters(dfetch, 'FETCH_SIZE'), counters(dwrite, 'WRITE_SIZE')
    out = {'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 '
                      '--no-cpu-baseline --depth 1 (separate passes; --depth 1 = one batch at a time: the counters are '
                      'device-wide, kernels of overlapping batches would be charged to each other)',
           'note': 'units KB; corrected bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md section HBM '
                   '(gfx950 reports half of the streamed read bytes)',
           'kernels': {}}
    for k in sorted(set(fe.index) | set(wr.index)):
        if k.startswith('__amd') or k.startswith('at::'):
            continue
        fk = float(fe.loc[k, 'mean']) if k in fe.index else 0.0
        wk = float(wr.loc[k, 'mean']) if k in wr.index else 0.0
        out['kernels'][k] = {'FETCH_SIZE_KB_mean_per_launch': fk, 'launches_FETCH_SIZE': int(fe.loc[k, 'count']) if k in fe.index else 0,
                             'WRITE_SIZE_KB_mean_per_launch': wk, 'launches_WRITE_SIZE': int(wr.loc[k, 'count']) if k in wr.index else 0,
                             'hbm_bytes_corrected': (2 * fk + wk) * 1024}
    if dfp64:
        cnt = {c: counters(dfp64, 'SQ_INSTS_VALU_%s_F64' % c) for c in ('ADD', 'MUL', 'FMA', 'TRANS')}
        out['fp64_note'] = ('fp64_flop_per_launch = 64 x (ADD_F64 + MUL_F64 + 2 FMA_F64 + TRANS_F64) wave-instructions per launch '
                            '(SQ_INSTS_VALU_*_F64, full exec mask assumed; exp / division expand into these)')
        for k in out['kernels']:
            if all(k in cnt[c].index for c in cnt):
                w = {c: float(cnt[c].loc[k, 'mean']) for c in cnt}
                out['kernels'][k]['fp64_wave_instructions'] = w
                out['kernels'][k]['fp64_flop_per_launch'] = 64.0 * (w['ADD'] + w['MUL'] + 2.0 * w['FMA'] + w['TRANS'])
    with open(os.path.join(ROOT, 'profiles', '%s_bench_cfg3_hbm_counters.json' % tag), 'w') as fp:
        json.dump(out, fp, indent=1)
    print(st[['Name', 'Calls', 'AverageNs', 'Percentage']].head(12).to_string())
    print({k: round(v['hbm_bytes_corrected'] / 1e6, 1) for k, v in out['kernels'].items()})


if __name__ == '__main__':
    main()
