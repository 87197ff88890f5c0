#!/usr/bin/env python3
"""
Turn raw rocprofv3 output (under gpurun_out/<tag>prof/, collected by scripts/profile_round.sh) into the summaries committed
under profiles/.

    python scripts/make_profiles.py <round tag> gpurun_out/<tag>prof

Sub-directories of the raw output (each one rocprofv3 run; --pmc never combined with a trace domain other than the
kernel trace, separate passes per counter set as MI355X_MICROARCH.md prescribes):
  stats, stats_alone        rocprofv3 --kernel-trace --stats: the driver's bench command / the same with --depth 1
  FETCH_SIZE, WRITE_SIZE    HBM traffic per kernel (--depth 1: the counters are device-wide)
  FP64                      SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64: executed float64 flop per launch
  FP32                      SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 (wave-instructions, a packed v_pk_* counted once) and
                            SQ_INSTS_VALU_FLOPS_FP32 (operations per lane, a packed instruction counted twice, an FMA twice): executed
                            float32 flop per launch = 64 x FLOPS_FP32
  RDREQ                     TCC_EA0_RDREQ / _32B: read requests by size (settles what FETCH_SIZE means for 8-byte loads)
  palmer_*                  the direct kernel k_ct_palmer alone (scripts/dev/ct_time.py with CT_FFT=0)
  cfg2_*                    BASELINE cfg2's size (bench.py --workload cfg2)
Writes profiles/<tag>_bench_cfg3_kernel_stats.csv, _alone_kernel_stats.csv, _hbm_counters.json (bench.py reads the latest of
these), <tag>_ct_palmer_counters.json, <tag>_bench_cfg2_kernel_stats.csv, <tag>_bench_cfg2_counters.json.
"""
import glob
import json
import os
import re
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)


def counters(d, counter):
    """mean per launch of one counter, per kernel (summed over the dimensions rocprofv3 splits it into)"""
    fs = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not fs:
        return None
    t = pd.read_csv(max(fs, key=os.path.getmtime))       # a directory that was used twice holds two runs: the newest
    t = t[t.Counter_Name == counter]
    if t.empty:
        return None
    t = t.assign(k=t.Kernel_Name.map(short))
    per_launch = t.groupby(['k', 'Dispatch_Id']).Counter_Value.sum().reset_index()
    return per_launch.groupby('k').Counter_Value.agg(['mean', 'count'])


def stats_csv(d, dst):
    fs = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)
    if not fs:
        return None
    st = pd.read_csv(max(fs, key=os.path.getmtime))
    st.to_csv(dst, index=False)
    return st


def skip(k):
    return k.startswith('__amd') or k.startswith('at::') or k.startswith('void at::')


def hbm_json(raw, prefix, command):
    fe, wr = counters(os.path.join(raw, prefix + 'FETCH_SIZE' if prefix == '' else prefix + 'FETCH'), 'FETCH_SIZE'), \
        counters(os.path.join(raw, prefix + 'WRITE_SIZE' if prefix == '' else prefix + 'WRITE'), 'WRITE_SIZE')
    if fe is None and wr is None:
        return None
    out = {'command': command,
           'note': 'units KB; corrected bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md section HBM '
                   '(gfx950 reports half of the streamed read bytes of wide loads; see the RDREQ note for 8-byte loads)',
           'kernels': {}}
    keys = sorted(set(fe.index if fe is not None else []) | set(wr.index if wr is not None else []))
    for k in keys:
        if skip(k):
            continue
        fk = float(fe.loc[k, 'mean']) if fe is not None and k in fe.index else 0.0
        wk = float(wr.loc[k, 'mean']) if wr is not None and k in wr.index else 0.0
        out['kernels'][k] = {'FETCH_SIZE_KB_mean_per_launch': fk, 'launches_FETCH_SIZE': int(fe.loc[k, 'count']) if fe is not None and k in fe.index else 0,
                             'WRITE_SIZE_KB_mean_per_launch': wk, 'launches_WRITE_SIZE': int(wr.loc[k, 'count']) if wr is not None and k in wr.index else 0,
                             'hbm_bytes_corrected': (2 * fk + wk) * 1024}
    dfp = os.path.join(raw, prefix + 'FP64')
    cnt = {c: counters(dfp, 'SQ_INSTS_VALU_%s_F64' % c) for c in ('ADD', 'MUL', 'FMA', 'TRANS')} if os.path.isdir(dfp) else {}
    if cnt and all(v is not None for v in cnt.values()):
        out['fp64_note'] = ('fp64_flop_per_launch = 64 x (ADD_F64 + MUL_F64 + 2 FMA_F64 + TRANS_F64) wave-instructions per launch '
                            '(SQ_INSTS_VALU_*_F64, full exec mask assumed; exp / division expand into these)')
        for k in out['kernels']:
            if all(k in cnt[c].index for c in cnt):
                w = {c: float(cnt[c].loc[k, 'mean']) for c in cnt}
                out['kernels'][k]['fp64_wave_instructions'] = w
                out['kernels'][k]['fp64_flop_per_launch'] = 64.0 * (w['ADD'] + w['MUL'] + 2.0 * w['FMA'] + w['TRANS'])
    dfp = os.path.join(raw, prefix + 'FP32')
    c32 = {c: counters(dfp, c) for c in ('SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_FLOPS_FP32')} \
        if os.path.isdir(dfp) else {}
    if c32 and all(v is not None for v in c32.values()):
        out['fp32_note'] = ('fp32_flop_per_launch = 64 x SQ_INSTS_VALU_FLOPS_FP32 (the counter tallies float32 operations per lane: 2 for an FMA, '
                            'twice that for a packed v_pk_*_f32; checked against ADD + MUL + 2 FMA wave-instructions: x 2 for a kernel whose '
                            'float32 arithmetic is packed); fp32_wave_instructions: a packed instruction counted once')
        for k in out['kernels']:
            if all(k in c32[c].index for c in c32):
                w = {c.replace('SQ_INSTS_VALU_', ''): float(c32[c].loc[k, 'mean']) for c in c32}
                if w['FLOPS_FP32'] > 0:
                    out['kernels'][k]['fp32_wave_instructions'] = {c: w[c] for c in ('ADD_F32', 'MUL_F32', 'FMA_F32')}
                    out['kernels'][k]['fp32_flop_per_launch'] = 64.0 * w['FLOPS_FP32']
    return out


def issue_json(raw, build_id):
    """Issue-side SQ counters (passes ISSUE1..3 of profile_round.sh), mean per launch and kernel.  SQ_* cycle counters are in
    quad-cycles summed over the waves / SIMDs that counted them (MI355X_MICROARCH.md, cycle constants)."""
    out = {}
    names = {}
    for p in ('ISSUE1', 'ISSUE2', 'ISSUE3', 'ISSUE4', 'ISSUE5', 'ISSUE6'):
        fs = glob.glob(os.path.join(raw, p, '**', '*counter_collection.csv'), recursive=True)
        if not fs:
            continue
        t = pd.read_csv(max(fs, key=os.path.getmtime))
        for c in sorted(set(t.Counter_Name)):
            r = counters(os.path.join(raw, p), c)
            if r is None:
                continue
            for k in r.index:
                if skip(k):
                    continue
                out.setdefault(k, {})[c] = float(r.loc[k, 'mean'])
                names[c] = 1
    if not out:
        return None
    for k, v in out.items():
        wc, busy = v.get('SQ_WAVE_CYCLES'), v.get('SQ_BUSY_CYCLES')
        if wc:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM',
                      'SQ_ACTIVE_INST_SCA', 'SQ_WAIT_INST_LDS'):
                if c in v:
                    v[c + '_per_WAVE_CYCLES'] = v[c] / wc
        if busy and 'SQ_ACTIVE_INST_VALU' in v:
            # SQ_BUSY_CYCLES counts per SE-level SQ; VALU activity summed over waves: busy fraction of the SIMDs = active / (4 SIMDs x CU-busy)
            if v.get('SQ_BUSY_CU_CYCLES'):
                v['VALU_busy_of_SIMD_time'] = v['SQ_ACTIVE_INST_VALU'] / (4.0 * v['SQ_BUSY_CU_CYCLES'])
    return {'build_id': build_id,
            'command': 'rocprofv3 --pmc <8 SQ counters per pass> --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --depth 1 ... (scripts/profile_round.sh, passes ISSUE1-3)',
            'note': 'mean per launch; *_per_WAVE_CYCLES = share of the waves\' lifetime (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES); '
                    'VALU_busy_of_SIMD_time = SQ_ACTIVE_INST_VALU / (4 x SQ_BUSY_CU_CYCLES)',
            'kernels': out}


def ceiling_file(tag, build_id, iss, hbm, alone):
    """profiles/<tag>_ct_rfft_issue.json: the counters of k_ct_rfft the round-3 review asked for under this name, with the kernel's
    distance from its issue-limited ceiling worked out from them (a float64 instruction holds a SIMD's vector pipe for 4 cycles,
    any other VALU instruction for 2; the two-wave ceiling is the probe's, profiles/r*_fp64_issue_rate.txt)."""
    key = next((k for k in iss['kernels'] if k.startswith('k_ct_rfft<')), None)
    if key is None or hbm is None or alone is None:
        return
    v = iss['kernels'][key]
    h = hbm['kernels'].get(key, {})
    row = alone[alone.Name.str.contains('k_ct_rfft<', regex=False)]
    if row.empty or 'fp64_flop_per_launch' not in h:
        return
    dur_us = float(row.AverageNs.iloc[0]) / 1e3
    fp64 = float(sum(h.get('fp64_wave_instructions', {}).values()))
    valu = v['SQ_INSTS_VALU']
    simds = 1024.0
    clk_ghz = v['GRBM_GUI_ACTIVE'] / 8.0 / dur_us / 1e3 if v.get('GRBM_GUI_ACTIVE') else None
    pipe_cycles = (4.0 * fp64 + 2.0 * (valu - fp64)) / simds
    probe = {}
    files = sorted(glob.glob(os.path.join(PROF, 'r*_fp64_issue_rate.txt')))
    if files:                                     # same reading as bench.py:fp64_issue_probe (20 000 x 16 instructions per wave)
        acc = {}
        for line in open(files[-1]):
            m = re.match(r'(\d+) wave\(s\) per SIMD\s+(v_\w+_f64)\s+.*kernel ([\d.]+) ms', line)
            if m:
                acc.setdefault(int(m.group(1)), []).append(int(m.group(1)) * 20000 * 16 / (float(m.group(3)) * 1e3))
        probe = {k: sum(x) / len(x) for k, x in acc.items()}
    d = {'build_id': build_id, 'kernel': key,
         'command': 'scripts/profile_round.sh passes ISSUE1-3 + FP64 (rocprofv3 --pmc, bench.py --depth 1) and stats_alone; scripts/make_profiles.py',
         'per_launch': {c: v[c] for c in ('SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_INST_CYCLES_VMEM_RD', 'SQ_INST_CYCLES_VMEM_WR', 'SQ_WAIT_INST_LDS',
                                          'SQ_ACTIVE_INST_LDS', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
                                          'SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES', 'SQ_WAVES', 'GRBM_GUI_ACTIVE') if c in v},
         'fp64_wave_instructions': fp64, 'other_valu_wave_instructions': valu - fp64, 'duration_alone_us': dur_us,
         'clock_GHz_under_this_kernel': clk_ghz,
         'vector_pipe_cycles_per_simd': pipe_cycles,
         'launch_cycles_per_simd': v['GRBM_GUI_ACTIVE'] / 8.0 if v.get('GRBM_GUI_ACTIVE') else None,
         'fp64_instr_per_us_per_simd': fp64 / simds / dur_us,
         'waves_per_simd': 2}
    if d['launch_cycles_per_simd']:
        d['frac_of_nominal_issue_rate'] = pipe_cycles / d['launch_cycles_per_simd']
    if 2 in probe:
        d['two_wave_ceiling_fp64_instr_per_us_per_simd'] = probe[2]
        # the other VALU instructions take half a float64 slot each
        d['frac_of_two_wave_ceiling'] = (fp64 + 0.5 * (valu - fp64)) / simds / dur_us / probe[2]
        d['fp64_only_frac_of_two_wave_ceiling'] = d['fp64_instr_per_us_per_simd'] / probe[2]
    f2 = d.get('frac_of_two_wave_ceiling')
    if f2 is not None:
        d['verdict'] = ('%.0f %% of what two waves per SIMD can issue: %s the 85 %% the round-3 review set as the bar for "only an instruction diet is left"'
                        % (100.0 * f2, 'below' if f2 < 0.85 else 'at or above'))
    with open(os.path.join(PROF, '%s_ct_rfft_issue.json' % tag), 'w') as fp:
        json.dump(d, fp, indent=1)
    print('ct_rfft issue:', {k: (round(x, 3) if isinstance(x, float) else x) for k, x in d.items() if k.startswith('frac') or k.startswith('fp64_')})


def ct32_file(tag, build_id, iss, hbm, alone):
    """profiles/<tag>_ct_rfft32_issue.json: where the float32 C(t) kernel stands -- executed float32 flop (PMC) over its duration alone
    against the FP32 vector peak, the split of its waves' lifetime, how busy the vector pipe and the LDS are, the occupancy it reaches.
    A packed instruction holds a SIMD's vector pipe for about 3.5 cycles, a plain one for 2 (profiles/r05_pk_issue_rate.txt)."""
    key = next((k for k in iss['kernels'] if k.startswith('k_ct_rfft32')), None)
    if key is None or hbm is None or alone is None:
        return
    v, h = iss['kernels'][key], hbm['kernels'].get(key, {})
    row = alone[alone.Name.str.contains('k_ct_rfft32')]
    if row.empty or 'fp32_flop_per_launch' not in h:
        return
    dur_us = float(row.AverageNs.iloc[0]) / 1e3
    pk = float(sum(h['fp32_wave_instructions'].values()))
    valu = v['SQ_INSTS_VALU']
    clk = v['GRBM_GUI_ACTIVE'] / 8.0 / dur_us / 1e3 if v.get('GRBM_GUI_ACTIVE') else None
    cyc = v['GRBM_GUI_ACTIVE'] / 8.0 if v.get('GRBM_GUI_ACTIVE') else None
    d = {'build_id': build_id, 'kernel': key,
         'command': 'scripts/profile_round.sh passes ISSUE1-6 + FP32 (rocprofv3 --pmc, bench.py --depth 1) and stats_alone; scripts/make_profiles.py',
         'per_launch': {c: v[c] for c in sorted(v) if not c.endswith('_per_WAVE_CYCLES')},
         'fp32_wave_instructions': h['fp32_wave_instructions'], 'fp32_flop_per_launch': h['fp32_flop_per_launch'],
         'duration_alone_us': dur_us, 'clock_GHz_under_this_kernel': clk,
         'TFLOPs_alone': h['fp32_flop_per_launch'] / dur_us / 1e6, 'frac_of_fp32_vector_peak': h['fp32_flop_per_launch'] / dur_us / 1e6 / 157.3,
         'wave_lifetime': {c: v[c + '_per_WAVE_CYCLES'] for c in ('SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY') if c + '_per_WAVE_CYCLES' in v}}
    if cyc:
        d['vector_pipe_busy'] = (3.5 * pk + 2.0 * (valu - pk)) / 1024.0 / cyc
        if v.get('SQ_LDS_IDX_ACTIVE'):
            d['lds_busy'] = v['SQ_LDS_IDX_ACTIVE'] / 256.0 / cyc
            d['lds_bank_conflict_share'] = v.get('SQ_LDS_BANK_CONFLICT', 0.0) / v['SQ_LDS_IDX_ACTIVE']
    with open(os.path.join(PROF, '%s_ct_rfft32_issue.json' % tag), 'w') as fp:
        json.dump(d, fp, indent=1)
    print('ct_rfft32:', {k: (round(x, 3) if isinstance(x, float) else x) for k, x in d.items() if k in ('TFLOPs_alone', 'frac_of_fp32_vector_peak', 'vector_pipe_busy', 'lds_busy', 'duration_alone_us')})


def main():
    tag, raw = sys.argv[1], sys.argv[2]
    try:
        build_id = open(os.path.join(raw, 'build_id.txt')).read().strip()
    except OSError:
        build_id = None
    short_flags = '--no-cpu-baseline --no-kernel-profile --no-cli-wall --spinup-s 0 --repeats 1 --steady-steps 0'
    st = stats_csv(os.path.join(raw, 'stats'), os.path.join(PROF, '%s_bench_cfg3_kernel_stats.csv' % tag))
    sa = stats_csv(os.path.join(raw, 'stats_alone'), os.path.join(PROF, '%s_bench_cfg3_alone_kernel_stats.csv' % tag))
    out = hbm_json(raw, '', 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 --kernel-trace -- python3 bench.py '
                            '--steps 3 --warmup 1 --depth 1 %s (separate passes; --depth 1 = one batch at a time: the counters are device-wide, '
                            'kernels of overlapping batches would be charged to each other)' % short_flags)
    if out is not None:
        out['build_id'] = build_id
        # read requests by size: FETCH_SIZE = 64 B x RDREQ; a request is 32 B (RDREQ_32B) or 64 B, so
        # bytes actually requested = 32 * RDREQ_32B + 64 * (RDREQ - RDREQ_32B)
        rq = counters(os.path.join(raw, 'RDREQ'), 'TCC_EA0_RDREQ_sum')
        rq32 = counters(os.path.join(raw, 'RDREQ'), 'TCC_EA0_RDREQ_32B_sum')
        if rq is not None and rq32 is not None:
            out['rdreq_note'] = ('TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum per launch: read requests to the fabric and how many of them were '
                                 '32-byte ones; request_bytes = 32 * RDREQ_32B + 64 * (RDREQ - RDREQ_32B).  FETCH_SIZE counts every request '
                                 'as 64 B tallied in KB, and a streaming 128-B line arrives as TWO counted... see profiles/README.md')
            for k in out['kernels']:
                if k in rq.index and k in rq32.index:
                    a, b = float(rq.loc[k, 'mean']), float(rq32.loc[k, 'mean'])
                    out['kernels'][k]['TCC_EA0_RDREQ'] = a
                    out['kernels'][k]['TCC_EA0_RDREQ_32B'] = b
                    out['kernels'][k]['rdreq_bytes_32_64'] = 32.0 * b + 64.0 * (a - b)
        with open(os.path.join(PROF, '%s_bench_cfg3_hbm_counters.json' % tag), 'w') as fp:
            json.dump(out, fp, indent=1)
    iss = issue_json(raw, build_id)
    if iss is not None:
        with open(os.path.join(PROF, '%s_issue_counters.json' % tag), 'w') as fp:
            json.dump(iss, fp, indent=1)
        for k in ('k_ct_rfft', 'k_order_search'):      # (k_ct_rfft32 included)
            for kk, v in iss['kernels'].items():
                if kk.startswith(k):
                    print('issue', kk[:40], {c: round(x, 3) for c, x in v.items() if c.endswith('_per_WAVE_CYCLES') or c.startswith('VALU_busy')})
        ceiling_file(tag, build_id, iss, out, sa)
        ct32_file(tag, build_id, iss, out, sa)
    # the direct kernel alone
    pal = {}
    for name, ctrs in (('palmer_FETCH', ['FETCH_SIZE']), ('palmer_WRITE', ['WRITE_SIZE']),
                       ('palmer_VALU', ['SQ_INSTS_VALU', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_ADD_F32'])):
        for c in ctrs:
            r = counters(os.path.join(raw, name), c)
            if r is None:
                continue
            for k in r.index:
                if k.startswith('k_ct_palmer'):
                    pal.setdefault(k, {})[c + '_mean_per_launch'] = float(r.loc[k, 'mean'])
    for k, v in pal.items():
        if 'FETCH_SIZE_mean_per_launch' in v and 'WRITE_SIZE_mean_per_launch' in v:
            v['hbm_bytes_corrected'] = (2 * v['FETCH_SIZE_mean_per_launch'] + v['WRITE_SIZE_mean_per_launch']) * 1024
        if 'SQ_INSTS_VALU_FMA_F32_mean_per_launch' in v:
            v['fp32_flop_per_launch'] = 64.0 * (2 * v['SQ_INSTS_VALU_FMA_F32_mean_per_launch'] + v.get('SQ_INSTS_VALU_MUL_F32_mean_per_launch', 0.0) +
                                               v.get('SQ_INSTS_VALU_ADD_F32_mean_per_launch', 0.0))
    if pal:
        with open(os.path.join(PROF, '%s_ct_palmer_counters.json' % tag), 'w') as fp:
            json.dump({'build_id': build_id,
                       'command': 'CT_FFT=0 rocprofv3 --pmc <one counter set per pass> --kernel-trace -- python3 scripts/dev/ct_time.py '
                                  '(the direct kernel alone on the cfg3 planes: 24 chunks x 4096 frames x 512 vectors)',
                       'kernels': pal}, fp, indent=1)
    # cfg2
    s2 = stats_csv(os.path.join(raw, 'cfg2_stats'), os.path.join(PROF, '%s_bench_cfg2_kernel_stats.csv' % tag))
    o2 = hbm_json(raw, 'cfg2_', 'rocprofv3 --pmc ... --kernel-trace -- python3 bench.py --workload cfg2 --steps 3 --warmup 1 --depth 1 %s' % short_flags)
    if o2 is not None:
        o2['build_id'] = build_id
        with open(os.path.join(PROF, '%s_bench_cfg2_counters.json' % tag), 'w') as fp:
            json.dump(o2, fp, indent=1)
    for label, t in (('in pipeline', st), ('alone (--depth 1)', sa), ('cfg2', s2)):
        if t is not None:
            print('---', label)
            print(t[['Name', 'Calls', 'AverageNs', 'MinNs', 'Percentage']].head(12).to_string())
    if out is not None:
        print({k: round(v['hbm_bytes_corrected'] / 1e6, 1) for k, v in out['kernels'].items()})
        print({k: '%.3g' % v['fp64_flop_per_launch'] for k, v in out['kernels'].items() if 'fp64_flop_per_launch' in v})
    print(pal)


if __name__ == '__main__':
    main()
