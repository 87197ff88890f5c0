#!/usr/bin/env python3
"""Developer tool: per-loop instruction mix (scratch traffic, FP64 VALU, LDS, barriers) of one function in a gfx950 .s
file produced with `hipcc -save-temps`.   usage: dev_isa_loops.py file.s <substring of the mangled name>"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
print(lines[start].split(':')[0][:120], '-', len(body), 'lines')
for l in lines[end:end + 14]:
    if 'num_vgpr' in l or 'private_seg' in l or 'num_agpr' in l:
        print('   ', l.split('.')[-1])
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r'\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i))


def count(a, b, pat):
    return sum(1 for l in body[a:b] if re.search(pat, l))


print('whole function: scratch_load %d scratch_store %d f64 %d ds %d' % (count(0, len(body), 'scratch_load'), count(0, len(body), 'scratch_store'),
                                                                       count(0, len(body), r'v_\w+_f64'), count(0, len(body), r'\sds_')))
print('%6s %6s %6s %5s %5s %5s %5s %4s' % ('len', 'start', 'end', 'sld', 'sst', 'f64', 'ds', 'bar'))
for a, b in sorted(loops, key=lambda t: t[0] - t[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print('%6d %6d %6d %5d %5d %5d %5d %4d' % (b - a, a, b, count(a, b, 'scratch_load'), count(a, b, 'scratch_store'), count(a, b, r'v_\w+_f64'),
                                                count(a, b, r'\sds_'), count(a, b, 's_barrier')))
