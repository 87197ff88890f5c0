#!/usr/bin/env python3
"""Developer tool: highest VGPR index referenced in windows of a kernel's assembly (where the register pressure peaks).
usage: vgpr_map.py file.s kernel-substring [window]"""
import re
import sys

fn, key = sys.argv[1], sys.argv[2]
win = int(sys.argv[3]) if len(sys.argv) > 3 else 60
inside = False
rows = []
for line in open(fn):
    if re.match(r'^_Z\S*:', line):
        inside = key in line
        continue
    if not inside:
        continue
    t = line.strip()
    if t.startswith('.Lfunc_end'):
        break
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    regs = [int(x) for x in re.findall(r'\bv(\d+)\b', t)]
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', t):
        regs.append(int(b))
    rows.append((max(regs) if regs else -1, t.split()[0]))
for i in range(0, len(rows), win):
    chunk = rows[i:i + win]
    mx = max(r for r, _ in chunk)
    marks = ' '.join(sorted({o for _, o in chunk if o in ('s_barrier', 's_cbranch_scc1', 's_branch')}))
    print('%5d  max v%-4d %s' % (i, mx, marks))
