#!/usr/bin/env python3
"""Developer tool: where a kernel's scratch traffic sits.  usage: spill_map.py file.s kernel-substring
Prints, per region between s_barrier instructions, the number of instructions, scratch stores / loads, and loop labels."""
import re
import sys

fn, key = sys.argv[1], sys.argv[2]
inside = False
region, n, st, ld, lab = 0, 0, 0, 0, []
for line in open(fn):
    if re.match(r'^_Z\S*:', line):
        inside = key in line
        continue
    if not inside:
        continue
    t = line.strip()
    if t.startswith('.Lfunc_end'):
        break
    if t.startswith('.LBB'):
        if 'Loop' in t:
            lab.append(t.split(':')[0] + '(loop)')
        continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    n += 1
    if 'scratch_store' in t:
        st += 1
    if 'scratch_load' in t:
        ld += 1
    if t.startswith('s_barrier'):
        print('region %2d: %5d instr  %3d scratch stores  %3d scratch loads  %s' % (region, n, st, ld, ' '.join(lab)))
        region, n, st, ld, lab = region + 1, 0, 0, 0, []
print('region %2d: %5d instr  %3d scratch stores  %3d scratch loads  %s' % (region, n, st, ld, ' '.join(lab)))
