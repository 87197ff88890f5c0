#!/bin/bash
# Developer tool: register / scratch / occupancy table of every kernel in one source file.
# usage: scripts/dev/kernel_resources.sh spinrelax_amd/csrc/sr_fit.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -c "$src" -o /tmp/_res_tmp.o \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    if cur is None: continue
    for k in ('VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','SGPRs','LDS Size \[bytes/block\]'):
        m=re.search(r'\s'+k+r': (\d+)',l)
        if m: cur[k.split(' ')[0]]=int(m.group(1))
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=n.replace('(anonymous namespace)::','')[:60]
    print('%-60s VGPR %3d AGPR %3d scratch %5d occ %d LDS %d'%(n,r.get('VGPRs',-1),r.get('AGPRs',-1),r.get('ScratchSize',-1),r.get('Occupancy',-1),r.get('LDS',-1)))
"
