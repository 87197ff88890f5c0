#!/usr/bin/env python3
"""Developer tool: build _variants/lib_<name>.so -- the library with extra compiler flags for SOME sources (the other objects
are the regular build's) -- for A/B runs on one box (SPINRELAX_HIP_LIB=_variants/lib_<name>.so python ...).
usage: variants.py <name> "<extra flags>" file.hip [file.hip ...]     (prints VGPR / scratch of the kernels named in $SHOW)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spinrelax_amd import build as B                                    # noqa: E402


def main():
    name, flags, files = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
    B.build(verbose=False)
    out = os.path.join(ROOT, '_variants')
    os.makedirs(out, exist_ok=True)
    objs = []
    for src in B.SOURCES:
        o = os.path.join(B.CSRC, src.replace('.hip', '.o'))
        if src in files:
            o = os.path.join(out, '%s_%s' % (name, src.replace('.hip', '.o')))
            extra = B.EXTRA.get(src, []) + (['-DSR_BUILD_ID="variant-%s"' % name] if src == 'sr_core.hip' else [])
            cmd = [B.HIPCC] + B.FLAGS + extra + flags + ['-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(B.CSRC, src), '-o', o]
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if p.returncode:
                print(p.stdout[-3000:])
                sys.exit(1)
            show = os.environ.get('SHOW', '')
            cur = None
            for l in p.stdout.splitlines():
                if 'Function Name:' in l:
                    cur = l.split('Function Name:')[1].strip()
                if show and cur and show in cur and any(k in l for k in (' VGPRs:', 'ScratchSize', 'Occupancy')):
                    print('   ', cur[:60], l.split('remark:')[-1].strip())
        objs.append(o)
    lib = os.path.join(out, 'lib_%s.so' % name)
    subprocess.check_call([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    print(lib)


if __name__ == '__main__':
    main()
