#!/usr/bin/env python3
"""Developer tool: registers / scratch / occupancy of every kernel of the library (kernel_resources.sh per source with the source's
flags from spinrelax_amd/build.py), headed by the build id.   usage: kernel_resources_all.py > profiles/rNN_kernel_resources.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spinrelax_amd import build                     # noqa: E402

print("# registers / scratch / occupancy of every kernel of the library, build id %s" % build.build_id())
print("# (scripts/dev/kernel_resources_all.py: kernel_resources.sh <source> <the source's extra flags>; hipcc -Rpass-analysis=kernel-resource-usage)")
for src in build.SOURCES:
    extra = build.EXTRA.get(src, [])
    print("## %s %s" % (src, ' '.join(extra)))
    r = subprocess.run(['bash', os.path.join(ROOT, 'scripts', 'dev', 'kernel_resources.sh'), os.path.join(build.CSRC, src), '-I' + os.path.join(ROOT, 'include')] + extra,
                       capture_output=True, text=True)
    print(r.stdout.rstrip())
