#!/usr/bin/env python3
"""
Merge the tallies a GPU test run dumped (SR_DUMP_TALLIES=<dir> python -m pytest tests -m gpu) into
tests/golden/fit_trial_tallies.json -- the committed counts the parity tests hold the kernels to
(tests/conftest.py:committed_tally).  Run after a DELIBERATE change of the fit kernels' arithmetic, look at the diff,
commit it with the change.

    gpurun -- 'SR_DUMP_TALLIES=gpurun_out/tallies python -m pytest tests -m gpu -q'
    python scripts/collect_tallies.py gpurun_out/tallies
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, 'tests', 'golden', 'fit_trial_tallies.json')


def main():
    src = sys.argv[1]
    try:
        with open(DST) as fp:
            out = json.load(fp)
    except OSError:
        out = {}
    n = 0
    for fn in sorted(glob.glob(os.path.join(src, 'tally__*__*.json'))):
        _, section, key = os.path.basename(fn)[:-5].split('__')
        with open(fn) as fp:
            new = json.load(fp)
        old = out.get(section, {}).get(key)
        if old != new:
            print('%s / %s: %s -> %s' % (section, key, old, new))
        out.setdefault(section, {})[key] = new
        n += 1
    out['_note'] = ('counts measured on MI355X by the GPU tests (SR_DUMP_TALLIES) and merged by scripts/collect_tallies.py; '
                    'the tests assert them as floors / ceilings (tests/conftest.py:committed_tally)')
    with open(DST, 'w') as fp:
        json.dump(out, fp, indent=1, sort_keys=True)
    print('%d entries merged into %s' % (n, os.path.relpath(DST, ROOT)))


if __name__ == '__main__':
    main()
