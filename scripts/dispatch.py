#!/usr/bin/env python3
"""
Dispatcher for running the reference's run-all.bash unchanged on top of the MI355X scripts:

    run-all.bash ... -pycmd "python /path/to/scripts/dispatch.py"

run-all.bash calls `$pycmd $script_loc/<name>.py <args>` (run-all.bash:476, 488, 510, 537); if a same-named
script exists next to this file it is run instead, otherwise the reference's own script is run (Step 1-2
scripts such as calculate-dq-distribution.py are not part of the GPU hot path).
"""
import os
import runpy
import sys

if len(sys.argv) < 2:
    print("usage: dispatch.py <script path> [args...]", file=sys.stderr)
    sys.exit(1)
here = os.path.dirname(os.path.abspath(__file__))
mine = os.path.join(here, os.path.basename(sys.argv[1]))
target = mine if os.path.isfile(mine) and os.path.basename(mine) != 'dispatch.py' else sys.argv[1]
sys.argv = [target] + sys.argv[2:]
runpy.run_path(target, run_name='__main__')
