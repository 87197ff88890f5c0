#!/bin/bash
# usage: scripts/dev/mkvariant.sh <name> "<extra hipcc flags>" file.hip [file.hip ...]
# Builds _variants/lib_<name>.so: the named sources recompiled with the extra flags, every other object taken from the
# in-tree build (run `python -m spinrelax_amd.build` first).  _variants/ is git-ignored and travels with gpurun.
set -e
name=$1; flags=$2; shift 2
root=$(cd $(dirname $0)/../.. && pwd)
c=$root/spinrelax_amd/csrc
mkdir -p $root/_variants/obj_$name
objs=""
for src in sr_core sr_ct sr_vechist sr_fit sr_relax sr_dq sr_traj sr_vectors sr_textio; do
  o=$c/$src.o
  for f in "$@"; do
    if [ "$f" = "$src.hip" ]; then
      o=$root/_variants/obj_$name/$src.o
      extra=""
      [ $src = sr_ct ] && extra="-fno-slp-vectorize"
      [ $src = sr_fit ] && extra="-ffp-contract=off"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function $extra $flags -c $c/$src.hip -o $o
    fi
  done
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/_variants/lib_$name.so $objs
echo $root/_variants/lib_$name.so
