#!/bin/bash
# Runs on the GPU box: register budgets (waves per SIMD) of one form of k_shade on the configuration that uses it — experiment.
# usage: tools/shade_occupancy.sh <macro> "<waves list>" <bench args...>   e.g.  PT_SHADE_NO_ENV_WAVES "2 3 4 5" --scene cornell_gem --width 1920 ...
source "$(dirname "$0")/lib_build.sh"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
MACRO=$1; WAVES=$2; shift 2
for w in $WAVES; do
  touch rust-pathtracer_amd/csrc/pt_kern_shade.hip
  pt_make -j8 -C rust-pathtracer_amd/csrc EXTRA="-D$MACRO=$w" libptamd.so
  timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$MACRO=$w: %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"
done
touch rust-pathtracer_amd/csrc/pt_kern_shade.hip; pt_make -j8 -C rust-pathtracer_amd/csrc libptamd.so
