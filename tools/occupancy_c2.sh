#!/bin/bash
# Runs on the GPU box: C2 with different register budgets of the sweep kernels (waves per SIMD) and park kernels (experiment; -j8 rebuilds).
source "$(dirname "$0")/lib_build.sh"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for w in ${@:-4 5 6}; do
  touch rust-pathtracer_amd/csrc/pt_kern_extend.hip rust-pathtracer_amd/csrc/pt_kern_shadow.hip
  pt_make -j8 -C rust-pathtracer_amd/csrc EXTRA="-DPT_SWEEP_WAVES=$w" libptamd.so
  python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --spp-per-step 120 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('sweep waves $w: %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"
done
touch rust-pathtracer_amd/csrc/pt_kern_extend.hip rust-pathtracer_amd/csrc/pt_kern_shadow.hip
pt_make -j8 -C rust-pathtracer_amd/csrc libptamd.so
