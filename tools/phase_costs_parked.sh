#!/bin/bash
# Runs on the GPU box: where does k_shadow_parked spend its time on C4?  Builds the kernel with parts left out (PT_PARKED_EXP bits: 1 the
# listing of live rays only, 2 up to the masks, 4 parked rays dropped = no mesh walks) and benches each; the product build is restored.
source "$(dirname "$0")/lib_build.sh"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for e in 1 2 4 0; do
  touch rust-pathtracer_amd/csrc/pt_kern_shadow.hip
  pt_make -j8 -C rust-pathtracer_amd/csrc EXTRA="-DPT_PARKED_EXP=$e" libptamd.so
  python bench.py --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('variant $e: k_shadow_parked %.0f us' % k['shadow']['avg_us'], {n: round(v['avg_us']) for n,v in k.items()})"
done
touch rust-pathtracer_amd/csrc/pt_kern_shadow.hip; pt_make -j8 -C rust-pathtracer_amd/csrc libptamd.so
