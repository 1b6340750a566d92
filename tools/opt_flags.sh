#!/bin/bash
# Runs on the GPU box: A/B of the optimisation level of the engine on C2..C5, alternating builds (experiment).
source "$(dirname "$0")/lib_build.sh"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
build() { (cd rust-pathtracer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 $1 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -w -shared -o libptamd.so pt_engine.hip pt_output.hip pt_compare.hip pt_scene_host.cpp pt_plan.cpp) > /dev/null 2>&1 || { echo "BUILD FAILED: $1"; exit 1; }; }
one() { timeout 200 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('   %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"; }
for round in 1 2; do for flags in "-O3" "-O2"; do
  build "$flags"; echo "== $flags (round $round)"
  one
  one --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60
  one --scene hdri_test --max-bounces 4 --light-samples 6
  one --hero 4 --spp-per-step 60
done; done
build "-O2"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
pt_make -C rust-pathtracer_amd/csrc clean; pt_make -C rust-pathtracer_amd/csrc all
