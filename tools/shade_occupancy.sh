#!/bin/bash
# Runs on the GPU box: occupancy of the k_shade forms without the environment branch (C2: NL = 1, C5: NL = 4) — experiment.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
build() { (cd rust-pathtracer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 $1 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -w -shared -o libptamd.so pt_engine.hip pt_output.hip pt_compare.hip pt_scene_host.cpp pt_plan.cpp) > /dev/null 2>&1; }
one() { timeout 200 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('   %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"; }
for flags in "-DPT_SHADE_LEAN_WAVES=5 -DPT_SHADE4_LEAN_WAVES=5" "-DPT_SHADE_LEAN_WAVES=4 -DPT_SHADE4_LEAN_WAVES=4" "-DPT_SHADE_LEAN_WAVES=2 -DPT_SHADE4_LEAN_WAVES=2"; do
  build "$flags"; echo "== $flags"
  one; one --hero 4 --spp-per-step 60
done
