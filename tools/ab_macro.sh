#!/bin/bash
# Runs on the GPU box: A/B of one compile-time macro on chosen bench workloads, alternating builds twice (experiment helper).
# usage: tools/ab_macro.sh "<flags A>" "<flags B>" -- <bench args...>   (several workloads: separate them with ';;')
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
A="$1"; B="$2"; shift 3
build() { (cd rust-pathtracer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 $1 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -w -shared -o libptamd.so pt_engine.hip pt_output.hip pt_compare.hip pt_scene_host.cpp pt_plan.cpp) > /dev/null 2>&1 || { echo "BUILD FAILED: $1"; exit 1; }; }
one() { timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('   %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"; }
for round in 1 2; do for flags in "$A" "$B"; do
  build "$flags"; echo "== $flags"
  args=(); for a in "$@"; do if [ "$a" == ";;" ]; then one "${args[@]}"; args=(); else args+=("$a"); fi; done; one "${args[@]}"
done; done
