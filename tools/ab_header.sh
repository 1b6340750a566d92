#!/bin/bash
# Runs on the GPU box: same-box A/B of two versions of pt_device.h (csrc/pt_device_old.h.txt = A, the working tree = B), alternating twice,
# on the four BASELINE configurations — experiment helper (put the old header there with `git show HEAD:.../pt_device.h`).
# The host side (pt_scene_host.cpp) is taken from csrc/pt_scene_host_old.cpp.txt for A when that file exists.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT/rust-pathtracer_amd/csrc
cp pt_device.h /tmp/new.h; cp pt_device_old.h.txt /tmp/old.h
cp pt_scene_host.cpp /tmp/new.cpp; if [ -f pt_scene_host_old.cpp.txt ]; then cp pt_scene_host_old.cpp.txt /tmp/old.cpp; else cp pt_scene_host.cpp /tmp/old.cpp; fi
build() { cp /tmp/$1.h pt_device.h; cp /tmp/$1.cpp pt_scene_host.cpp; /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -w -shared -o libptamd.so pt_engine.hip pt_output.hip pt_compare.hip pt_scene_host.cpp pt_plan.cpp > /dev/null 2>&1; }
one() { (cd $ROOT; timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('   %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"); }
for round in 1 2; do for v in old new; do
  build $v; echo "== $v"
  one; one --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60; one --scene hdri_test --max-bounces 4 --light-samples 6
done; done
cp /tmp/new.h pt_device.h; cp /tmp/new.cpp pt_scene_host.cpp
