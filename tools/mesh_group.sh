#!/bin/bash
# Runs on the GPU box: C3 with different group sizes of the mesh sweep (PT_MESH_GROUP leaves per group box; pt_blob.h).
source "$(dirname "$0")/lib_build.sh"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for g in ${@:-8 16 32}; do
  pt_make -C rust-pathtracer_amd/csrc clean; rm -f tests/host_emulation/libptemu.so
  pt_make -j8 -C rust-pathtracer_amd/csrc EXTRA="-DPT_MESH_GROUP=$g" all
  python bench.py --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('group $g: %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"
done
pt_make -C rust-pathtracer_amd/csrc clean; pt_make -j8 -C rust-pathtracer_amd/csrc all
