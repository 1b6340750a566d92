#!/bin/bash
# Runs on the GPU box: one fuzz seed under each diagnostic switch of the engine (which traversal / staging form does a mismatch need?).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for sw in NONE PT_AMD_NO_LDS PT_AMD_NO_CORE_LDS PT_AMD_NO_SWEEP PT_AMD_NO_MESH_SWEEP PT_AMD_NO_PARK PT_AMD_EXACT_SLAB PT_AMD_NO_CULL; do
  echo "== $sw"; env $sw=1 timeout 300 python tools/fuzz_seed.py $1 2>&1 | grep -E "valid equal|^\("
done
