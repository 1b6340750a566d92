#!/bin/bash
# Runs on the GPU box: the staging mode of the scene blob (whole blob in LDS / core section only) against the occupancy it leaves, on C2 and C3.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$1: %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"; }
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 --steps 3 --warmup 1 --cpu-seconds 0"
python bench.py $C3 2>/dev/null | line "C3 whole blob (65 KB), dynamic units"
PT_AMD_PARK_DYNAMIC=0 python bench.py $C3 2>/dev/null | line "C3 whole blob, static"
PT_AMD_LDS_ALL_LIMIT=32768 python bench.py $C3 2>/dev/null | line "C3 core only (6.7 KB), static"
PT_AMD_LDS_ALL_LIMIT=32768 PT_AMD_PARK_DYNAMIC=1 python bench.py $C3 2>/dev/null | line "C3 core only, dynamic units"
for b in 4 8 16; do PT_AMD_LDS_ALL_LIMIT=32768 PT_AMD_PARK_DYNAMIC=1 PT_AMD_PARK_BLOCKS_PER_CU=$b python bench.py $C3 2>/dev/null | line "C3 core only, dynamic units [$b]"; done
C2="--steps 3 --warmup 1 --cpu-seconds 0 --spp-per-step 240"
python bench.py $C2 2>/dev/null | line "C2 whole blob (14 KB)"
PT_AMD_LDS_ALL_LIMIT=8192 python bench.py $C2 2>/dev/null | line "C2 core only (7 KB)"
