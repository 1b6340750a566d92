#!/bin/bash
# Runs on the GPU box.  Round-3 verdict item 3: C3's parked mesh kernels with the whole 65 KB gem blob in LDS, in workgroups of 512 / 1024 threads
# (PT_AMD_PARK_BLOCK), against the default (256 threads, the blob's 6.7 KB core staged, the mesh read through L1/L2) — same box, same run.
one() { timeout 600 python bench.py --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('   %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"; }
for rep in 1 2; do
echo "default (256 threads, core staged)"; one
echo "PT_AMD_PARK_BLOCK=512"; PT_AMD_PARK_BLOCK=512 one
echo "PT_AMD_PARK_BLOCK=1024"; PT_AMD_PARK_BLOCK=1024 one
done
echo "256 threads, whole blob staged, static segments (round 2's losing form: two workgroups per CU)"; PT_AMD_LDS_ALL_LIMIT=65536 PT_AMD_PARK_DYNAMIC=0 one
