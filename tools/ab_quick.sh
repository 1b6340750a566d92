#!/bin/bash
# Runs on the GPU box: the four BASELINE configurations (short) and the GPU suite — the quick check after a kernel change.
one() { timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('   %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"; }
one; one --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60; one --scene hdri_test --max-bounces 4 --light-samples 6; one --hero 4 --spp-per-step 60
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
