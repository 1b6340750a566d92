#!/bin/bash
# Runs on the GPU box: batch size (path slots per pass) x workgroups per CU for the C2 workload.
for cfg in "67108864 16 60" "67108864 24 60" "67108864 32 60" "67108864 48 60" "67108864 64 60" "33554432 32 30" "134217728 32 120"; do set -- $cfg
  export PT_AMD_BATCH=$1 PT_AMD_BLOCKS_PER_CU=$2
  timeout 200 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --spp-per-step $3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('batch $1 bpc $2 spp $3: %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"
done
