#!/bin/bash
OUT=gpurun_out/r5w_group_evict.txt; : > $OUT
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
run() { echo "== $1 :: ${*:2}" >> $OUT; env ${1//,/ } python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "${@:2}" 2>>$OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['roofline']['kernels']
print('%.1f Msamples/s  ' % d['value'] + '  '.join('%s %.0f' % (n, v['avg_us']) for n, v in k.items()))" >> $OUT; }
for rep in 1 2; do for ev in 1 8 16 32 48; do run PT_AMD_GROUP_EVICT_BELOW=$ev $C3; done; done
cat $OUT
