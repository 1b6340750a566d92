#!/bin/bash
OUT=gpurun_out/r5q_evict3.txt; : > $OUT
G2F="--scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120"
run() { echo "== $1 :: ${*:2}" >> $OUT; env ${1//,/ } python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "${@:2}" 2>>$OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['roofline']['kernels']
print('%.1f Msamples/s  ' % d['value'] + '  '.join('%s %.0f' % (n, v['avg_us']) for n, v in k.items()))" >> $OUT; }
for rep in 1 2; do for ev in 1 24 32 40 48 56; do run PT_AMD_TOP_EVICT_BELOW=$ev $G2F; done; done
run X=1 --scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120
cat $OUT
