#!/bin/bash
OUT=gpurun_out/r5f_x3.txt; : > $OUT
run() { echo "== $1 :: ${*:2}" >> $OUT; env ${1//,/ } python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "${@:2}" 2>>$OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['roofline']['kernels']
print('%.1f Msamples/s  ' % d['value'] + '  '.join('%s %.0f' % (n, v['avg_us']) for n, v in k.items()))" >> $OUT; }
GEM="--scene test_bokeh_floor_gem --max-bounces 8 --light-samples 2 --spp-per-step 120"
for rep in 1 2; do for ev in 1 0 16 48; do run PT_AMD_TOP_EVICT_BELOW=$ev $GEM; done; done
run X=1 --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120
run X=1 --scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120
cat $OUT
