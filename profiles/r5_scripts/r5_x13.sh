#!/bin/bash
OUT=gpurun_out/r5s_topsweep.txt; : > $OUT
G2="--scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120"
G2F="--scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120"
run() { echo "== $1 :: ${*:2}" >> $OUT; env ${1//,/ } python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "${@:2}" 2>>$OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['roofline']['kernels']
print('%.1f Msamples/s  ' % d['value'] + '  '.join('%s %.0f' % (n, v['avg_us']) for n, v in k.items()))" >> $OUT; }
for rep in 1 2; do for sw in 1 0 2; do run PT_AMD_TOP_SWEEP=$sw $G2F; done; done
for sw in 1 0 2; do run PT_AMD_TOP_SWEEP=$sw $G2; done
run PT_AMD_TOP_SWEEP=0,PT_AMD_LIGHT_PREPASS_MAX=4294967295 $G2F
cat $OUT
