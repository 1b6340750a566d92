#!/bin/bash
OUT=gpurun_out/r5d_x1.txt; : > $OUT
G2F="--scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120"
run() { echo "== $1 :: ${*:2}" >> $OUT; env $1 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "${@:2}" 2>>$OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['roofline']['kernels']
print('%.1f Msamples/s  ' % d['value'] + '  '.join('%s %.0f us (%.1fM)' % (n, v['avg_us'], v['items_per_launch']/1e6) for n, v in k.items()))" >> $OUT; }
run X=1 $G2F
run PT_AMD_X_PARK_MESHLESS=1 $G2F
run PT_AMD_X_PARK_MESHLESS=1 $G2F --env-sampling-probability 0.0
run PT_AMD_X_PARK_MESHLESS=1 --scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120
PT_AMD_X_PARK_MESHLESS=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bokeh" 2>&1 | tail -3 >> $OUT
cat $OUT; tail -5 $OUT.err
