#!/bin/bash
OUT=gpurun_out/r5x_final_check.txt; : > $OUT
run() { echo "== $1 :: ${*:2}" >> $OUT; env ${1//,/ } python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "${@:2}" 2>>$OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['roofline']['kernels']
print('%.1f Msamples/s  ' % d['value'] + '  '.join('%s %.0f' % (n, v['avg_us']) for n, v in k.items()))" >> $OUT; }
run X=1 --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60
run X=1 --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120
run X=1 --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120
cat $OUT
