#!/bin/bash
# A/B: the light pre-pass (nearest hit among all lights bounds a light-sample ray) against plain closest-hit searches, on G2F (82 lights) and G1 / C3 (1-2 lights)
OUT=gpurun_out/r5b_prepass.txt; : > $OUT
run() { echo "== $1 :: ${*:2}" >> $OUT; env $1 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['roofline']['kernels']
print('%.1f Msamples/s  ' % d['value'] + '  '.join('%s %.0f us' % (n, v['avg_us']) for n, v in k.items()))" >> $OUT; }
for v in 4294967295 0 1; do
  run PT_AMD_LIGHT_PREPASS_MAX=$v --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120
done
for v in 0 1; do
  run PT_AMD_LIGHT_PREPASS_MAX=$v --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120
done
cat $OUT
