#!/bin/bash
# top-level walk: walk_box against aabb_hit_node, and the eviction threshold — on G2F, G2 and the mesh scenes without their sweep tables
OUT=gpurun_out/r5e_x2.txt; : > $OUT
G2F="--scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120"
G2="--scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120"
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
run() { echo "== $1 :: ${*:2}" >> $OUT; env ${1//,/ } python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "${@:2}" 2>>$OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['roofline']['kernels']
print('%.1f Msamples/s  ' % d['value'] + '  '.join('%s %.0f' % (n, v['avg_us']) for n, v in k.items()))" >> $OUT; }
for rep in 1 2; do
for ev in 1 16 32 48; do
  run PT_AMD_TOP_EVICT_BELOW=$ev $G2F
done
run PT_AMD_TOP_EVICT_BELOW=1,PT_AMD_LIBRARY=$PWD/variants/topbox0.so $G2F
run PT_AMD_TOP_EVICT_BELOW=32,PT_AMD_LIBRARY=$PWD/variants/topbox0.so $G2F
done
for ev in 1 32; do run PT_AMD_TOP_EVICT_BELOW=$ev $G2; done
run PT_AMD_TOP_EVICT_BELOW=1,PT_AMD_LIBRARY=$PWD/variants/topbox0.so $G2
for ev in 1 32 48; do run PT_AMD_TOP_EVICT_BELOW=$ev,PT_AMD_NO_SWEEP=1 $C3; run PT_AMD_TOP_EVICT_BELOW=$ev,PT_AMD_NO_SWEEP=1 $C4; done
run PT_AMD_TOP_EVICT_BELOW=1,PT_AMD_NO_SWEEP=1,PT_AMD_LIBRARY=$PWD/variants/topbox0.so $C3
run PT_AMD_TOP_EVICT_BELOW=1,PT_AMD_NO_SWEEP=1,PT_AMD_LIBRARY=$PWD/variants/topbox0.so $C4
cat $OUT
