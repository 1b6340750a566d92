#!/bin/bash
# what the per-launch HIP events cost a short frame: G2 / G1 / C2 with and without them (PT_AMD_STAGE_TIMING=0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_x26.txt; cd $ROOT
one() { timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().replace('PT_BENCH_RECORD ',''))
print('   %.1f Ms/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
G2="--scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120"
G1="--scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120"
C2="--spp-per-step 120"
for r in 1 2; do for t in 1 0; do echo "== PT_AMD_STAGE_TIMING=$t"; for w in "$G2" "$G1" "$C2"; do PT_AMD_STAGE_TIMING=$t one $w; done; done; done > $OUT 2>&1
cat $OUT
