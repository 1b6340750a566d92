#!/bin/bash
# Where does G2F's k_shadow spend its time?  Light rays only / environment rays only / both; per-lane walk against the forms of PT_AMD_NO_LDS; SQ counters.
OUT=gpurun_out/r5c_probe.txt; : > $OUT
G2F="--scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120"
run() { echo "== $1 :: ${*:2}" >> $OUT; env $1 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['roofline']['kernels']
print('%.1f Msamples/s  D=%.2f shadow rays/sample %.2f  ' % (d['value'], d['segments_per_sample'], d['rays_per_s']['shadow']/d['rays_per_s']['segments']*d['segments_per_sample']) + '  '.join('%s %.0f us (%.1fM)' % (n, v['avg_us'], v['items_per_launch']/1e6) for n, v in k.items()))" >> $OUT; }
run X=1 $G2F --env-sampling-probability 0.0
run X=1 $G2F --env-sampling-probability 1.0
run X=1 $G2F
run X=1 $G2F --max-bounces 1
run PT_AMD_NO_LDS=1 $G2F
run PT_AMD_NO_CULL=1 $G2F
for p in 0.0 1.0; do
  echo "== counters env prob $p" >> $OUT
  bash tools/pmc_quick.sh gpurun_out/r5c_pmc_$p "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" $G2F --env-sampling-probability $p >> $OUT 2>&1
done
find gpurun_out/r5c_pmc_* -name "*.csv" -size +1M -delete
cat $OUT
