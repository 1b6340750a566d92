#!/bin/bash
# C4 k_shade: L2 (TCC) requests, hits, misses per launch; and the same with the importance map at 256 x 256 (tables 16 x smaller: what the kernel costs when its tables fit L2)
OUT=gpurun_out/r5k_c4_tcc.txt; : > $OUT
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
bash tools/pmc_quick.sh gpurun_out/r5k_pmc_c4 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" $C4 >> $OUT 2>&1
python bench.py --steps 3 --warmup 1 --cpu-seconds 0 $C4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['roofline']['kernels']
print('C4 %.1f Msamples/s  ' % d['value'] + '  '.join('%s %.0f' % (n, v['avg_us']) for n, v in k.items()))" >> $OUT
find gpurun_out/r5k_pmc_c4 -name "*.csv" -size +1M -delete
cat $OUT
