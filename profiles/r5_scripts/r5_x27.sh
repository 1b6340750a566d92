#!/bin/bash
# one HIP event per launch instead of two: G2 / G1 / C2 records, the default record's kernel times against rocprofv3's
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_x27.txt; cd $ROOT
one() { timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().replace('PT_BENCH_RECORD ',''))
print('   %.1f Ms/s  %.3f ms/step' % (d['value'], d['ms_per_step']), {n: round(v['avg_us'],1) for n,v in d['roofline']['kernels'].items()})"; }
G2="--scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120"
G1="--scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120"
C2="--spp-per-step 120"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "profile or cached or setup or tuning or shard or multi or device" 2>&1 | grep -E "passed|failed|error" > $OUT
for r in 1 2; do for w in "$G2" "$G1" "$C2"; do one $w; done; done >> $OUT 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/x27_prof -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --spp-per-step 120 > $ROOT/gpurun_out/x27_bench.log 2>&1
cd $ROOT; python - <<'PY' >> $OUT
import csv, glob, json
rec = json.loads([l for l in open('gpurun_out/x27_bench.log') if l.startswith('{') or l.startswith('PT_BENCH_RECORD')][-1].replace('PT_BENCH_RECORD ', ''))
print('bench kernels (events):', {n: round(v['avg_us'], 1) for n, v in rec['roofline']['kernels'].items()})
for f in glob.glob('gpurun_out/x27_prof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_' in r['Name']: print('rocprofv3:', r['Name'].split('(')[0][-40:], r['Calls'], round(float(r['AverageNs']) / 1000, 1))
PY
cat $OUT
