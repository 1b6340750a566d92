#!/bin/bash
# C4: L2 (TCC) hits and misses per launch of k_shade on the final kernels (round-4 verdict item 5: "a logged negative with TCC_HIT/MISS")
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5z_tcc; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 1100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 60 > $OUT/bench.log 2>&1
echo "rc=$?" >> $OUT/bench.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("ptk::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    if k.startswith("k_"):
        print(k, {c: ("%.4g" % (acc[k][c] / cnt[k][c])) for c in sorted(acc[k])}, "launches", max(cnt[k].values()))
PY
tail -3 $OUT/bench.log | cut -c1-300
find $OUT -name "*.csv" -size +1M -delete
