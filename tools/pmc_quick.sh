#!/bin/bash
# Runs on the GPU box: a few PMC counters of a bench workload, summed / averaged per kernel.  usage: tools/pmc_quick.sh <out dir> "<counters>" <bench args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/$1; CTRS=$2; shift 2
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("ptk::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    if k.startswith("k_"):
        print(k, {c: ("%.3g" % (acc[k][c] / cnt[k][c])) for c in sorted(acc[k])}, "launches", max(cnt[k].values()))
PY
