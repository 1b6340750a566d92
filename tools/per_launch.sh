#!/bin/bash
# Runs on the GPU box: every launch of one bench step, in order, with its duration (the per-bounce view the --stats summary averages away).
# usage: tools/per_launch.sh <out file> <bench args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/$1; shift
D=/tmp/per_launch_$$; mkdir -p $D; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 "$@" > $D/bench.log 2>&1
python3 - > $OUT <<PY
import csv, glob
rows = []
for f in glob.glob("$D/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ptk::", "")))
rows.sort()
# the last step: from the last k_generate on
last = max(i for i, r in enumerate(rows) if r[2].startswith("k_generate"))
for s, e, n in rows[last:]:
    print("%9.1f us  %s" % ((e - s) / 1e3, n[:90]))
PY
rm -rf $D
