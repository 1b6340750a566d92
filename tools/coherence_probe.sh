#!/bin/bash
# Runs on the GPU box: tools/coherence_probe.py under rocprofv3, then the duration of every k_probe_intersect launch.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/coherence; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/coherence_probe.py "$@" > $OUT/probe.log 2>&1
cat $OUT/probe.log | grep -v "^W2\|rocprof" | tail -8
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "k_probe_intersect" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for k, r in enumerate(rows):
    print("launch %d: %.0f us" % (k + 1, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000))
PY
