#!/bin/bash
# A/B of SQ counters for the traversal kernels: fast (filtered slab + cull) vs exact walk.  Runs on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_ab
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
for mode in fast exact; do
  if [ $mode = exact ]; then export PT_AMD_EXACT_SLAB=1 PT_AMD_NO_CULL=1; else unset PT_AMD_EXACT_SLAB PT_AMD_NO_CULL; fi
  timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$OUT/$mode" -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-seconds 0 > "$OUT/$mode.log" 2>&1
done
cd "$ROOT"; python3 - <<'PY'
import csv, glob, collections, re
for mode in ("fast","exact"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("gpurun_out/pmc_ab/%s/**/*counter_collection.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"])
            if not m: continue
            acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": n[m.group(1)] += 1
    for k, v in acc.items():
        print(mode, k, n[k], {a: "%.3g" % (b / max(1, n[k])) for a, b in sorted(v.items())})
PY
find "$OUT" -name "*.csv" -size +5M -delete
