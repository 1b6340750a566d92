#!/bin/bash
# Runs on the GPU box: rebuilds the engine with different register budgets for the parked kernels and benches C3 / C4.
source "$(dirname "$0")/lib_build.sh"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for w in "$@"; do
  pt_make -C rust-pathtracer_amd/csrc clean; pt_make -C rust-pathtracer_amd/csrc libptamd.so EXTRA="-DPT_PARK_WAVES=$w"
  for cfg in "C3 --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 30" "C4 --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 30"; do
    set -- $cfg; name=$1; shift
    timeout 200 python bench.py "$@" --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/park_$name.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open("gpurun_out/park_$name.json").read().strip().split("\n")[-1])
print("park_waves $w $name", round(d["value"],1), {k:round(v["avg_us"]) for k,v in d["roofline"]["kernels"].items()})
PY
  done
done
pt_make -C rust-pathtracer_amd/csrc clean; pt_make -C rust-pathtracer_amd/csrc libptamd.so
