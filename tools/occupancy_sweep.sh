#!/bin/bash
# Runs on the GPU box: rebuilds the engine with different register budgets and benches C2..C5 with each (experiment).
source "$(dirname "$0")/lib_build.sh"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
show() { python - <<PY
import json
try:
    d=json.loads(open("$1").read().strip().split("\n")[-1])
    print("$2:", round(d["value"],1), {k:round(v["avg_us"]) for k,v in d["roofline"]["kernels"].items()})
except Exception as e: print("$2 failed", e)
PY
}
for cfg in "${@:-3 1 5 1}"; do
  set -- $cfg
  pt_make -C rust-pathtracer_amd/csrc clean
  pt_make -C rust-pathtracer_amd/csrc EXTRA="-DPT_SHADE_WAVES=$1 -DPT_SHADE4_WAVES=$2 -DPT_SWEEP_WAVES=$3 -DPT_WALK_WAVES=$4"
  echo "== shade $1 shade4 $2 sweep $3 walk $4"
  T=gpurun_out/occ_$1_$2_$3_$4
  python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > ${T}_C2.json 2> ${T}_C2.err; show ${T}_C2.json C2
  python bench.py --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 30 --steps 2 --warmup 1 --cpu-seconds 0 > ${T}_C3.json 2> ${T}_C3.err; show ${T}_C3.json C3
  python bench.py --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 30 --steps 2 --warmup 1 --cpu-seconds 0 > ${T}_C4.json 2> ${T}_C4.err; show ${T}_C4.json C4
  python bench.py --hero 4 --spp-per-step 10 --steps 2 --warmup 1 --cpu-seconds 0 > ${T}_C5.json 2> ${T}_C5.err; show ${T}_C5.json C5
done
