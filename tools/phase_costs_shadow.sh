#!/bin/bash
# Runs on the GPU box: where does k_shadow spend its time?  (The shadow half of tools/phase_costs.sh: measurement variants of the kernel with
# parts left out, launched in front of the real kernel on the same input.)
source "$(dirname "$0")/lib_build.sh"
OUT=${1:-gpurun_out/phase_costs_shadow}; mkdir -p $OUT
cd rust-pathtracer_amd/csrc && touch pt_kern_extend.hip pt_kern_shadow.hip && pt_make -j8 EXTRA=-DPT_EXPERIMENTS libptamd.so; cd ../..
PT_AMD_NO_POOL=1 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/exp_none.json 2> $OUT/exp_none.err
for e in 8 2 4 0; do
  PT_AMD_EXP_SHADOW=$e PT_AMD_NO_POOL=1 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/shexp_$e.json 2> $OUT/shexp_$e.err
done
python - <<PY
import json
base = json.loads(open("$OUT/exp_none.json").read().strip().split("\\n")[-1])["roofline"]["kernels"]["shadow"]["avg_us"]
print("real kernel alone: %.0f us" % base)
for e, name in (("8", "load + nearest light hit"), ("2", "+ masks (phases 1-2)"), ("4", "+ phase 3"), ("0", "+ hit record, emission (= the real kernel's work)")):
    us = json.loads(open("$OUT/shexp_%s.json" % e).read().strip().split("\\n")[-1])["roofline"]["kernels"]["shadow"]["avg_us"]
    print("shadow %-2s %-50s stage %7.0f us  variant %7.0f us" % (e, name, us, us - base))
PY
cd rust-pathtracer_amd/csrc && touch pt_kern_extend.hip pt_kern_shadow.hip && pt_make -j8 libptamd.so
