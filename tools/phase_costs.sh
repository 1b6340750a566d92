#!/bin/bash
# Runs on the MI355X box: where does k_extend spend its instructions?  Builds the engine with the measurement variants
# (-DPT_EXPERIMENTS: k_extend_exp<bits>, pt_kernels.h), runs the bench once per variant — the variant is launched in front of the real
# kernel, so the extend stage grows by the variant's cost — and prints the differences.  The product build is restored afterwards.
# bits: 1 pooled form, 2 no phase 3, 4 no hit record, pooled only: 8 no triangle chunks, 0x10 no replay, 0x20 no winner, 0x40 no candidate lists
source "$(dirname "$0")/lib_build.sh"
OUT=${1:-gpurun_out/phase_costs}; mkdir -p $OUT
cd rust-pathtracer_amd/csrc && touch pt_kern_extend.hip pt_kern_shadow.hip && pt_make -j8 EXTRA=-DPT_EXPERIMENTS libptamd.so; cd ../..
for e in none 6 4 0 7 125 61 53 37 5 1; do
  if [ $e = none ]; then unset PT_AMD_EXP; else export PT_AMD_EXP=$e; fi
  PT_AMD_NO_POOL=1 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/exp_$e.json 2> $OUT/exp_$e.err
done
unset PT_AMD_EXP
for e in 8 2 4 0; do
  PT_AMD_EXP_SHADOW=$e PT_AMD_NO_POOL=1 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/shexp_$e.json 2> $OUT/shexp_$e.err
done
python - <<PY
import json
base = json.loads(open("$OUT/exp_none.json").read().strip().split("\\n")[-1])["roofline"]["kernels"]["shadow"]["avg_us"]
for e, name in (("8", "load + nearest light hit"), ("2", "+ masks (phases 1-2)"), ("4", "+ phase 3"), ("0", "+ hit record, emission (= the real kernel)")):
    us = json.loads(open("$OUT/shexp_%s.json" % e).read().strip().split("\\n")[-1])["roofline"]["kernels"]["shadow"]["avg_us"]
    print("shadow %-2s %-45s stage %7.0f us  variant %7.0f us" % (e, name, us, us - base))
PY
python - <<PY
import json
names = {"none": "real kernel alone (lane-by-lane sweep)", "6": "lane: phases 1-2 only", "4": "lane: + phase 3", "0": "lane: + hit record (= the real kernel)",
         "7": "pooled: phases 1-2 only", "125": "pooled: + ray records, scan", "61": "pooled: + candidate lists, owner tests", "53": "pooled: + triangle chunks", "37": "pooled: + replay",
         "5": "pooled: + winner", "1": "pooled: + hit record (= k_extend_pooled)"}
base = None
for e in ["none", "6", "4", "0", "7", "125", "61", "53", "37", "5", "1"]:
    try:
        d = json.loads(open("$OUT/exp_%s.json" % e).read().strip().split("\n")[-1])
        us = d["roofline"]["kernels"]["extend"]["avg_us"]
        if base is None: base = us
        print("%-4s %-50s extend stage %7.0f us  variant %7.0f us" % (e, names[e], us, us - base))
    except Exception as ex:
        print(e, "failed", ex)
PY
cd rust-pathtracer_amd/csrc && touch pt_kern_extend.hip pt_kern_shadow.hip && pt_make -j8 libptamd.so
