#!/bin/bash
# Round 6: the convex-body certificates (pt_blob.h PT_INST_CONVEX_*) on and off, same box, alternating: C3, G1 (prism: convex?), C2 (no certified instance: the cost of the code alone)
OUT=gpurun_out/r6c_ab_convex.txt; : > $OUT
c3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 --steps 3 --warmup 1 --cpu-seconds 0.5"
g1="--scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120 --steps 3 --warmup 1 --cpu-seconds 0.5"
c2="--steps 4 --warmup 1 --cpu-seconds 0.5 --spp-per-step 120"
line() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
k = d["roofline"]["kernels"]
print("%.1f Msamples/s; us per launch: %s" % (d["value"], " ".join("%s %.0f" % (n, k[n]["avg_us"]) for n in ("extend", "shade", "shadow") if n in k)))
PY
}
for rep in 1 2; do
  for cfg in c3 g1 c2; do
    for off in 0 1; do
      if [ $off = 1 ]; then export PT_AMD_NO_CONVEX=1; else unset PT_AMD_NO_CONVEX; fi
      python bench.py ${!cfg} > /tmp/ab.json 2> /tmp/ab.err || tail -5 /tmp/ab.err >> $OUT
      echo "$cfg run $rep NO_CONVEX=$off: $(line /tmp/ab.json)" >> $OUT
    done
  done
done
unset PT_AMD_NO_CONVEX
cat $OUT
