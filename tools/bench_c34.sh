#!/bin/bash
# C3 and C4 only
OUT=$1; mkdir -p $OUT
python bench.py --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/C3.json 2> $OUT/C3.err
python bench.py --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120 --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/C4.json 2> $OUT/C4.err
for c in C3 C4; do python - <<PY
import json
d=json.loads(open("$OUT/$c.json").read().strip().split("\n")[-1])
print("$c", "%.1f Ms/s" % d["value"], {k:round(v["avg_us"]) for k,v in d["roofline"]["kernels"].items()})
PY
done
