#!/bin/bash
# Runs bench.py on the BASELINE.json configurations C2..C5 (C1 is the CPU plumbing case) and stores one JSON line each.
OUT=${1:-gpurun_out/configs}; mkdir -p $OUT
python bench.py --steps 4 --warmup 1 --cpu-seconds 10 --spp-per-step 120 > $OUT/C2.json 2> $OUT/C2.err
python bench.py --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 --steps 3 --warmup 1 --cpu-seconds 10 \
  --workload "C3: cornell_box_diamond_gem (moissanite GGX, brilliant_diamond.obj, env constant 0), 1920x1080, max_bounces=12, L=2" > $OUT/C3.json 2> $OUT/C3.err
python bench.py --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120 --steps 3 --warmup 1 --cpu-seconds 10 \
  --workload "C4: hdri_test sphere + monkey.obj (ggx_gold), synthetic 1024x512 HDRI, 1024x1024 importance map, env_sampling_probability 0.9, 1024x1024, max_bounces=4, L=6" > $OUT/C4.json 2> $OUT/C4.err
python bench.py --hero 4 --spp-per-step 60 --steps 3 --warmup 1 --cpu-seconds 10 \
  --workload "C5: Cornell box, hero wavelength (4 wavelengths per path), 1024x1024, max_bounces=8, L=2" > $OUT/C5.json 2> $OUT/C5.err
# G1: not a BASELINE configuration — the reference tree's data/scenes/test_prism.toml (transformed mesh + lights + environment sampling): the general kernel forms
python bench.py --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120 --steps 3 --warmup 1 --cpu-seconds 10 \
  --workload "G1: test_prism.toml (rect room, xenon SharpLight, prism.obj 836 triangles in dispersive glass under a transform stack, synthetic HDRI, env_sampling_probability 0.1), 1024x1024, max_bounces=8, L=2" > $OUT/G1.json 2> $OUT/G1.err
# G2: the reference tree's data/scenes/test_bokeh.toml — 82 sphere lights, more than 64 instances: no sweep table, the top-level BVH walk (round-4 verdict, item 1); G2F: the same with a floor
# (not a reference scene: in G2 every surface is a light and no light-sample ray is ever traced; with the floor the 82-entry light list is sampled)
python bench.py --scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120 --steps 3 --warmup 1 --cpu-seconds 10 \
  --workload "G2: test_bokeh.toml (82 sphere lights of radius 0.01 = no sweep table: the top-level BVH walk; synthetic HDRI strength 0.1, env_sampling_probability 0.5, thin lens 0.1), 1024x1024, max_bounces=8, L=2" > $OUT/G2.json 2> $OUT/G2.err
python bench.py --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120 --steps 3 --warmup 1 --cpu-seconds 10 \
  --workload "G2F: test_bokeh.toml + a Lambertian floor and three spheres (not a reference scene: the 82-entry light list sampled, light-sample rays through the top-level walk), 1024x1024, max_bounces=8, L=2" > $OUT/G2F.json 2> $OUT/G2F.err
for c in C2 C3 C4 C5 G1 G2 G2F; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/$c.json").read().strip().split("\n")[-1])
    print("$c", "%.1f Ms/s" % d["value"], "ms/step %.1f" % d["ms_per_step"], "D=%.2f" % d["segments_per_sample"], "dominant", d["roofline"]["kernel"], "%.0f GB/s" % d["roofline"]["achieved"], "cpu %.3f Ms/s on %d threads" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
except Exception as e:
    print("$c failed", e, open("$OUT/$c.err").read()[-500:])
PY
done
