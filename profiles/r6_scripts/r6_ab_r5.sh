#!/bin/bash
# Round 6, on the GPU box: the round-5 final library (variants/r5final.so, built from commit c3ff4b0) against this round's on ONE box, every configuration: what the round's
# changes cost the configurations they were not made for (the vertex code's certificate test, hit_record's flag word).
bash tools/ab_libs.sh "variants/r5final.so variants/r6cur.so" -- --spp-per-step 120 ";;" --hero 4 --spp-per-step 60 ";;" --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120 ";;" \
  --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" --scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" \
  --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60
