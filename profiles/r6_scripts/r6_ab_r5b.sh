#!/bin/bash
# Round 6, on the GPU box: round 5's final library, this round's before (variants/r6cur.so) and after (variants/r6cur2.so) the certificate code went behind scalar branches
bash tools/ab_libs.sh "variants/r5final.so variants/r6cur4.so" -- --spp-per-step 120 ";;" --hero 4 --spp-per-step 60 ";;" \
  --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" \
  --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120 ";;" --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60
