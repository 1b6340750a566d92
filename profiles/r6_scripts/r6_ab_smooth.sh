#!/bin/bash
# Round 6, on the GPU box: the product library (certificates face by face, smooth-shaded bodies too: G1's prism) against variants/r6cur4.so (flat-shaded bodies only)
bash tools/ab_libs.sh "variants/r6cur4.so rust-pathtracer_amd/csrc/libptamd.so" -- --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" \
  --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 ";;" --spp-per-step 120 ";;" --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120
