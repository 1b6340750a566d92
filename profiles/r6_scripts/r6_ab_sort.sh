#!/bin/bash
# Runs on the GPU box: k_shade's surface vertices sorted by material kind (PT_SHADE_SORT, the product) against the build without (variants/r6nosort.so), alternating: C3, G1, G2F, C4.
bash tools/ab_libs.sh "variants/r6nosort.so rust-pathtracer_amd/csrc/libptamd.so" -- --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 ";;" \
  --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" \
  --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120
