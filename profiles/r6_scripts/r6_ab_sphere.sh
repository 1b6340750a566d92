#!/bin/bash
# Round 6, on the GPU box: certified culling of top-level nodes that hold spheres (beyond_sphere) — the product library against variants/r6nosph.so (-DPT_SPHERE_CULL=0)
bash tools/ab_libs.sh "variants/r6nosph.so rust-pathtracer_amd/csrc/libptamd.so" -- --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" \
  --scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120
