#!/bin/bash
# Runs on the GPU box: the inside rule in mesh_walk's while-while form too (the product) against the library before it (variants/r6prev.so), alternating: G1 (the prism is walked), C3 (the gem is swept).
bash tools/ab_libs.sh "variants/r6prev.so rust-pathtracer_amd/csrc/libptamd.so" -- --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" \
  --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60
