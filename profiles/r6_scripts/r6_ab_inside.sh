#!/bin/bash
# Round 6, on the GPU box: path segments that start INSIDE the scene's one certified convex body end that body's sweep at the first triangle accepted well inside itself
# (mesh_walk `inside`; the product) against variants/r6prev.so (the same source without it): C3, G1
bash tools/ab_libs.sh "variants/r6prev.so rust-pathtracer_amd/csrc/libptamd.so" -- --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 ";;" \
  --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120
