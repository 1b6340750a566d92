#!/bin/bash
# Runs on the GPU box: the top level by groups (the product; PT_AMD_NO_TOP_GROUPS=1 = the tree walk on the same library) against the library before it (variants/r6prev.so), alternating: G2F, G2.
L=rust-pathtracer_amd/csrc/libptamd.so
bash tools/ab_libs.sh "variants/r6prev.so $L PT_AMD_NO_TOP_GROUPS=1:$L" -- --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" --scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120
