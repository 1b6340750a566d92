#!/bin/bash
# Runs on the GPU box: the leaf sweep against the top-level walk on ONE scene of 62 instances (test_bokeh_floor_58: G2F with 29 lights per row), alternating; then G2F itself.
L=rust-pathtracer_amd/csrc/libptamd.so
bash tools/ab_libs.sh "$L PT_AMD_NO_SWEEP=1:$L" -- --scene test_bokeh_floor_58 --max-bounces 8 --light-samples 2 --spp-per-step 120
bash tools/ab_libs.sh "$L" -- --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120
