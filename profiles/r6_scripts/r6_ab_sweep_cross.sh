#!/bin/bash
# Runs on the GPU box: where the leaf sweep stops paying — test_bokeh_floor with 5 / 9 / 13 / 21 lights per row (14 / 22 / 30 / 46 instances), sweep against PT_AMD_NO_SWEEP=1.
L=rust-pathtracer_amd/csrc/libptamd.so
for n in 10 18 26 42; do echo "#### test_bokeh_floor_$n"; bash tools/ab_libs.sh "$L PT_AMD_NO_SWEEP=1:$L" -- --scene test_bokeh_floor_$n --max-bounces 8 --light-samples 2 --spp-per-step 120; done
echo "#### C2 (the Cornell box: 9 instances)"; bash tools/ab_libs.sh "$L PT_AMD_NO_SWEEP=1:$L" -- --spp-per-step 120
