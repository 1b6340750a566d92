#!/bin/bash
# Runs on the GPU box: the leaf sweep against the top-level walk (PT_AMD_NO_SWEEP=1) on a DENSE scene of analytic instances: a closed room with 12 / 28 / 44 spheres (19 / 35 / 51 instances).
L=rust-pathtracer_amd/csrc/libptamd.so
for n in rect_room_12 rect_room rect_room_44; do echo "#### $n"; bash tools/ab_libs.sh "$L PT_AMD_NO_SWEEP=1:$L" -- --scene $n --max-bounces 8 --light-samples 2 --spp-per-step 120; done
