#!/bin/bash
# Runs on the GPU box: the eviction thresholds once more, after the inside rule changed what the parked closest-hit kernel's waves hold: C3 (the gem's grouped sweep: group_evict_below), G1 (the prism's walk: walk_evict_below).
L=rust-pathtracer_amd/csrc/libptamd.so
echo "#### C3, PT_AMD_GROUP_EVICT_BELOW (default 32)"
bash tools/ab_libs.sh "$L PT_AMD_GROUP_EVICT_BELOW=16:$L PT_AMD_GROUP_EVICT_BELOW=24:$L PT_AMD_GROUP_EVICT_BELOW=40:$L PT_AMD_GROUP_EVICT_BELOW=48:$L" -- --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60
echo "#### G1, PT_AMD_WALK_EVICT_BELOW (default 32)"
bash tools/ab_libs.sh "$L PT_AMD_WALK_EVICT_BELOW=16:$L PT_AMD_WALK_EVICT_BELOW=24:$L PT_AMD_WALK_EVICT_BELOW=40:$L PT_AMD_WALK_EVICT_BELOW=48:$L" -- --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120
