#!/bin/bash
# Round 6, on the GPU box: the vertex forms of scenes with certificates (K_SHADE_NC / K_SHADE_FC) before (variants/r6prev.so) and after the certificate's values stopped living
# through the sampling code (marks packed into the ray counter, the instance re-read from the hit queue, the certificate worked out where it is needed); and the grouped sweep's
# eviction threshold now that the ray mix of k_extend_parked has changed
bash tools/ab_libs.sh "variants/r6prev.so rust-pathtracer_amd/csrc/libptamd.so PT_AMD_GROUP_EVICT_BELOW=16:rust-pathtracer_amd/csrc/libptamd.so PT_AMD_GROUP_EVICT_BELOW=48:rust-pathtracer_amd/csrc/libptamd.so PT_AMD_WALK_EVICT_BELOW=16:rust-pathtracer_amd/csrc/libptamd.so PT_AMD_WALK_EVICT_BELOW=48:rust-pathtracer_amd/csrc/libptamd.so" -- \
  --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 ";;" --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120
