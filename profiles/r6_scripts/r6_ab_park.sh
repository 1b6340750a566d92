#!/bin/bash
# Round 6, on the GPU box: C3 after the convex certificates — path marks (the product library against variants/r6base.so, built before them), and the parked kernels' occupancy now
# that few rays walk: workgroup size (PT_AMD_PARK_BLOCK) and waves per SIMD (variants/r6w6.so, r6w8.so: PT_PARK_WAVES 6 / 8)
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
bash tools/ab_libs.sh "variants/r6base.so rust-pathtracer_amd/csrc/libptamd.so PT_AMD_PARK_BLOCK=256:variants/r6base.so PT_AMD_PARK_BLOCK=512:variants/r6base.so PT_AMD_PARK_BLOCK=256:variants/r6w6.so PT_AMD_PARK_BLOCK=256:variants/r6w8.so variants/r6w6.so" -- $C3
