#!/bin/bash
# Runs on the GPU box: C2 (and C5) with k_shadow's live-ray lists (k_shadow_live at 6 / 5 / 4 waves per SIMD, and built without machine LICM) against the plain k_shadow.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash tools/ab_libs.sh "PT_AMD_NO_LIVE_LISTS=1:variants/live6.so variants/live6.so variants/live5.so variants/live4.so variants/live6nolicm.so PT_AMD_NO_LIVE_LISTS=1:variants/live6nolicm.so" -- --spp-per-step 240 ";;" --hero 4 --spp-per-step 60
