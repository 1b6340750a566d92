#!/bin/bash
# Runs on the GPU box: the engine built with -mllvm -disable-machine-licm (constants are rematerialised inside the loops instead of being hoisted into
# registers that stay live across the kernels' long outer loops) against the default build, on C2..C5.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash tools/ab_libs.sh "PT_AMD_NO_LIVE_LISTS=1:variants/live6.so PT_AMD_NO_LIVE_LISTS=1:variants/live6nolicm.so" -- --spp-per-step 240 ";;" --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 ";;" --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120 ";;" --hero 4 --spp-per-step 60
