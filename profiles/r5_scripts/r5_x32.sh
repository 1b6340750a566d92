#!/bin/bash
# A/B: machine-scheduler strategies for the vertex and light-sample kernels (hipcc -mllvm flags), the build's own flags = base
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_x32.txt; cd $ROOT
C2="--spp-per-step 240"
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
C5="--hero 4 --spp-per-step 120"
bash tools/ab_libs.sh "variants/base.so variants/maxilp.so variants/maxmem.so variants/bias100.so variants/trackers.so" -- $C2 ";;" $C3 ";;" $C4 ";;" $C5 > $OUT 2>&1
cat $OUT
