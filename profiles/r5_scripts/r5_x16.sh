#!/bin/bash
OUT=gpurun_out/r5v2_dop.txt; : > $OUT
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
G1="--scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120"
GEM="--scene test_bokeh_floor_gem --max-bounces 8 --light-samples 2 --spp-per-step 120"
bash tools/ab_libs.sh "variants/dop0.so variants/dopk.so" -- $C3 ";;" $C4 ";;" $G1 ";;" $GEM >> $OUT 2>&1
cat $OUT
