#!/bin/bash
OUT=gpurun_out/r5l_exit.txt; : > $OUT
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
G1="--scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120"
G2="--scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120"
bash tools/ab_libs.sh "variants/exit0.so variants/exit1.so" -- --spp-per-step 240 ";;" $C3 ";;" $C4 ";;" $G1 ";;" $G2 >> $OUT 2>&1
cat $OUT
