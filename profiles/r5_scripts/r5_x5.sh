#!/bin/bash
OUT=gpurun_out/r5i_group.txt; : > $OUT
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
G2FG="--scene test_bokeh_floor_gem --max-bounces 8 --light-samples 2 --spp-per-step 120"
bash tools/ab_libs.sh "variants/grp0.so variants/grp1.so" -- $C3 ";;" $G2FG >> $OUT 2>&1
cat $OUT
