#!/bin/bash
OUT=gpurun_out/r5m3_ball.txt; : > $OUT
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
bash tools/ab_libs.sh "variants/ball0.so variants/ball1.so" -- $C3 >> $OUT 2>&1
cat $OUT
