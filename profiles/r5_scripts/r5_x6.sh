#!/bin/bash
OUT=gpurun_out/r5j2_camera.txt; : > $OUT
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
G1="--scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120"
bash tools/ab_libs.sh "variants/cam0.so variants/cam1.so" -- $C3 ";;" $G1 >> $OUT 2>&1
cat $OUT
