#!/bin/bash
OUT=gpurun_out/r5u_ballany.txt; : > $OUT
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
G1="--scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120"
bash tools/ab_libs.sh "variants/ballnl.so variants/ballany.so" -- $C4 ";;" $G1 >> $OUT 2>&1
cat $OUT
