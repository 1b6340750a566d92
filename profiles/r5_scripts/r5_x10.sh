#!/bin/bash
OUT=gpurun_out/r5n_bound.txt; : > $OUT
bash tools/ab_libs.sh "variants/bound0.so variants/bound1.so" -- --spp-per-step 240 ";;" --hero 4 --spp-per-step 60 ";;" --scene disk_lamp --spp-per-step 120 >> $OUT 2>&1
cat $OUT
