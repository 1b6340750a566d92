#!/bin/bash
OUT=gpurun_out/r5p_ww.txt; : > $OUT
G2="--scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120"
G2F="--scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120"
GEM="--scene test_bokeh_floor_gem --max-bounces 8 --light-samples 2 --spp-per-step 120"
bash tools/ab_libs.sh "variants/ww0.so variants/ww1.so PT_AMD_WALK_SEARCH_BELOW=1:variants/ww1.so PT_AMD_WALK_SEARCH_BELOW=32:variants/ww1.so" -- $G2F ";;" $G2 ";;" $GEM >> $OUT 2>&1
cat $OUT
