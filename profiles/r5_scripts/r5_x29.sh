#!/bin/bash
# A/B: a GGX vertex' incoming-direction terms once for all its light samples (PT_GGX_WI_ONCE 1) against every evaluation computing them (0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_x29.txt; cd $ROOT
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
G1="--scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120"
G2F="--scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "film_parity or gem or hdri or prism or bokeh or full_size or material" 2>&1 | grep -E "passed|failed|error" > $OUT
bash tools/ab_libs.sh "variants/gw0.so variants/gw1.so" -- $C3 ";;" $C4 ";;" $G1 ";;" $G2F >> $OUT 2>&1
cat $OUT
