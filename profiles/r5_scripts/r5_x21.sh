#!/bin/bash
# A/B: the grouped mesh sweep's group boxes decided as wave masks (PT_GROUP_WAVE_MASKS 1) against the per-lane classification (0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_x21.txt; cd $ROOT
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
G1="--scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120"
GEM="--scene test_bokeh_floor_gem --max-bounces 8 --light-samples 2 --spp-per-step 120"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gem or prism or hdri or mesh or forms" 2>&1 | tail -3 > $OUT
bash tools/ab_libs.sh "variants/gwm0.so variants/gwm1.so" -- $C3 ";;" $C4 ";;" $G1 ";;" $GEM >> $OUT 2>&1
cat $OUT
