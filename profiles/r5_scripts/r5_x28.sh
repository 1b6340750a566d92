#!/bin/bash
# A/B: an environment direction's emission and pdf from one EnvPoint (env1) against each taking the direction and its texture coordinates itself (env0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_x28.txt; cd $ROOT
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
G1="--scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120"
G2="--scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120"
G2F="--scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120"
C4H="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 60 --hero 4"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "film_parity or hdri or prism or bokeh or full_size" 2>&1 | grep -E "passed|failed|error" > $OUT
bash tools/ab_libs.sh "variants/env0.so variants/env1.so" -- $C4 ";;" $G1 ";;" $G2 ";;" $G2F ";;" $C4H >> $OUT 2>&1
cat $OUT
