#!/bin/bash
# A/B: hero-wavelength paths carry the wavelength sample (no Philox draw per vertex for the four wavelengths); hero0 = the library before
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_x24.txt; cd $ROOT
C5="--hero 4 --spp-per-step 120"
C3H="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 30 --hero 4"
C4H="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 60 --hero 4"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hero or film_parity" 2>&1 | grep -E "passed|failed|error" > $OUT
bash tools/ab_libs.sh "variants/hero0.so variants/hero1.so" -- $C5 ";;" $C3H ";;" $C4H >> $OUT 2>&1
cat $OUT
