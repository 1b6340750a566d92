#!/bin/bash
# A/B: slot / chunk_pixels and pixel / width by the host's reciprocal (PT_MAGIC_DIVISION 1) against the compiler's expansion of / by a run-time value (0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_x31.txt; cd $ROOT
C2="--spp-per-step 240"
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
C5="--hero 4 --spp-per-step 120"
G2="--scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "film_parity or full_size or shard or multi or camera or sample_range" 2>&1 | grep -E "passed|failed|error" > $OUT
bash tools/ab_libs.sh "variants/div0.so variants/div1.so" -- $C2 ";;" $C3 ";;" $C4 ";;" $C5 ";;" $G2 >> $OUT 2>&1
cat $OUT
