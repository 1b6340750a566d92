#!/bin/bash
# A/B: one origin per listed light-sample item (PT_SHARED_ORIGIN 1; so0 = the library before)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_x25.txt; cd $ROOT
C2="--spp-per-step 240"
C5="--hero 4 --spp-per-step 120"
DL="--scene disk_lamp --spp-per-step 120"
python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "film_parity or forms or cornell or fuzz or random" 2>&1 | grep -E "passed|failed|error" > $OUT
bash tools/ab_libs.sh "variants/so0.so variants/so1.so" -- $C2 ";;" $C5 ";;" $DL >> $OUT 2>&1
cat $OUT
