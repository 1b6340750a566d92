#!/bin/bash
# Round 6, on the GPU box (round-5 verdict item 5: "measure before declining again"): C4's FULL vertex kernel with its table reads folded onto 1/2 and 1/16 of the importance map's
# rows and the environment's texel rows (variants/r6fold*.so, -DPT_EXP_TABLE_FOLD: a TIMING experiment, the results are wrong) — the same instructions, the same number of reads,
# a working set of 10 MB / 1.25 MB instead of 20 MB: the ceiling of any reordering of the searches (band passes).  Then the product library on C3 and C4.
C4="--scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120"
bash tools/ab_libs.sh "variants/r6fold1.so variants/r6fold2.so variants/r6fold16.so" -- $C4
echo "== product library: C3, C4"
bash tools/ab_libs.sh "rust-pathtracer_amd/csrc/libptamd.so" -- --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60 ";;" $C4
