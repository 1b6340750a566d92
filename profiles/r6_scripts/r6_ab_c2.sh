#!/bin/bash
# Round 6, on the GPU box: (1) C2 / C5 with the entry-distance reject in phase 3 of the sweep kernels (variants/r6er.so, -DPT_SWEEP_ENTRY_REJECT=1) against the same source without
# it (variants/r6cur.so) — round-5 verdict item 6; (2) the FULL vertex form at 3 / 4 / 5 waves per SIMD on the final kernels (variants/r6sw3.so, r6cur.so, r6sw5.so): G1 (the form with
# a light list: 144 B of scratch at four waves) and C4 — verdict item 5's second half.
bash tools/ab_libs.sh "variants/r6cur.so variants/r6er.so" -- --spp-per-step 120 ";;" --hero 4 --spp-per-step 60
echo "== FULL vertex form by waves per SIMD: G1, C4"
bash tools/ab_libs.sh "variants/r6sw3.so variants/r6cur.so variants/r6sw5.so" -- --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120 ";;" --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120
