#!/bin/bash
# Runs on the GPU box: C3's k_shadow_parked with parts left out (variants/pexp{1,2,4}.so = -DPT_PARKED_EXP bits: 1 the listing of live rays only, 2 up to the
# masks, 4 parked rays dropped = no mesh walks) against the product build, and the per-wave timeline (variants/tl.so) whose "work" column counts the parked
# rays a launch resumed: the byte account of the park records (round-3 verdict, item 3).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash tools/ab_libs.sh "variants/base.so variants/pexp1.so variants/pexp2.so variants/pexp4.so" -- --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60
PT_AMD_LIBRARY=$ROOT/variants/tl.so python tools/wave_timeline.py cornell_gem 1920 1080 60 12 2 2>&1 | grep -v "^           " | head -80
