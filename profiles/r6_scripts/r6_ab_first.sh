#!/bin/bash
# Runs on the GPU box: the grouped sweep trying an inside ray's farthest-reaching group first (variants/r6first.so = -DPT_INSIDE_FIRST_GROUP=1) against the product, alternating: C3.
bash tools/ab_libs.sh "rust-pathtracer_amd/csrc/libptamd.so variants/r6first.so" -- --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60
