#!/bin/bash
# Runs on the GPU box: leaves per group of the grouped mesh sweep (pt_blob.h PT_MESH_GROUP: 6) after the inside rule — 5 (variants/r6g5.so, PT_MESH_SWEEP_MAX 320) and 8 (r6g8.so): C3.
bash tools/ab_libs.sh "rust-pathtracer_amd/csrc/libptamd.so variants/r6g5.so variants/r6g8.so" -- --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60
