#!/bin/bash
# Runs on the GPU box: one profile tag for every configuration (round-4 verdict, item 6) — tools/profile_gpu.sh for C2..C5, G1, G2, G2F on the kernels as they are,
# the bench records of the same build (tools/bench_configs.sh) and the default bench.py record.  usage: tools/profile_all.sh <tag>
TAG=${1:-r6z}
mkdir -p gpurun_out
p() { name=$1; shift; bash tools/profile_gpu.sh ${TAG}_$name "$@" > gpurun_out/${TAG}_prof_$name.log 2>&1; find gpurun_out/prof_${TAG}_$name -name "*.csv" -size +2M -delete; }
p C2 --spp-per-step 120
p C3 --scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60
p C4 --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120
p C5 --hero 4 --spp-per-step 60
p G1 --scene test_prism --max-bounces 8 --light-samples 2 --spp-per-step 120
p G2 --scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120
p G2F --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120
# the default record's own workload (1024 spp per step) needs its own summary: the workload key holds spp_per_step
p C2full
bash tools/bench_configs.sh gpurun_out/${TAG}_configs
for c in C2 C3 C4 C5 G1 G2 G2F; do cp gpurun_out/${TAG}_configs/$c.json gpurun_out/${TAG}_bench_$c.json; done
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 600 gpurun_out/${TAG}_bench_default.json
