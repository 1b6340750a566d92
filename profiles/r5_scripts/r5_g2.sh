#!/bin/bash
# G2 / G2F: bench + profile (round 5).  usage: tools/r5_g2.sh <tag>
TAG=${1:-r5a}
mkdir -p gpurun_out
G2W="G2: test_bokeh.toml (82 sphere lights of radius 0.01 = no sweep table: the top-level BVH walk; synthetic HDRI strength 0.1, env_sampling_probability 0.5, thin lens 0.1), 1024x1024, max_bounces=8, L=2"
G2FW="G2F: test_bokeh.toml + a Lambertian floor and three spheres (not a reference scene: the 82-entry light list sampled, light-sample rays through the top-level walk), 1024x1024, max_bounces=8, L=2"
python bench.py --scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120 --steps 3 --warmup 1 --cpu-seconds 10 --workload "$G2W" > gpurun_out/${TAG}_bench_G2.json 2> gpurun_out/${TAG}_bench_G2.err
python bench.py --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120 --steps 3 --warmup 1 --cpu-seconds 10 --workload "$G2FW" > gpurun_out/${TAG}_bench_G2F.json 2> gpurun_out/${TAG}_bench_G2F.err
bash tools/profile_gpu.sh ${TAG}_G2 --scene test_bokeh --max-bounces 8 --light-samples 2 --spp-per-step 120 > gpurun_out/${TAG}_prof_G2.log 2>&1
bash tools/profile_gpu.sh ${TAG}_G2F --scene test_bokeh_floor --max-bounces 8 --light-samples 2 --spp-per-step 120 > gpurun_out/${TAG}_prof_G2F.log 2>&1
find gpurun_out/prof_${TAG}_G2 gpurun_out/prof_${TAG}_G2F -name "*.csv" -size +2M -delete
cat gpurun_out/${TAG}_bench_G2.json gpurun_out/${TAG}_bench_G2F.json | cut -c1-1500
