cd $GRAFT_REPO_ROOT
one() { timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('   %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"; }
C3="--scene cornell_gem --width 1920 --height 1080 --max-bounces 12 --spp-per-step 60"
for rep in 1 2; do
echo "C3 default (shadow 512 / extend 256)"; one $C3
echo "C3 PT_AMD_PARK_BLOCK=256"; PT_AMD_PARK_BLOCK=256 one $C3
echo "C3 PT_AMD_PARK_BLOCK=512"; PT_AMD_PARK_BLOCK=512 one $C3
done
echo C2; one; echo C4; one --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120; echo C5; one --hero 4 --spp-per-step 60
# C4: L2 hit rates of k_shade (item 6)
bash tools/pmc_quick.sh gpurun_out/pmc_c4_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120 2>&1 | grep "k_"
bash tools/pmc_quick.sh gpurun_out/pmc_c4_tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" --scene hdri_test --max-bounces 4 --light-samples 6 --spp-per-step 120 2>&1 | grep "k_"
find gpurun_out/pmc_c4_tcc gpurun_out/pmc_c4_tcp -name "*.csv" -size +1M -delete
