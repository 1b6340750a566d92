#!/bin/bash
# Runs on the GPU box: where do k_extend's LDS bank conflicts come from?  (round-2 verdict, item 8: 1.24 conflict cycles per LDS instruction.)
# The measurement variants of k_extend (variants/exp.so = tools/build_variant.sh exp -DPT_EXPERIMENTS; k_extend_exp<bits> is launched in front of
# the real kernel and leaves parts out: 6 = phases 1-2 only, 4 = + phase 3, 0 = + hit record) are run under rocprofv3 with the LDS counters;
# the differences between the variants are the phases' own numbers.  usage: tools/lds_conflicts.sh <out dir>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/${1:-gpurun_out/lds_conflicts}; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
export PT_AMD_LIBRARY=$ROOT/variants/exp.so PT_AMD_NO_FUSE=1
for e in 6 4 0; do
  PT_AMD_EXP=$e timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES --kernel-trace --output-format csv -d $OUT/exp_$e -- python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --spp-per-step 240 > $OUT/bench_$e.log 2>&1
done
python3 - <<PY
import csv, glob, collections
res = {}
for e in (6, 4, 0):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for f in glob.glob("$OUT/exp_%d/**/*counter_collection.csv" % e, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").replace("ptk::", "").split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
    for k in sorted(acc):
        if k.startswith(("k_extend", "k_shade", "k_shadow")):
            a = acc[k]; n = max(1, cnt[k])
            print("exp %d  %-40s launches %3d  LDS instr/launch %.4g  conflict cycles/launch %.4g  per LDS instr %.3f  idx_active %.4g addr_conflict %.4g unaligned %.4g" %
                  (e, k[:40], n, a["SQ_INSTS_LDS"] / n, a["SQ_LDS_BANK_CONFLICT"] / n, a["SQ_LDS_BANK_CONFLICT"] / max(1.0, a["SQ_INSTS_LDS"]), a["SQ_LDS_IDX_ACTIVE"] / n, a["SQ_LDS_ADDR_CONFLICT"] / n, a["SQ_LDS_UNALIGNED_STALL"] / n))
            if k.startswith("k_extend_exp"): res[e] = (a["SQ_INSTS_LDS"] / n, a["SQ_LDS_BANK_CONFLICT"] / n)
if len(res) == 3:
    print("phases 1-2: %.4g LDS instr, %.4g conflict cycles (%.3f per instr)" % (res[6][0], res[6][1], res[6][1] / res[6][0]))
    print("phase 3   : %.4g LDS instr, %.4g conflict cycles (%.3f per instr)" % (res[4][0] - res[6][0], res[4][1] - res[6][1], (res[4][1] - res[6][1]) / max(1.0, res[4][0] - res[6][0])))
    print("hit record: %.4g LDS instr, %.4g conflict cycles (%.3f per instr)" % (res[0][0] - res[4][0], res[0][1] - res[4][1], (res[0][1] - res[4][1]) / max(1.0, res[0][0] - res[4][0])))
PY
