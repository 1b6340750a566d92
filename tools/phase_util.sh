#!/bin/bash
# Runs on the GPU box.  Round-3 verdict item 4: per-PHASE lane utilisation of C2's traversal and light-sample kernels.  The measurement variants of
# variants/exp.so (tools/build_variant.sh exp -DPT_EXPERIMENTS: k_extend_exp<bits> / k_shadow_exp<bits>, the kernels with parts left out, launched in front
# of the real ones on the same input) run under the SQ counters; a phase's utilisation is the difference of two variants:
#   SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU).  The fused k_shade of the default C2 run = k_extend's phases + the unfused k_shade's vertex code.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/phase_util; rm -rf $OUT; mkdir -p $OUT
CTRS="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVE_CYCLES"
run() { # tag, extend bits, shadow bits
  PT_AMD_LIBRARY=$ROOT/variants/exp.so PT_AMD_NO_FUSE=1 PT_AMD_EXP=$2 PT_AMD_EXP_SHADOW=$3 bash tools/pmc_quick.sh $OUT/$1 "$CTRS" > $OUT/$1.txt 2>&1
  PT_AMD_LIBRARY=$ROOT/variants/exp.so PT_AMD_NO_FUSE=1 PT_AMD_EXP=$2 PT_AMD_EXP_SHADOW=$3 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 > $OUT/$1.json
  find $OUT/$1 -name "*.csv" -size +1M -delete
}
run a 6 8; run b 4 2; run c 0 4; run d 0 0
PT_AMD_LIBRARY=$ROOT/variants/exp.so bash tools/pmc_quick.sh $OUT/fused "$CTRS" > $OUT/fused.txt 2>&1
python3 - <<'PY'
import ast, json, re
out = "gpurun_out/phase_util"
def counters(tag):
    d = {}
    for line in open("%s/%s.txt" % (out, tag)):
        m = re.match(r"(k_\w+) (\{.*\}) launches (\d+)", line)
        if m: d[m.group(1)] = {k: float(v) for k, v in ast.literal_eval(m.group(2)).items()}
    return d
def us(tag, stage): return json.loads(open("%s/%s.json" % (out, tag)).read())["roofline"]["kernels"][stage]["avg_us"]
c = {t: counters(t) for t in "abcd"}
base_ext, base_sh = None, None
def row(name, hi, lo, kern):
    a = c[hi][kern]; b = c[lo][kern] if lo else {k: 0.0 for k in a}
    dt = a["SQ_THREAD_CYCLES_VALU"] - b["SQ_THREAD_CYCLES_VALU"]; da = a["SQ_ACTIVE_INST_VALU"] - b["SQ_ACTIVE_INST_VALU"]; di = a["SQ_INSTS_VALU"] - b["SQ_INSTS_VALU"]
    ds = a["SQ_INSTS_SALU"] - b["SQ_INSTS_SALU"]
    print("%-58s VALU instr per launch %9.3g M  lane utilisation %.2f  SALU instr per launch %9.3g M" % (name, di / 1e6, dt / (64.0 * da) if da else float("nan"), ds / 1e6))
print("k_extend (= the traversal half of the fused k_shade), per launch averages of C2:")
row("  load + phases 1-2 (22 box tests, masks)", "a", None, "k_extend_exp")
row("  phase 3 (candidate primitives, lane by lane)", "b", "a", "k_extend_exp")
row("  hit record + store", "c", "b", "k_extend_exp")
row("  whole kernel", "c", None, "k_extend_exp")
print("k_shadow (two rays per item):")
row("  load + nearest light hit (light leaves)", "a", None, "k_shadow_exp")
row("  phases 1-2 (masks)", "b", "a", "k_shadow_exp")
row("  phase 3", "c", "b", "k_shadow_exp")
row("  light's record + emission + contribution", "d", "c", "k_shadow_exp")
row("  whole kernel", "d", None, "k_shadow_exp")
for k in ("k_shade", "k_extend", "k_shadow"):
    if k in c["d"]: row("real %s (unfused run)" % k, "d", None, k)
f = counters("fused")
for k, v in f.items():
    if k in ("k_shade", "k_shadow"):
        print("default run (fused) %-10s VALU instr per launch %9.3g M  lane utilisation %.2f  SALU instr per launch %9.3g M" % (k, v["SQ_INSTS_VALU"] / 1e6, v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), v["SQ_INSTS_SALU"] / 1e6))
PY
