#!/bin/bash
# Runs on the MI355X box (through gpurun): rocprofv3 kernel-trace/stats and, in separate passes, the HBM PMC counters
# for the bench workload.  Raw output under gpurun_out/ (scratch); tools/summarize_profile.py distils profiles/*.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --cpu-seconds 0 $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
echo "trace rc=$?" >> "$OUT/bench_trace.log"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_fetch.log" 2>&1
echo "fetch rc=$?" >> "$OUT/bench_fetch.log"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_write.log" 2>&1
echo "write rc=$?" >> "$OUT/bench_write.log"
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_sq.log" 2>&1
echo "sq rc=$?" >> "$OUT/bench_sq.log"
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_sq2.log" 2>&1
echo "sq2 rc=$?" >> "$OUT/bench_sq2.log"
# (round 6) L2 hits and misses per kernel, a pass of its own: what the FETCH_SIZE excess of a scene with big tables is made of (bench.py roofline.l2)
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_tcc" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_tcc.log" 2>&1
echo "tcc rc=$?" >> "$OUT/bench_tcc.log"
cd "$ROOT" && python3 tools/summarize_profile.py "$OUT" "$TAG" > "$OUT/summary.log" 2>&1
find "$OUT" -name "*.csv" -size +20M -delete
ls -R "$OUT" | head -50
tail -5 "$OUT"/bench_*.log | cut -c1-400
cat "$OUT/summary.log" | tail -40
