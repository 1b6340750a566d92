"""Distils rocprofv3 CSV output (tools/profile_gpu.sh) into profiles/<tag>_*.{csv,json}: per-kernel time statistics
from --kernel-trace --stats, and per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes.

FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B? -> rocprofv3 reports them in kilobytes (MI355X_MICROARCH.md
§HBM: derived from TCC_EA0_RDREQ/WRREQ); on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, other widths are
uncalibrated, so the summary also prints the ratio against the known byte count of k_generate / k_accumulate."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_]+)", name)
    return m.group(1) if m else name[:40]


def kernel_trace(dirpath):
    rows = []
    for f in glob.glob(os.path.join(dirpath, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    return rows


def counter_rows(dirpath):
    rows = []
    for f in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    return rows


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    summary = {"tag": tag}
    # the workload the profiled command ran (bench.py's own JSON line in the trace pass), so that bench.py can tell whether
    # these per-launch numbers belong to the run it is reporting
    try:
        for line in open(os.path.join(out_dir, "bench_trace.log")):
            if line.startswith("{") and '"metric"' in line:
                b = json.loads(line)
                summary["workload"] = {"scene": b["config"]["scene"], "spp_per_step": b["config"]["spp_per_step"], "samples_per_step": b["config"]["samples_per_step"],
                                       "metric": b["metric"], "n_gpus": b["n_gpus"]}
                summary["workload_key"] = b["config"].get("workload_key")   # bench.py only takes counters from a profile whose key equals its own
    except (OSError, ValueError, KeyError):
        pass
    # 1. kernel stats
    stats = defaultdict(list)
    meta = {}
    for r in kernel_trace(os.path.join(out_dir, "trace")):
        k = short(r.get("Kernel_Name", ""))
        try:
            stats[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        except (KeyError, ValueError):
            continue
        meta[k] = {x: r.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    total = sum(sum(v) for v in stats.values()) or 1
    kern = {}
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        kern[k] = {"calls": len(v), "total_ms": sum(v) / 1e6, "avg_us": sum(v) / len(v) / 1e3, "min_us": v[0] / 1e3, "max_us": v[-1] / 1e3,
                   "median_us": v[len(v) // 2] / 1e3, "share": sum(v) / total, **meta.get(k, {})}
    summary["kernels"] = kern
    with open(os.path.join(prof, "%s_kernel_stats.csv" % tag), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "median_us", "min_us", "max_us", "share", "vgpr", "sgpr", "lds_bytes", "scratch"])
        for k, s in kern.items():
            w.writerow([k, s["calls"], "%.3f" % s["total_ms"], "%.2f" % s["avg_us"], "%.2f" % s["median_us"], "%.2f" % s["min_us"], "%.2f" % s["max_us"],
                        "%.4f" % s["share"], s.get("VGPR_Count"), s.get("SGPR_Count"), s.get("LDS_Block_Size"), s.get("Scratch_Size")])
    # copy rocprof's own stats file if present
    for f in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as src, open(os.path.join(prof, "%s_rocprofv3_kernel_stats.csv" % tag), "w") as dst:
            dst.write(src.read())
    # 2. PMC passes
    for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        acc = defaultdict(list)
        for r in counter_rows(os.path.join(out_dir, sub)):
            if r.get("Counter_Name") != name:
                continue
            acc[short(r.get("Kernel_Name", ""))].append(float(r["Counter_Value"]))
        summary[name] = {k: {"launches": len(v), "avg_per_launch": sum(v) / len(v), "total": sum(v)} for k, v in acc.items()}
    sq = defaultdict(lambda: defaultdict(float))
    for sub in ("pmc_sq", "pmc_sq2"):
        for r in counter_rows(os.path.join(out_dir, sub)):
            sq[short(r.get("Kernel_Name", ""))][r.get("Counter_Name")] += float(r["Counter_Value"])
    summary["SQ"] = {k: dict(v) for k, v in sq.items()}
    # derived: lanes active per VALU instruction (SIMT efficiency) and VALU busy share of the wave lifetime
    derived = {}
    for k, v in sq.items():
        if not k.startswith("k_"):
            continue
        dv = {}
        if v.get("SQ_ACTIVE_INST_VALU") and v.get("SQ_THREAD_CYCLES_VALU"):
            dv["valu_lane_utilization"] = v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"])
        if v.get("SQ_WAVE_CYCLES") and v.get("SQ_ACTIVE_INST_VALU"):
            dv["valu_active_share_of_wave_cycles"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
        if v.get("SQ_WAVE_CYCLES") and v.get("SQ_WAIT_INST_ANY"):
            dv["issue_wait_share_of_wave_cycles"] = v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]
        if v.get("SQ_INSTS_VALU") and v.get("SQ_WAVES"):
            dv["valu_instructions_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
        derived[k] = dv
    summary["derived"] = derived
    # L2 (TCC) hits and misses per launch, when the pass was made: <tag>_tcc.txt, one line per kernel (the format of profiles/r5z_C4_tcc.txt, which bench.py reads)
    tcc = defaultdict(lambda: defaultdict(list))
    for r in counter_rows(os.path.join(out_dir, "pmc_tcc")):
        tcc[short(r.get("Kernel_Name", ""))][r.get("Counter_Name")].append(float(r["Counter_Value"]))
    tcc_lines = ["%s %s launches %d" % (k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(cs.items())}, max(len(v) for v in cs.values())) for k, cs in sorted(tcc.items()) if k.startswith("k_")]
    if tcc_lines:
        with open(os.path.join(prof, "%s_tcc.txt" % tag), "w") as fh:
            fh.write("\n".join(tcc_lines) + "\n")
        summary["TCC"] = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in tcc.items() if k.startswith("k_")}
    with open(os.path.join(prof, "%s_summary.json" % tag), "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)
    # a copy next to the raw output: gpurun brings gpurun_out/ back, not profiles/
    for name in ("%s_summary.json" % tag, "%s_kernel_stats.csv" % tag, "%s_rocprofv3_kernel_stats.csv" % tag, "%s_tcc.txt" % tag):
        try:
            with open(os.path.join(prof, name)) as src, open(os.path.join(out_dir, name), "w") as dst:
                dst.write(src.read())
        except OSError:
            pass
    print(json.dumps(summary.get("derived", {}), indent=1, sort_keys=True)[:4000])


if __name__ == "__main__":
    main()
