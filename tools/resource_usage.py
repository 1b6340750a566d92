#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels of one translation unit, from hipcc's -Rpass-analysis=kernel-resource-usage
(cross-compiles for gfx950: runs without a GPU).

  tools/resource_usage.py pt_kern_shadow.hip [-DPT_SHADE_NL=4 ...] [--filter k_shadow] [--save profiles/r3_resource_usage.txt]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust-pathtracer_amd", "csrc")
FLAGS = "-std=c++17 -O2 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -w".split()


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"^void ptk::", "", re.sub(r"\(.*$", "", o)) for o in out]


def usage(source, extra):
    # (the Makefile's per-file flags: the light-sample and vertex kernels are built without machine LICM)
    per_file = ["-mllvm", "-disable-machine-licm"] if os.path.basename(source) in ("pt_kern_shadow.hip", "pt_kern_shade.hip") and "-disable-machine-licm" not in extra and "--licm" not in extra else []
    extra = [e for e in extra if e != "--licm"]
    if os.path.basename(source) == "pt_kern_shade.hip" and not any(e.startswith("-DPT_SHADE_PART") for e in extra):
        # (both parts of the vertex family in one table: the lean forms, then the rest with its PT_QUEUE_NT)
        return usage(source, extra + ["-DPT_SHADE_PART=0"]) + usage(source, extra + ["-DPT_SHADE_PART=1", "-DPT_QUEUE_NT=1"])
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + FLAGS + per_file + extra + ["--cuda-device-only", "-c", source, "-o", "/dev/null",
                                                                                 "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if not m:
            if "error:" in line:
                print(line, file=sys.stderr)
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    return rows


def main():
    args = sys.argv[1:]
    flt, save = None, None
    if "--filter" in args:
        i = args.index("--filter"); flt = args[i + 1]; del args[i:i + 2]
    if "--save" in args:
        i = args.index("--save"); save = args[i + 1]; del args[i:i + 2]
    source, extra = args[0], args[1:]
    rows = usage(source, extra)
    names = demangle([r["name"] for r in rows])
    lines = ["%-64s %5s %5s %5s %7s %6s %6s %4s %6s" % ("kernel (%s %s)" % (source, " ".join(extra)), "VGPR", "AGPR", "SGPR", "scratch", "vspill", "sspill", "occ", "LDS")]
    for r, n in zip(rows, names):
        if flt and flt not in n:
            continue
        lines.append("%-64s %5s %5s %5s %7s %6s %6s %4s %6s" % (n, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"), r.get("ScratchSize", "?"),
                                                                 r.get("VGPRs Spill", "?"), r.get("SGPRs Spill", "?"), r.get("Occupancy", "?"), r.get("LDS Size", "?")))
    text = "\n".join(lines)
    print(text)
    if save:
        with open(os.path.join(ROOT, save), "a") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
