#!/usr/bin/env python3
"""Fills the R5_* placeholders of DESIGN.md's "Round 5 at a glance" table from profiles/<tag>_bench_*.json and <tag>_*_summary.json (tools/profile_all.sh).
usage: tools/fill_design_numbers.py <tag> [file ...]   (default file: DESIGN.md; the placeholders are replaced in place, so it runs once per file)"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1]
files = sys.argv[2:] or ["DESIGN.md"]


def bench(name):
    return json.loads(open(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, name))).read().strip().split("\n")[-1])


vals = {}
for c in ("C2", "C3", "C4", "C5", "G1", "G2", "G2F"):
    d = bench(c); r = d["roofline"]; k = r["kernels"]
    vals["R5_%s" % c] = "%.0f" % d["value"]
    vals["R5_%s_FRAC" % c] = "%.3f" % r["frac"]
    vals["R5_%s_TR" % c] = ("%.2f" % (r["traffic"] / r["algorithmic_bytes_per_launch"])) if r.get("traffic") else "n/a"
    vals["R5_%s_VALU" % c] = ("%.0f G" % r["valu"]["achieved"]) if r.get("valu") else "n/a"
    for key, stage in (("SHADE", "shade"), ("SHADOW", "shadow"), ("EXT", "extend"), ("ACC", "accumulate"), ("GEN", "generate")):
        if stage in k:
            vals["R5_%s_%s" % (c, key)] = "%.0f" % k[stage]["avg_us"]
d = bench("default")
vals["R5_DEF"] = "%.0f" % d["value"]; vals["R5_DEFMS"] = "%.1f" % d["ms_per_step"]
for f in files:
    p = os.path.join(ROOT, f)
    s = open(p).read()
    for key in sorted(vals, key=len, reverse=True):
        s = s.replace(key, vals[key])
    s = s.replace("r5z_", tag + "_") if tag != "r5z" else s
    open(p, "w").write(s)
    left = [w for w in s.split() if w.startswith("R5_") or "**R5_" in w]
    print(f, "placeholders left:", left[:10])
