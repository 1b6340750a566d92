#!/usr/bin/env python3
"""The "round at a glance" table of profiles/r<N>_experiments.md section 0, from the committed records of one profile tag (tools/profile_all.sh):
profiles/<tag>_bench_<cfg>.json (bench.py's line on that build) next to the previous round's.  usage: tools/round_table.py <tag> [<previous tag>]   (markdown on stdout)"""
import json
import os
import sys

P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
tag, prev = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
ROWS = [("default", "C2, the default `bench.py` run: the whole 1024-spp frame per step"), ("C2", "C2 Cornell 1024², depth 8, L = 2 (120 spp per step)"),
        ("C3", "C3 gem 1920×1080, depth 12 (60 spp per step)"), ("C4", "C4 HDRI + monkey 1024², depth 4, L = 6"), ("C5", "C5 hero wavelengths (4 λ per path)"),
        ("G1", "G1 `test_prism.toml` (general forms)"), ("G2", "G2 `test_bokeh.toml`: 82 instances, no sweep table"), ("G2F", "G2F = G2 + a floor (not a reference scene)")]


def rec(t, c):
    p = os.path.join(P, "%s_bench_%s.json" % (t, c))
    return json.loads(open(p).read().strip().split("\n")[-1]) if os.path.exists(p) else None


print("| configuration | Msamples/s %s→ **%s** | ms per step | dominant kernel, µs per launch | `roofline.frac` (HBM) | counter traffic ÷ algorithmic | L2 hit rate | vector instr/s (of 600 G) | lane util. | other kernels, µs | CPU oracle, 16 threads |"
      % ((prev + " ") if prev else "", tag))
print("|---|---|---|---|---|---|---|---|---|---|---|")
for c, label in ROWS:
    d = rec(tag, c)
    if d is None:
        continue
    r = d["roofline"]; k = r["kernels"]; v = r.get("valu") or {}; l2 = r.get("l2") or {}
    dom = max(k, key=lambda n: k[n]["avg_us"] * k[n]["launches"])
    before = rec(prev, c) if prev else None
    others = ", ".join("`k_%s` %.0f" % (n, k[n]["avg_us"]) for n in ("generate", "extend", "shade", "shadow", "accumulate") if n in k and n != dom)
    print("| %s | %s**%.0f** | %.1f | `%s` %.0f | %.3f | %s | %s | %s | %s | %s | %.2f |" % (
        label, ("%.0f → " % before["value"]) if before else "", d["value"], d["ms_per_step"], v.get("kernel", "k_" + dom), k[dom]["avg_us"], r["frac"],
        ("%.2f" % (r["traffic"] / r["algorithmic_bytes_per_launch"])) if r.get("traffic") else "n/a", ("%.2f" % l2["hit_rate"]) if l2 else "n/a",
        ("%.0f G" % v["achieved"]) if v else "n/a", ("%.2f" % v["lane_utilization"]) if v else "n/a", others, d["cpu_baseline"]["value"]))
