#!/usr/bin/env python3
"""How far the oracle's film moves under each ALTERNATIVE reading of the un-vendored `math` / `rust_optics` crates.

The oracle (oracle/ptref.cpp) restates crates that are not under /root/reference and cannot be checked against them here ("parity
unpinned").  This script tells a maintainer who has the crates which of the restated choices to check first: it builds one private
copy of the oracle per alternative (`-DPTREF_ALT_<NAME>`, compiled into a scratch directory — the test oracle itself is never built with
any of them), renders the same scenes with the same seeds, and reports

  * matched seeds, BASELINE C1 (Cornell 256 x 256, 16 spp, depth 4): L-inf and RMSE of the XYZ film against the default reading — whether
    the alternative changes paths at all, and how far a single pixel can move;
  * converged means (Cornell 64 x 64 at 512 spp; gem and HDRI scenes at 256 spp): the relative change of the film's mean X, Y, Z and of
    its mean chromaticity (x, y) — what survives averaging, i.e. a bias a reference render would show.

CPU only (test infrastructure): python tools/oracle_sensitivity.py [--out profiles/r3_oracle_sensitivity.md] [--threads 8]
"""
import argparse
import importlib
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALTERNATIVES = [
    ("CUBIC_CATMULL_ROM", "`InterpolationMode::Cubic` = uniform Catmull-Rom through four knots (restated: Hermite with zero tangents between two)"),
    ("TABULATED_ZERO_OUTSIDE", "`Tabulated::evaluate` outside its knots = 0 (restated: the end value)"),
    ("LINEAR_CLAMP", "`Linear::evaluate` outside its bounds = the end samples (restated: 0)"),
    ("BLACKBODY_MEAN_VISIBLE", "`Blackbody` with boost normalised by its mean over 380-750 nm (restated: by its Wien peak)"),
    ("XYZ_F32", "colour-matching fit evaluated in f32 (restated: f64, rounded once)"),
    ("XYZ_SINGLE_LOBE", "the single-lobe Wyman-Sloan-Shirley fit (restated: the multi-lobe one)"),
    ("FRAME_FRISVAD", "`TangentFrame::from_normal` = Frisvad 2012 (restated: Duff et al. 2017)"),
    ("COSINE_CONCENTRIC", "`random_cosine_direction` by the concentric disk mapping (restated: polar, phi = 2 pi u, r = sqrt v)"),
    ("COSINE_SWAP_UV", "`random_cosine_direction` with the roles of u and v swapped"),
    ("SPHERE_Z_FROM_X", "`random_on_unit_sphere` with z from the first sample"),
    ("CHOOSE_LE", "`Sample1D::choose` with `<=`"),
    ("CHOOSE_NO_RESCALE", "`Sample1D::choose` without rescaling the sample"),
    ("POWER_BETA1", "`power_heuristic` with beta = 1 (restated: 2)"),
    ("UV_Y_UP", "`uv_to_direction` / `direction_to_uv` with the polar axis along +y (restated: +z)"),
    ("APERTURE_POLAR", "circular aperture by polar disk sampling (restated: rejection from the square)"),
]
FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-unsafe-math-optimizations", "-pthread", "-w", "-shared"]


def build(workdir, name):
    out = os.path.join(workdir, "libptref_%s.so" % (name or "default"))
    subprocess.check_call(["g++"] + FLAGS + (["-DPTREF_ALT_" + name] if name else []) + ["-o", out, os.path.join(ROOT, "oracle", "ptref.cpp")])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 4)
    ap.add_argument("--quick", action="store_true", help="a tenth of the samples (smoke test of the script)")
    args = ap.parse_args()
    pkg = importlib.import_module("rust-pathtracer_amd")
    import oracle_loader
    q = 10 if args.quick else 1
    scenes = [  # name, builder, (w, h, spp, depth, L)
        ("C1 matched seeds", "cornell_box", (256, 256, max(1, 16 // q), 4, 2)),
        ("Cornell converged", "cornell_box", (64, 64, 512 // q, 4, 2)),
        ("gem (GGX, sharp light)", "cornell_gem", (64, 36, 256 // q, 8, 2)),
        ("HDRI + sphere", "hdri_small", (48, 48, 256 // q, 4, 3)),
        ("sun + metal (blackbody)", "sun_test", (48, 48, 256 // q, 4, 2)),
    ]
    with tempfile.TemporaryDirectory() as work:
        names = [""] + [a for a, _ in ALTERNATIVES]
        with ThreadPoolExecutor(max_workers=min(8, args.threads)) as ex:
            libs = dict(zip(names, ex.map(lambda n: build(work, n), names)))
        films = {}
        for n in names:
            lib = pkg.api.Library(libs[n], "ptref_", optional=("render_device", "device_info"))
            lib.lib.ptref_render_mt.restype = oracle_loader.C.c_int32
            lib.lib.ptref_render_mt.argtypes = [oracle_loader.C.c_void_p, oracle_loader.C.POINTER(pkg.api.RenderDesc), oracle_loader.C.POINTER(oracle_loader.C.c_float),
                                                oracle_loader.C.POINTER(pkg.api.Profile), oracle_loader.C.c_uint32]
            for label, sname, (w, h, spp, depth, L) in scenes:
                sc = lib.create_scene(pkg.scene.SCENES[sname]())
                film, _ = oracle_loader.render_mt(lib, sc, pkg.api.render_desc(w, h, spp, depth, light_samples=L, seed=1), args.threads)
                films[(n, label)] = film[..., :3].astype(np.float64)
                if n == "":   # the noise floor of the converged means: the default reading at another seed
                    film2, _ = oracle_loader.render_mt(lib, sc, pkg.api.render_desc(w, h, spp, depth, light_samples=L, seed=2), args.threads)
                    films[("SEED2", label)] = film2[..., :3].astype(np.float64)
                sc.close()

    def chroma(m):
        return m[0] / m.sum(), m[1] / m.sum()
    lines = ["# Oracle sensitivity to the restated `math` / `rust_optics` choices (tools/oracle_sensitivity.py)", "",
             "Every row is the oracle built with ONE alternative reading (`-DPTREF_ALT_<NAME>`, a private copy: the test oracle is never built this way)",
             "against the default reading, same scenes, same seeds.  `C1 L-inf / RMSE`: BASELINE C1 (Cornell 256 x 256, 16 spp, depth 4) at matched seeds —",
             "0 means the alternative never changes a path or a value on this scene.  `d mean Y`, `d(x, y)`: relative change of the film's mean luminance and",
             "absolute change of its mean chromaticity on converged renders — a bias that a reference render would show, where it exceeds the row NOISE FLOOR",
             "(the default reading at another seed: what sampling noise alone does to these means; the gem's caustics make its mean the noisiest).  An alternative that changes values but no path (curve interpolation, colour matching, blackbody, MIS weights, the uv axis) is compared at matched seeds on identical paths: its difference is exact, far below the noise floor of a single render; one that changes paths (cosine mapping, tangent frame, aperture) shows a difference of the size of the noise floor, i.e. no bias beyond it.  Read it as a checklist order: the rows with the largest converged change are the ones to verify first against the crates.", "",
             "| alternative reading | C1 L-inf | C1 RMSE | Cornell d mean Y | Cornell d(x, y) | gem d mean Y | HDRI d mean Y | HDRI d(x, y) | sun d mean Y |", "|---|---|---|---|---|---|---|---|---|"]
    rows = []
    for name, text in ALTERNATIVES + [("SEED2", "NOISE FLOOR: the default reading itself at another seed")]:
        ref, alt = films[("", "C1 matched seeds")], films[(name, "C1 matched seeds")]
        linf = float(np.abs(alt - ref).max()); rmse = float(np.sqrt(((alt - ref) ** 2).mean()))
        cells = []
        score = 0.0
        for label in ("Cornell converged", "gem (GGX, sharp light)", "HDRI + sphere", "sun + metal (blackbody)"):
            m0, m1 = films[("", label)].reshape(-1, 3).mean(0), films[(name, label)].reshape(-1, 3).mean(0)
            dy = (m1[1] - m0[1]) / m0[1]
            c0, c1 = chroma(m0), chroma(m1)
            dxy = max(abs(c1[0] - c0[0]), abs(c1[1] - c0[1]))
            cells.append((dy, dxy))
            score = max(score, abs(dy), 20 * dxy)
        rows.append((score, "| %s (`%s`) | %.3g | %.3g | %+.2f %% | %.1e | %+.2f %% | %+.2f %% | %.1e | %+.2f %% |" %
                     (text, name, linf, rmse, 100 * cells[0][0], cells[0][1], 100 * cells[1][0], 100 * cells[2][0], cells[2][1], 100 * cells[3][0])))
    lines += [r for _, r in sorted(rows, key=lambda x: -x[0])]
    m = films[("", "Cornell converged")].reshape(-1, 3).mean(0)
    lines += ["", "Default reading, Cornell converged: mean XYZ = (%.5f, %.5f, %.5f), chromaticity (%.4f, %.4f)." % (m[0], m[1], m[2], *chroma(m))]
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        open(os.path.join(ROOT, args.out), "w").write(text)


if __name__ == "__main__":
    main()
