"""Collects the measured spectra the benchmark scenes need into one JSON data file.

Run in the build container (reads the reference's *data* files under /root/reference/data;
never its sources):  python tools/make_scene_data.py
Output: rust-pathtracer_amd/data/spectra.json  (committed; travels to the GPU box).

  tabulated: {name: {"x": [...], "y": [...]}}   <- data/curves/csv/*.csv columns (TabulatedCSV, src/parsing/curves.rs:136-166)
  linear:    {name: {"start": s, "step": d, "y": [...]}} <- data/curves/spectra/*.spectra (Linear, src/parsing/curves.rs:168-204)
"""
import csv
import json
import os

REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(__file__), "..", "rust-pathtracer_amd", "data", "spectra.json")


def csv_columns(path, columns):
    out = {}
    rows = list(csv.reader(open(path)))
    for name, col in columns.items():
        xs, ys = [], []
        for r in rows:
            try:
                x, y = float(r[0].strip()), float(r[col].strip())
            except (ValueError, IndexError):
                continue  # header / malformed lines are skipped, as the reference does
            xs.append(x)
            ys.append(y)
        out[name] = {"x": xs, "y": ys}
    return out


def spectra(path):
    lines = [l.strip() for l in open(path).read().split("\n") if l.strip()]
    start, step = [float(v) for v in lines[0].split(",")]
    return {"start": start, "step": step, "y": [float(v) for v in lines[1:]]}


def main():
    tab = {}
    tab.update(csv_columns(f"{REF}/curves/csv/cornell.csv", {"cornell_white": 1, "cornell_green": 2, "cornell_red": 3}))
    tab.update(csv_columns(f"{REF}/curves/csv/cornell_light.csv", {"cornell_light": 1}))
    tab.update(csv_columns(f"{REF}/curves/csv/gold.csv", {"gold_n": 1, "gold_k": 2}))          # wavelength in micrometres
    tab.update(csv_columns(f"{REF}/curves/csv/copper-mcpeak.csv", {"copper_n": 1, "copper_k": 2}))
    tab.update(csv_columns(f"{REF}/curves/basis/simple-spectral-srgb-1931.csv", {"srgb_r": 1, "srgb_g": 2, "srgb_b": 3}))
    tab.update(csv_columns(f"{REF}/curves/csv/D65.csv", {"D65": 1}))                           # CIE standard illuminant D65, 1 nm, 300-830 nm (lib_curves.toml:1-5)
    lin = {"fluorescent": spectra(f"{REF}/curves/spectra/fluorescent.spectra"),
           "xenon_lamp": spectra(f"{REF}/curves/spectra/xenon_lamp.spectra")}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump({"tabulated": tab, "linear": lin}, open(OUT, "w"))
    print("wrote", os.path.abspath(OUT), {k: len(v["x"]) for k, v in tab.items()}, {k: len(v["y"]) for k, v in lin.items()})


if __name__ == "__main__":
    main()


def meshes():
    """data/meshes/*.obj -> npz (positions, normals, faces) with tobj single-index semantics."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("objmesh", os.path.join(os.path.dirname(__file__), "..", "rust-pathtracer_amd", "objmesh.py"))
    om = importlib.util.module_from_spec(spec); spec.loader.exec_module(om)
    outdir = os.path.join(os.path.dirname(OUT), "meshes")
    os.makedirs(outdir, exist_ok=True)
    for name in ("brilliant_diamond", "monkey", "gem", "prism"):
        models = om.load_obj(f"{REF}/meshes/{name}.obj")
        assert len(models) == 1, (name, len(models))
        p, n, f = models[0].arrays()
        np.savez_compressed(os.path.join(outdir, name + ".npz"), positions=p, normals=n if n is not None else np.zeros((0, 3), np.float32),
                            faces=f, material=np.array(models[0].material or ""))
        print(name, p.shape, None if n is None else n.shape, f.shape, models[0].material)


if __name__ == "__main__":
    meshes()
