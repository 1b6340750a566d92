#!/usr/bin/env python3
"""compare_films — the reference's compare_exr tool (src/bin/compare_exr.rs) for this repository's films.

  tools/compare_films.py --compare-file a.npy --ground-truth-file b.npy --output-file diff [--mode absolute_difference|rmse|relative]

Inputs: .npy float32 [H,W,4] (raw XYZ films as returned by pt_render) or the uncompressed RGB .exr files pt_write_exr writes.
Output: <output-file>.npy (the difference image), for --mode rmse also <output-file>.png (viridis, as the reference), and the
statistics (per-channel L-inf, mean |diff|, RMSE, the per-pixel min/max the reference prints) as one JSON line.
Runs on the GPU through libptamd.so (pt_compare_films)."""
import argparse
import importlib
import json
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def read_exr(path):
    """Reader for the single-part uncompressed scanline float RGB files of pt_write_exr."""
    data = open(path, "rb").read()
    if struct.unpack_from("<I", data, 0)[0] != 20000630:
        raise ValueError("%s: not an OpenEXR file" % path)
    pos, attrs = 8, {}
    while data[pos] != 0:
        end = data.index(b"\0", pos); name = data[pos:end].decode(); pos = end + 1
        end = data.index(b"\0", pos); typ = data[pos:end].decode(); pos = end + 1
        size = struct.unpack_from("<i", data, pos)[0]; pos += 4
        attrs[name] = (typ, data[pos:pos + size]); pos += size
    pos += 1
    if attrs["compression"][1] != b"\0":
        raise ValueError("%s: only uncompressed files are supported" % path)
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, p, ch = [], 0, attrs["channels"][1]
    while ch[p] != 0:
        end = ch.index(b"\0", p); names.append(ch[p:end].decode()); p = end + 1 + 16
    offsets = struct.unpack_from("<%dQ" % h, data, pos)
    img = np.zeros((h, w, 4), np.float32)
    for y in range(h):
        o = offsets[y] + 8
        for k, nme in enumerate(names):  # channels are stored alphabetically: B, G, R
            row = np.frombuffer(data, np.float32, w, o + 4 * w * k)
            img[y, :, {"R": 0, "G": 1, "B": 2, "A": 3}[nme]] = row
    return img


def load(path):
    if path.endswith(".npy"):
        a = np.load(path).astype(np.float32)
        if a.ndim == 3 and a.shape[2] == 3:
            a = np.concatenate([a, np.zeros(a.shape[:2] + (1,), np.float32)], axis=2)
        return a
    if path.endswith(".exr"):
        return read_exr(path)
    raise ValueError("unsupported file type: " + path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--compare-file", required=True)
    ap.add_argument("--ground-truth-file", required=True)
    ap.add_argument("--output-file", required=True)
    ap.add_argument("--mode", default="absolute_difference")
    o = ap.parse_args()
    pkg = importlib.import_module("rust-pathtracer_amd")
    lib = pkg.load()
    mode = {"rmse": pkg.api.COMPARE_RMSE, "relative": pkg.api.COMPARE_RELATIVE}.get(o.mode, pkg.api.COMPARE_ABSOLUTE)  # compare_exr.rs:45-52
    try:
        image, truth = load(o.compare_file), load(o.ground_truth_file)
    except (OSError, ValueError) as e:
        print("failed to parse images for some reason. check whether the paths exist (%s)" % e)
        return 1
    out, st = lib.compare_films(image, truth, mode)
    base = o.output_file[:-4] if o.output_file.endswith((".exr", ".npy", ".png")) else o.output_file
    np.save(base + ".npy", out)
    if mode == pkg.api.COMPARE_RMSE:
        print("minmax: %s -> %s" % (st.pixel_min, st.pixel_max))
        rgba = (np.clip(out, 0, 1) * 255.0).astype(np.uint8)  # (r * 255.0) as u8, compare_exr.rs:137
        rgba[..., 3] = 255
        lib.write_png(base + ".png", rgba)
    print(json.dumps(st.as_dict()))
    print("saved, exiting")
    return 0


if __name__ == "__main__":
    sys.exit(main())
