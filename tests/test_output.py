"""Film output stage: output_film = tonemap + colour space + OETF + 8-bit, and the PNG / EXR writers.
CPU: the oracle's restatement against hand-computed values.  GPU: the engine against the oracle for every tonemapper /
colour space (8-bit within one code value, linear RGB within 1e-6), and the files decoded back."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest


def oracle_output(oracle, pkg, film, **kw):
    a = pkg.api
    film = np.ascontiguousarray(film, np.float32)
    h, w = film.shape[:2]
    d = a.OutputDesc(w, h, kw.get("tonemap", 0), int(kw.get("luminance_only", True)), kw.get("exposure", 0.0), kw.get("key_value", 0.18),
                     kw.get("white_point", 1.0), kw.get("colorspace", 0), kw.get("factor", 1.0))
    fn = oracle.lib.ptref_output_film
    fn.restype = C.c_int32
    fn.argtypes = [C.POINTER(a.OutputDesc), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_float)]
    rgba = np.zeros((h, w, 4), np.uint8); lin = np.zeros((h, w, 3), np.float32)
    oracle.check(fn(C.byref(d), film.ctypes.data_as(C.POINTER(C.c_float)), rgba.ctypes.data_as(C.POINTER(C.c_uint8)), lin.ctypes.data_as(C.POINTER(C.c_float))))
    return rgba, lin


def make_film(pkg, oracle, n=64):
    sc = oracle.create_scene(pkg.scene.cornell_box())
    film, _ = sc.render(pkg.api.render_desc(n, n, 8, 4))
    return film


def test_oracle_clamp_and_srgb_known_values(pkg, oracle):
    film = np.zeros((2, 2, 4), np.float32)
    film[0, 0, :3] = [0.9505, 1.0, 1.089]        # D65 white at Y = 1; Clamp clips XYZ to [0,1] first (clamp.rs:94-100), so Z -> 1
    film[0, 1, :3] = [0.2, 0.2, 0.2]
    film[1, 0, :3] = [0.0, 0.0, 0.0]
    film[1, 1, :3] = [np.nan, 1.0, 1.0]          # NaN pixel -> MAUVE (src/lib.rs:46, clamp.rs:78-80)
    rgba, lin = oracle_output(oracle, pkg, film, tonemap=pkg.api.TONEMAP_CLAMP, luminance_only=False)
    m = np.array([[3.24096994, -1.53738318, -0.49861076], [-0.96924364, 1.8759675, 0.04155506], [0.05563008, -0.20397696, 1.05697151]])
    w = m @ np.array([0.9505, 1.0, 1.0])
    enc = np.where(w < 0.0031308, 12.92 * w, 1.055 * np.maximum(w, 0) ** (1 / 2.4) - 0.055)
    assert np.abs(rgba[0, 0, :3].astype(int) - np.clip(np.ceil(enc * 255), 0, 255)).max() <= 1
    assert (rgba[1, 0, :3] == 0).all() and (rgba[..., 3] == 255).all()
    v = m @ np.array([0.2, 0.2, 0.2])
    expect = np.ceil((1.055 * v ** (1 / 2.4) - 0.055) * 255)
    assert np.abs(rgba[0, 1, :3].astype(int) - expect).max() <= 1
    assert np.allclose(lin[0, 1], v, rtol=1e-6)
    assert rgba[1, 1, 1] == 255                  # mauve is very green


def test_oracle_reinhard_maps_log_average_to_key(pkg, oracle):
    """Reinhard0: a pixel at the log-average luminance maps to key / (1 + key) (reinhard0.rs:84-103)."""
    film = make_film(pkg, oracle, 32)
    lum = film[..., 1].astype(np.float64)
    lw = np.exp(np.log(0.001 + lum).mean())
    probe = film.copy(); probe[0, 0, :3] = [lw, lw, lw]
    lw2 = np.exp(np.log(0.001 + probe[..., 1].astype(np.float64)).mean())
    rgba, lin = oracle_output(oracle, pkg, probe, tonemap=pkg.api.TONEMAP_REINHARD0, key_value=0.18, colorspace=pkg.api.COLORSPACE_REC709)
    l = 0.18 * lw / lw2
    y = l / (1 + l) * lw
    m = np.array([[3.24096994, -1.53738318, -0.49861076], [-0.96924364, 1.8759675, 0.04155506], [0.05563008, -0.20397696, 1.05697151]])
    v = m @ np.array([y, y, y])
    enc = np.where(v < 0.01805397, 4.5 * v, 1.0992968 * np.maximum(v, 0) ** 0.45 - 0.09929682)
    assert np.abs(rgba[0, 0, :3].astype(int) - np.ceil(enc * 255)).max() <= 1


CASES = [dict(tonemap=0, luminance_only=True, exposure=1.5, colorspace=0), dict(tonemap=0, luminance_only=False, exposure=-0.5, colorspace=2),
         dict(tonemap=1, luminance_only=True, key_value=0.18, colorspace=1), dict(tonemap=1, luminance_only=False, key_value=0.3, colorspace=0),
         dict(tonemap=2, luminance_only=True, key_value=0.18, white_point=1.0, colorspace=2, factor=10.0),
         dict(tonemap=2, luminance_only=False, key_value=0.18, white_point=2.0, colorspace=0)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_engine_output_matches_oracle(pkg, engine, oracle, case):
    film = make_film(pkg, oracle, 96)
    film[3, 5, 0] = np.nan; film[7, 9, 1] = np.inf
    want_rgba, want_lin = oracle_output(oracle, pkg, film, **case)
    rgba, lin = engine.output_film(film, **case)
    assert np.abs(rgba.astype(int) - want_rgba.astype(int)).max() <= 1
    assert (rgba != want_rgba).mean() < 0.02
    ok = np.isfinite(want_lin)
    assert np.allclose(lin[ok], want_lin[ok], rtol=2e-6, atol=1e-9)


def read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks, idat = 8, {}, b""
    while pos < len(data):
        n, t = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(t + body) & 0xffffffff)
        if t == b"IDAT": idat += body
        else: chunks[t] = body
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    raw = zlib.decompress(idat)
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 4 * w)
    assert (rows[:, 0] == 0).all() and depth == 8 and ctype == 6
    return rows[:, 1:].reshape(h, w, 4), chunks


@pytest.mark.gpu
def test_png_and_exr_files(pkg, engine, oracle, tmp_path):
    film = make_film(pkg, oracle, 40)[:, :33]     # non-square
    rgba, lin = engine.output_film(film, tonemap=pkg.api.TONEMAP_REINHARD1, key_value=0.18, white_point=1.0, colorspace=pkg.api.COLORSPACE_REC2020)
    png, exr = str(tmp_path / "beauty.png"), str(tmp_path / "beauty.exr")
    engine.write_png(png, rgba, pkg.api.COLORSPACE_REC2020)
    engine.write_exr(exr, lin, pkg.api.COLORSPACE_REC2020)
    back, chunks = read_png(png)
    assert np.array_equal(back, rgba)
    assert struct.unpack(">I", chunks[b"gAMA"])[0] == round(100000 / 2.4)            # effective_gamma, mod.rs:201-203
    assert struct.unpack(">8I", chunks[b"cHRM"]) == (31270, 32900, 70800, 29200, 29200, 17000, 13100, 4600)  # REC2020 as in mod.rs:85-92
    d = open(exr, "rb").read()
    assert struct.unpack("<II", d[:8]) == (20000630, 2)
    h, w = lin.shape[:2]
    assert b"chromaticities\0" in d and b"channels\0chlist\0" in d
    end = d.index(b"screenWindowWidth\0float\0") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1
    offs = np.frombuffer(d[end:end + 8 * h], "<u8")
    for y in (0, h // 2, h - 1):
        yy, sz = struct.unpack("<ii", d[offs[y]:offs[y] + 8])
        assert yy == y and sz == 12 * w
        row = np.frombuffer(d[offs[y] + 8:offs[y] + 8 + sz], "<f4").reshape(3, w)
        assert np.array_equal(row[2], lin[y, :, 0]) and np.array_equal(row[1], lin[y, :, 1]) and np.array_equal(row[0], lin[y, :, 2])
