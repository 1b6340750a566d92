import ctypes as C

import numpy as np


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def numerics(oracle, which, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x) if y is None else np.ascontiguousarray(y, np.float32)
    out = np.zeros_like(x)
    oracle.lib.ptref_numerics(which, x.size, fptr(x), fptr(y), fptr(out))
    return out


def unit_sphere(rng, n):
    """math::random::random_on_unit_sphere on uniform samples (src/props.rs:16-20)."""
    u = rng.random(n, dtype=np.float32)
    v = rng.random(n, dtype=np.float32)
    phi = u * 2 * np.pi
    z = v * 2 - 1
    r = np.sqrt(np.maximum(0.0, 1 - z * z))
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1).astype(np.float32)


def film_metrics(a, b):
    """The three comparison modes of the reference's src/bin/compare_exr.rs:39-52 + per-channel L-inf."""
    a = np.asarray(a, np.float64)[..., :3]
    b = np.asarray(b, np.float64)[..., :3]
    d = np.abs(a - b)
    return {"linf": float(d.max()), "linf_channels": [float(d[..., c].max()) for c in range(3)],
            "rmse": float(np.sqrt((d * d).mean())),
            "relative": float((d / np.maximum(np.abs(b), 1e-6)).max())}
