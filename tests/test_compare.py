"""Film comparison tooling: the three modes of src/bin/compare_exr.rs:70-170 + per-channel statistics.
CPU: the oracle's restatement against numpy.  GPU: the engine against the oracle (per-pixel outputs bit for bit, statistics
to rounding), and the variance-aware agreement test at unmatched seeds."""
import numpy as np
import pytest


def images(seed=3, h=37, w=53):
    rng = np.random.default_rng(seed)
    truth = rng.uniform(0.0, 2.0, (h, w, 4)).astype(np.float32)
    image = (truth + rng.normal(0, 0.05, (h, w, 4))).astype(np.float32)
    truth[0, 0] = 0.0                      # relative error: division by zero -> 0 (compare_exr.rs:156-159)
    image[1, 1, 2] = np.nan                # counted, excluded from the statistics
    truth[2, 2, 0] = np.inf
    return image, truth


def test_oracle_modes_match_numpy(pkg, oracle):
    a = pkg.api
    image, truth = images()
    good = np.isfinite(image).all(axis=2) & np.isfinite(truth).all(axis=2)
    d = image - truth
    out, st = oracle.compare_films(image, truth, a.COMPARE_ABSOLUTE)
    assert np.array_equal(out[good], np.abs(d)[good])
    assert st.nonfinite == 2
    assert np.allclose(list(st.linf), np.abs(d[good]).max(axis=0), rtol=0, atol=0)
    assert np.allclose(list(st.mean_abs), np.abs(d[good]).astype(np.float64).mean(axis=0), rtol=1e-12)
    assert np.isclose(st.rmse, np.sqrt((d[good].astype(np.float64) ** 2).mean()), rtol=1e-12)
    out, st = oracle.compare_films(image, truth, a.COMPARE_RELATIVE)
    with np.errstate(all="ignore"):
        rel = np.abs(d) / truth
    rel[~np.isfinite(rel)] = 0.0
    assert np.array_equal(out[good], rel[good]) and (out[0, 0] == 0).all()
    out, st = oracle.compare_films(image, truth, a.COMPARE_RMSE)
    per_pixel = np.sqrt(((d[..., 0] ** 2 + d[..., 1] ** 2) + (d[..., 2] ** 2 + d[..., 3] ** 2)) / np.float32(4.0))
    assert np.isclose(st.pixel_min, per_pixel[good].min()) and np.isclose(st.pixel_max, per_pixel[good].max())
    lo, hi = np.unravel_index(np.argmin(np.where(good, per_pixel, np.inf)), good.shape), np.unravel_index(np.argmax(np.where(good, per_pixel, -np.inf)), good.shape)
    # viridis end colours: #440154 at the minimum, #fee825 at the maximum (B-spline end points interpolate the end keys)
    assert np.allclose(out[lo][:3], np.array([0x44, 0x01, 0x54]) / 255.0, atol=1e-6) and out[lo][3] == 1.0
    assert np.allclose(out[hi][:3], np.array([0xfe, 0xe8, 0x25]) / 255.0, atol=1e-6)
    assert (out[good][:, :3] >= 0).all() and (out[good][:, :3] <= 1).all()
    # the gradient itself: three pixels at t = 0, 1/2, 1; the middle of viridis is the teal key colour #26838f (the uniform
    # B-spline passes within a few percent of an interior key)
    truth3 = np.zeros((1, 3, 4), np.float32); image3 = truth3.copy()
    image3[0, 1] = 0.5; image3[0, 2] = 1.0
    out3, st3 = oracle.compare_films(image3, truth3, a.COMPARE_RMSE)
    assert st3.pixel_min == 0.0 and st3.pixel_max == 1.0
    assert np.allclose(out3[0, 1, :3], np.array([0x26, 0x83, 0x8f]) / 255.0, atol=0.03)
    assert np.allclose(out3[0, 0, :3], np.array([0x44, 0x01, 0x54]) / 255.0, atol=1e-6)


def test_oracle_identical_images(pkg, oracle):
    image, _ = images()
    image = np.nan_to_num(image, nan=0.5)
    out, st = oracle.compare_films(image, image, pkg.api.COMPARE_ABSOLUTE)
    assert (out == 0).all() and st.rmse == 0 and max(st.linf) == 0 and st.nonfinite == 0


def test_dimension_mismatch_is_rejected(pkg, oracle):
    with pytest.raises(ValueError):
        oracle.compare_films(np.zeros((4, 4, 4), np.float32), np.zeros((4, 5, 4), np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_engine_matches_oracle(pkg, engine, oracle, mode):
    image, truth = images(seed=11, h=301, w=517)
    eo, es = engine.compare_films(image, truth, mode)
    oo, os_ = oracle.compare_films(image, truth, mode)
    assert np.array_equal(eo.view(np.uint32), oo.view(np.uint32))
    assert list(es.linf) == list(os_.linf) and es.nonfinite == os_.nonfinite
    assert es.pixel_min == os_.pixel_min and es.pixel_max == os_.pixel_max
    assert np.allclose(list(es.mean_abs), list(os_.mean_abs), rtol=1e-12) and np.isclose(es.rmse, os_.rmse, rtol=1e-12)


@pytest.mark.gpu
def test_statistical_agreement_at_unmatched_seeds(pkg, engine, oracle):
    """Variance-aware check: engine and oracle rendered with DIFFERENT seeds must agree within the Monte Carlo
    error.  K independent renders each; z = (mean_e - mean_o) / sqrt(var_e/K + var_o/K) per pixel and channel."""
    a = pkg.api
    b = pkg.scene.cornell_box()
    K, n, spp = 8, 48, 32
    se, so = engine.create_scene(b), oracle.create_scene(b)
    fe = np.stack([se.render(a.render_desc(n, n, spp, 6, seed=100 + k))[0][..., :3] for k in range(K)]).astype(np.float64)
    fo = np.stack([so.render(a.render_desc(n, n, spp, 6, seed=900 + k))[0][..., :3] for k in range(K)]).astype(np.float64)
    me, mo = fe.mean(axis=0), fo.mean(axis=0)
    se2 = fe.var(axis=0, ddof=1) / K + fo.var(axis=0, ddof=1) / K
    lit = se2 > 1e-12
    z = (me - mo)[lit] / np.sqrt(se2[lit])
    assert lit.mean() > 0.9
    # t-distributed with ~2(K-1) degrees of freedom: mean 0, standard deviation ~1.08; 4096*3 samples
    assert abs(z.mean()) < 0.05, z.mean()
    assert 0.9 < z.std() < 1.3, z.std()
    assert (np.abs(z) > 6).mean() < 1e-3
    # and the same statistic through the comparison tool: RMSE of the two means is of the order of the Monte Carlo error
    pad = lambda f: np.concatenate([f, np.zeros(f.shape[:2] + (1,))], axis=2).astype(np.float32)
    _, st = engine.compare_films(pad(me), pad(mo), a.COMPARE_RMSE, want_image=False)
    assert st.rmse < 3 * np.sqrt(se2.mean() * 3 / 4)
