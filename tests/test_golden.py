"""The committed golden vectors (tests/golden, made by tools/make_golden.py) pin the oracle itself: a rebuilt
oracle must reproduce them bit for bit.  (The reference's own tests hold no numeric vectors for this path —
the properties they do assert are replayed in tests/test_oracle.py.)"""
import numpy as np

import parity_suite as ps


def test_oracle_reproduces_golden_films(oracle):
    for name in ps.GOLDEN_RENDERS:
        film, prof, ref, counters = ps.golden_render(oracle, name)
        assert np.array_equal(film, ref), name
        assert (prof.camera_rays, prof.bounce_rays, prof.shadow_rays, prof.env_hits) == tuple(int(c) for c in counters)


def test_oracle_reproduces_golden_hits_and_materials(oracle):
    for scene in ("cornell_box", "mixed_primitives", "cornell_gem"):
        got, want = ps.golden_hits(oracle, scene)
        ps.assert_hits_equal(got, want)
        ps.golden_materials(oracle, scene)
