"""Property tests (hypothesis) of the neighbour searches: grid == brute force on adversarial point sets
(points on cell faces, lattice spacing equal to the radius, duplicates, empty cells).  CPU: the oracle's grid;
the GPU variant lives in test_gpu_hypothesis below and is marked gpu."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

import oracle


def _points(draw_seed, n, mode, radius):
    rng = np.random.default_rng(draw_seed)
    if mode == 0:      # uniform
        pts = rng.random((n, 3)) * rng.choice([3.0, 12.0, 60.0])
    elif mode == 1:    # lattice with spacing == radius (pairs exactly at the cut-off, points on cell faces)
        g = np.arange(0, 6) * radius
        pts = np.array([[x, y, z] for x in g for y in g[:3] for z in g[:2]])[:n]
        pts = pts + rng.choice([0.0, 0.0, 1e-6, -1e-6], size=pts.shape)
    elif mode == 2:    # few distinct sites, many duplicates
        sites = rng.random((max(n // 8, 1), 3)) * 9.0
        pts = sites[rng.integers(0, len(sites), n)]
    else:              # two far clusters + an outlier
        pts = np.concatenate([rng.normal(0, 2.0, (n // 2, 3)), rng.normal(0, 2.0, (n - n // 2, 3)) + 500.0, [[-300.0, 40.0, 7.0]]])
    return np.ascontiguousarray(pts, np.float32)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(2, 220), mode=st.integers(0, 3),
       radius=st.sampled_from([5.0, 6.0, 2.5, 1.0, 7.5]), masked=st.booleans())
def test_oracle_grid_equals_brute(seed, n, mode, radius, masked):
    pts = _points(seed, n, mode, radius)
    act = (np.random.default_rng(seed + 1).random(len(pts)) < 0.7).astype(np.uint8) if masked else None
    gi, gj, cand = oracle.search_all(pts, radius, active=act, grid=True)
    bi, bj, _ = oracle.search_all(pts, radius, active=act, grid=False)
    assert np.array_equal(gi, bi) and np.array_equal(gj, bj)
    assert cand >= len(gi)


@pytest.mark.gpu
def test_gpu_search_all_hypothesis():
    from arpeggio_amd import _capi
    from helpers import tiny_complex
    ctx = _capi.Context(0)

    @settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.too_slow])
    @given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(2, 400), mode=st.integers(0, 3),
           radius=st.sampled_from([5.0, 6.0, 2.5, 1.0, 7.5]), masked=st.booleans())
    def prop(seed, n, mode, radius, masked):
        pts = _points(seed, n, mode, radius)
        act = (np.random.default_rng(seed + 1).random(len(pts)) < 0.7).astype(np.uint8) if masked else None
        ctx.set_complex(tiny_complex(pts))
        gi, gj = ctx.search_all(radius, active=act)
        bi, bj, _ = oracle.search_all(pts, radius, active=act, grid=False)
        assert np.array_equal(gi, bi) and np.array_equal(gj, bj)

    try:
        prop()
    finally:
        ctx.close()
