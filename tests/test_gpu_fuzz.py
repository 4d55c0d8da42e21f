"""A few seeds of the fuzz campaign (scripts/fuzz_parity.py runs it at length) as a regular GPU test."""
import dataclasses

import numpy as np
import pytest

import fuzz_scenes as fz
import helpers as h

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    from raytracing_amd.backend.lib import HipBackend
    return HipBackend(0)


@pytest.mark.parametrize("first", [0, 1000])
def test_tie_provoking_scenes(backend, first):
    """Scenes snapped to a grid, with duplicated, degenerate and coplanar triangles, boxes, touching spheres; rays
    aimed at vertices, edges and silhouettes, axis-parallel rays from grid points.  Every answer of
    rt_intersect_batch equals the oracle's (identity exact; depth, normal to 1e-5, position to 1e-5 of |o| + t)
    through the linear tables and through a hierarchy; the one exemption is fz.sphere_depth_noise.  The small
    renders may differ from the oracle's where float32 shading decides between coincident surfaces, but never on a
    ray of the kernel's own that the oracle answers differently, and with equal hits only by a neighbouring sky texel."""
    pool = fz.materials_pool()
    rays = ties = 0
    for seed in range(first, first + 8):
        sc, rng = fz.make_scene(seed, pool)
        o, d = fz.make_rays(sc, rng, 3000)
        want = h.oracle_intersect(sc, o, d)
        hit = want["obj"] >= 0
        o2 = want["pos"][hit].astype(np.float32).astype(np.float64); g2 = want["gprim"][hit].astype(np.int32)
        d2 = fz.unit32(rng.normal(size=o2.shape))
        with h.origin_rule():
            want2 = h.oracle_intersect(sc, o2, d2, origin_prim=g2)
        for mode in ("linear", "hierarchy"):
            if mode == "linear":
                with h.linear_tables():
                    backend.set_scene(sc)
            else:
                backend.set_scene(dataclasses.replace(sc, library_bvh=True))
            for got, w, oo in ((backend.intersect(o, d), want, o), (backend.intersect(o2, d2, origin_prim=g2), want2, o2)):
                off, tie = fz.check_hits(got, w, oo, sc)
                assert not off.any(), f"seed {seed} {mode}: {int(off.sum())} of {len(oo)} rays off"
                rays += len(oo); ties += int(tie.sum())
            npx, ndiff, nwrong, nshade, nsky = fz.check_render(backend, sc)
            assert nwrong == 0 and nshade == 0, f"seed {seed} {mode}: {nwrong} paths answered differently on their own ray, {nshade} shaded differently"
    assert ties <= 4 and rays > 100000
