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


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_clustered_meshes_through_the_hierarchy(backend, seed):
    """12,000 small triangles in clusters, strips of parallelograms, spheres among them, coordinates snapped to a grid
    in two seeds of three: a hierarchy several levels deep with leaves that share edges and vertices.  rt_intersect_batch
    (free rays, rays aimed at vertices, edges and silhouettes, bounce rays) and a small render, walked through the wide
    hierarchy by the kernels and by brute force by the oracle: the same bars as test_tie_provoking_scenes."""
    pool = fz.materials_pool()
    sc, rng = fz.make_big_scene(seed, pool)
    backend.set_scene(sc)
    assert backend.scene_layout()["bvh_nodes"] > 1000
    o, d = fz.make_rays(sc, rng, 3000)
    want = h.oracle_intersect(sc, o, d)
    hit = want["obj"] >= 0
    assert 0.3 < hit.mean()
    off, tie = fz.check_hits(backend.intersect(o, d), want, o, sc)
    assert not off.any(), f"{int(off.sum())} of {len(o)} rays off"
    o2 = want["pos"][hit].astype(np.float32).astype(np.float64); g2 = want["gprim"][hit].astype(np.int32)
    d2 = fz.unit32(rng.normal(size=o2.shape))
    with h.origin_rule():
        want2 = h.oracle_intersect(sc, o2, d2, origin_prim=g2)
    off2, tie2 = fz.check_hits(backend.intersect(o2, d2, origin_prim=g2), want2, o2, sc)
    assert not off2.any(), f"{int(off2.sum())} of {len(o2)} bounce rays off"
    npx, ndiff, nwrong, nshade, nsky = fz.check_render(backend, sc, W=64, H=36, spp=2, bounces=5)
    assert nwrong == 0 and nshade == 0, f"{nwrong} paths answered differently on their own ray, {nshade} shaded differently"
    assert int(tie.sum()) + int(tie2.sum()) <= 2
