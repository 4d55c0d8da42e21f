"""GPU tests of the bounding-volume-hierarchy path: it must return exactly what the linear table
walk and upstream's loops return (hit object and face EXACT, depth / normal to 1e-5 relative).

Scene: upstream's room plus upstream's other model asset (chess_horse.obj, 288 faces, placed with
upstream's own setters; tests/golden/scene_room_horse.npz and the ray dump rays_room_horse.npz were
produced by the upstream code).  Larger meshes are the same surface with every face split into four
by its edge midpoints, 2 / 4 / 6 times (4,608 ... 1,179,648 faces); for those the expected answers
come from the CPU oracle's brute-force loop on the same rays.
"""
import dataclasses

import numpy as np
import pytest

import helpers as h
from test_gpu_parity import _check_hits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    from raytracing_amd.backend.lib import HipBackend

    be = HipBackend(0)
    yield be
    be.close()


def _path_rays(dump="rays_room_horse.npz"):
    z, _ = h.load_npz(dump)
    o32 = z["org"].astype(np.float32).astype(np.float64)
    d32 = z["dir"].astype(np.float32).astype(np.float64)
    return z, o32, d32


def _random_rays(n, seed, scale=1.2):
    rng = np.random.default_rng(seed)
    o = (rng.uniform(-1, 1, (n, 3)) * scale).astype(np.float32).astype(np.float64)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    return o, d


def test_upstream_dump_double_rays(backend):
    """Upstream's own Scene.intersect answers on the room + horse scene (double rays: exact code)."""
    sc = h.packed_scene("room_horse")
    backend.set_scene(sc)
    z, _, _ = _path_rays()
    got = backend.intersect(z["org"], z["dir"])
    _check_hits(got, dict(obj=z["obj"], prim=z["prim"], t=z["t"], normal=z["normal"], pos=z["pos"]))


@pytest.mark.parametrize("mode", ["linear", "library_bvh", "host_bvh"])
def test_small_mesh_all_search_modes(backend, mode):
    """310 triangles through the linear walk, a hierarchy built inside rt_set_scene, and one built
    by the caller with rt_build_bvh: same answers as the oracle, which equals upstream on this
    scene (tests/test_oracle_golden.py)."""
    sc = h.packed_scene("room_horse")
    if mode == "library_bvh":
        sc = dataclasses.replace(sc, library_bvh=True)
    elif mode == "host_bvh":
        from raytracing_amd.backend.bvh import build_flat_bvh

        nodes, order = build_flat_bvh(sc.tri_verts)
        sc = dataclasses.replace(sc, bvh_nodes=nodes, bvh_tri_index=order)
    if mode == "linear":
        with h.linear_tables():          # (left to itself the library takes the hierarchy above 48 triangles)
            backend.set_scene(sc)
    else:
        backend.set_scene(sc)
    z, o32, d32 = _path_rays()
    want = h.oracle_intersect(sc, o32, d32)
    backend.reset_stats()
    got = backend.intersect(o32, d32)
    _check_hits(got, want)
    # bounce rays tagged with the primitive they leave, under the same rule in the oracle
    exact = h.oracle_intersect(sc, z["org"], z["dir"])
    origin = np.full(len(o32), -1, np.int32)
    origin[1:] = np.where(z["seg"][1:] > 0, exact["gprim"][:-1], -1)
    with h.origin_rule():
        want = h.oracle_intersect(sc, o32, d32, origin_prim=origin)
    _check_hits(backend.intersect(o32, d32, origin_prim=origin), want)
    o, d = _random_rays(20000, 11)
    _check_hits(backend.intersect(o, d), h.oracle_intersect(sc, o, d))


def test_axis_parallel_and_far_rays_through_hierarchy(backend):
    """Zero direction components (upstream's 1e10 pseudo-inverse in the box gate) and origins far
    outside the scene, with the hierarchy in use."""
    sc = dataclasses.replace(h.packed_scene("room_horse"), library_bvh=True)
    backend.set_scene(sc)
    o = np.array([[0.5, -0.5, -2.0], [0.5, 0.5, 0.25], [0.6, -0.7, -2.0], [0.45, 3.0, 0.25], [40.0, 30.0, -80.0], [0.5, -0.6, 0.3]])
    d = np.array([[0, 0, 1.0], [0, -1.0, 0], [0, 0, 1.0], [0, -1.0, 0], [-0.4, -0.3, 0.8], [1.0, 0, 0]])
    d[4] /= np.linalg.norm(d[4])
    d = d.astype(np.float32).astype(np.float64)
    _check_hits(backend.intersect(o, d), h.oracle_intersect(sc, o, d))


def test_zero_direction_components_through_hierarchy(backend):
    """Rays with one or two direction components exactly zero, in every sign combination, from origins inside
    and outside the mesh's box and from origins whose coordinates EQUAL a vertex coordinate (the ray then runs
    in a box plane of some node: 0 x inf in the walk's slab test).  The walk may only ever visit too many
    boxes, never too few: hits equal the oracle's (which gates on upstream's 1e10 pseudo-inverse, mesh.py:70-74)."""
    base = h.packed_scene("room_horse")
    sc = base.with_subdivided_mesh(base.n_meshes - 1, 2)
    backend.set_scene(sc)
    rng = np.random.default_rng(77)
    n = 6000
    verts = np.asarray(sc.tri_verts, np.float64).reshape(-1, 3)
    mesh_verts = verts[3 * 22:]                                   # the subdivided horse (the room's 22 triangles come first)
    o = (rng.uniform(-1, 1, (n, 3)) * 1.2)
    pick = mesh_verts[rng.integers(0, len(mesh_verts), n)]
    on_plane = rng.random((n, 3)) < 0.35                          # some coordinates taken from a mesh vertex
    o = np.where(on_plane, pick, o).astype(np.float32).astype(np.float64)
    d = rng.normal(size=(n, 3))
    zero = rng.random((n, 3)) < 0.5
    zero[zero.all(axis=1), 0] = False                             # at least one component stays
    d[zero] = 0.0
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32).astype(np.float64)
    neg_zero = rng.random((n, 3)) < 0.5                           # -0.0 as well as +0.0
    d = np.where((d == 0.0) & neg_zero, -0.0, d)
    # and components that are tiny instead of zero, down to float32 denormals, either sign
    tiny = (d == 0.0) & (rng.random((n, 3)) < 0.3)
    mag = (10.0 ** -rng.integers(10, 45, (n, 3)).astype(np.float64)).astype(np.float32).astype(np.float64)
    d = np.where(tiny, np.where(neg_zero, -mag, mag), d)
    assert (d == 0.0).any(axis=1).mean() > 0.5 and ((d != 0.0) & (np.abs(d) < 1e-9)).any(axis=1).mean() > 0.2
    _check_hits(backend.intersect(o, d), h.oracle_intersect(sc, o, d))


@pytest.mark.parametrize("levels,n_path,n_rand", [(2, 10004, 20000), (4, 6000, 6000), (6, 1500, 1000)])
def test_subdivided_mesh_against_oracle(backend, levels, n_path, n_rand):
    """4.6 k, 74 k and 1.18 M triangles (the largest is BASELINE config 4's mesh size): the library
    builds the hierarchy (more than 4096 primitives), the oracle loops over every triangle."""
    base = h.packed_scene("room_horse")
    sc = base.with_subdivided_mesh(base.n_meshes - 1, levels)
    assert sc.n_tris == 22 + 288 * 4 ** levels
    backend.set_scene(sc)
    z, o32, d32 = _path_rays()
    o32, d32 = o32[:n_path], d32[:n_path]
    backend.reset_stats()
    got = backend.intersect(o32, d32)
    _check_hits(got, h.oracle_intersect(sc, o32, d32))
    o, d = _random_rays(n_rand, 100 + levels)
    _check_hits(backend.intersect(o, d), h.oracle_intersect(sc, o, d))
    # size-independent property: the subdivided surface is the same surface up to the rounding of
    # the new midpoints, so object and depth agree with the 310-triangle scene for all but the rays
    # that graze an edge or start on the surface (upstream's ~1e-8 self-hits come and go)
    coarse = h.oracle_intersect(base, o32, d32)
    same = got["obj"] == coarse["obj"]
    assert same.mean() > 0.99
    both = same & (got["obj"] >= 0)
    assert (np.abs(got["t"][both] - coarse["t"][both]) > 1e-5 * np.maximum(coarse["t"][both], 1e-3)).mean() < 0.01


def test_walk_of_last_resort(backend, monkeypatch):
    """A lane whose stack of wide nodes runs full walks the threaded two-child hierarchy without a stack instead.
    With the stack cut to six entries (RT_SCENE_TINY_WALK_STACK) most walks that enter the 74 k-triangle mesh
    do: hits still equal the oracle's, closest-hit entry and whole paths alike, and with the full stack
    (the builder reports what the deepest walk can need) the same rays never take that road."""
    base = h.packed_scene("room_horse")
    sc = base.with_subdivided_mesh(base.n_meshes - 1, 4)
    z, o32, d32 = _path_rays()
    o32, d32 = o32[:6000], d32[:6000]
    o, d = _random_rays(4000, 77)
    want_path, want_rand = h.oracle_intersect(sc, o32, d32), h.oracle_intersect(sc, o, d)
    backend.set_scene(sc)
    assert backend.scene_layout()["walk_stack_need"] > 6
    before = backend.walk_fallbacks()
    _check_hits(backend.intersect(o32, d32), want_path)
    assert backend.walk_fallbacks() == before, "the full stack overflowed on the test's rays"
    monkeypatch.setenv("RAYTRACING_AMD_TINY_WALK_STACK", "1")
    backend.set_scene(sc)
    monkeypatch.delenv("RAYTRACING_AMD_TINY_WALK_STACK")
    _check_hits(backend.intersect(o32, d32), want_path)
    _check_hits(backend.intersect(o, d), want_rand)
    assert backend.walk_fallbacks() > before + 100, "the tiny stack never overflowed: the test exercises nothing"
    # whole paths through the render path's own device functions
    W, H, b = 64, 36, 3
    rng = np.random.default_rng(5)
    xys = np.stack([rng.integers(20, 60, 60), rng.integers(2, 30, 60), np.zeros(60, np.int64)], 1).astype(np.int32)
    tr = backend.trace_paths(xys, W, H, b, h.SEED_RENDER)
    bad = 0
    with h.origin_rule():
        for i, (x, y, s_) in enumerate(xys):
            want = h.oracle_trace_pixel(sc, W, H, int(x), int(y), int(s_), b)[0]
            bad += not np.allclose(tr["rgb"][i], want, rtol=2e-3, atol=2e-4)
    assert bad <= 1, f"{bad} of {len(xys)} paths differ"
    backend.set_scene(sc)   # (the module's other tests get the full stack back)


def test_render_hierarchy_equals_linear_walk(backend):
    """Same image from the linear walk and from the hierarchy (310 triangles, both supported):
    identical hits give identical paths; only which hits get a double-precision refinement may
    differ, which moves radiance by float32 rounding.  Bound: 1e-3 relative RMSE, measured ~1e-5."""
    W, H, spp, b = 128, 72, 8, 6
    imgs = []
    for lib_bvh in (False, True):
        sc = dataclasses.replace(h.packed_scene("room_horse"), library_bvh=lib_bvh)
        if lib_bvh:
            backend.set_scene(sc)
        else:
            with h.linear_tables():
                backend.set_scene(sc)
        backend.resize(W, H)
        backend.reset_stats()
        backend.render(W, H, 0, spp, b, h.SEED_RENDER)
        backend.synchronize()
        imgs.append(backend.read())
        st = backend.stats()
        assert (st["bvh_node_tests"] > 0) == lib_bvh
    rmse = np.sqrt(np.mean((imgs[0] - imgs[1]) ** 2)) / np.sqrt(np.mean(imgs[0] ** 2))
    differing = (np.abs(imgs[0] - imgs[1]).max(axis=2) > 1e-4 * spp).mean()
    assert rmse < 1e-3 and differing < 2e-3, (rmse, differing)


def test_render_large_mesh_against_oracle_paths(backend):
    """1.18 M-triangle scene rendered through the hierarchy: per-path radiance of sampled pixels
    against the oracle (origin rule) - the oracle is brute force, so only a few hundred paths."""
    base = h.packed_scene("room_horse")
    sc = base.with_subdivided_mesh(base.n_meshes - 1, 6)
    backend.set_scene(sc)
    W, H, b = 64, 36, 3
    rng = np.random.default_rng(3)
    xys = np.stack([rng.integers(20, 60, 160), rng.integers(2, 30, 160), np.zeros(160, np.int64)], 1).astype(np.int32)
    tr = backend.trace_paths(xys, W, H, b, h.SEED_RENDER)
    bad = 0
    with h.origin_rule():
        for i, (x, y, s) in enumerate(xys):
            want = h.oracle_trace_pixel(sc, W, H, int(x), int(y), int(s), b)[0]
            if not np.allclose(tr["rgb"][i], want, rtol=2e-3, atol=2e-4):
                bad += 1
    assert bad <= 2, f"{bad} of {len(xys)} paths differ"


def test_render_rectangle_of_large_mesh_against_oracle(backend):
    """rt_render ITSELF at 1.18 M triangles - the lean kernel with its sliced, resumable walk of the wide hierarchy,
    the per-lane stack and pending leaves in LDS, the set-aside paths and the launch that finishes them - against the
    oracle's brute-force render of the same rectangle (origin rule, same sample sequence).  The rectangle is
    put on the mesh's silhouette, where paths graze the dense mesh.  Pixels agree to float32 rounding; at most
    two may differ (a path that runs along an edge of the room can pick the other face with float32 origins)."""
    base = h.packed_scene("room_horse")
    sc = base.with_subdivided_mesh(base.n_meshes - 1, 6)
    backend.set_scene(sc)
    assert backend.scene_layout()["bvh_nodes"] > 500_000
    W, H, b, spp, rw, rh = 96, 54, 8, 4, 24, 14
    # where the camera sees the mesh: the window with mesh and room pixels in equal parts
    xys = np.array([(x, y, 0) for x in range(W) for y in range(H)], np.int32)
    first = backend.trace_paths(xys, W, H, 0, h.SEED_RENDER)["obj"][:, 0].reshape(W, H)
    on_mesh = (first == int(sc.mesh_obj[-1])).astype(np.int64)
    csum = np.pad(on_mesh.cumsum(0).cumsum(1), ((1, 0), (1, 0)))
    cnt = csum[rw:, rh:] - csum[:-rw, rh:] - csum[rw:, :-rh] + csum[:-rw, :-rh]
    x0, y0 = np.unravel_index(np.argmin(np.abs(cnt - rw * rh // 2)), cnt.shape)
    rect = (int(x0), int(x0) + rw, int(y0), int(y0) + rh)
    assert 0.2 * rw * rh < cnt[x0, y0] < 0.8 * rw * rh, f"no window on the mesh's silhouette: best holds {cnt[x0, y0]} mesh pixels"
    backend.resize(W, H)
    backend.reset_stats()
    backend.render(W, H, 0, spp, b, h.SEED_RENDER, rect=rect)
    backend.synchronize()
    img = backend.read().astype(np.float64)
    st, cnts = backend.stats(), backend.render_counters()
    with h.origin_rule():
        ref, rays = h.oracle_render(sc, W, H, spp, b, rect=rect)
    inside = np.zeros((W, H), bool)
    inside[rect[0]:rect[1], rect[2]:rect[3]] = True
    assert not img[~inside].any() and np.isfinite(img).all()
    off = (np.abs(img - ref).max(axis=2) > 1e-3 * (1.0 + ref.max(axis=2))) & inside
    print(f"\nrectangle {rect}: {int(off.sum())} of {rw * rh} pixels differ; rays {st['rays']} (oracle {rays}), "
          f"boxes per ray {st['bvh_node_tests'] / st['rays']:.1f}, set aside {cnts['paths_set_aside']} paths")
    assert abs(st["rays"] - rays) <= 16 and st["bvh_node_tests"] > 0
    assert off.sum() <= 2, f"{int(off.sum())} pixels differ from the oracle"


def test_bad_hierarchy_is_rejected(backend):
    from raytracing_amd.backend.bvh import build_flat_bvh
    from raytracing_amd.backend.lib import BackendError

    sc = h.packed_scene("room_horse")
    nodes, order = build_flat_bvh(sc.tri_verts)
    shrunk = nodes.copy()
    leaf = int(np.nonzero(shrunk["tri"] >= 0)[0][5])
    shrunk["hi"][leaf] = shrunk["lo"][leaf]            # a leaf box that no longer holds its triangles
    with pytest.raises(BackendError, match="hierarchy"):
        backend.set_scene(dataclasses.replace(sc, bvh_nodes=shrunk, bvh_tri_index=order))
    looped = nodes.copy()
    looped["skip"][3] = 1                              # a backward link would never terminate
    with pytest.raises(BackendError, match="hierarchy"):
        backend.set_scene(dataclasses.replace(sc, bvh_nodes=looped, bvh_tri_index=order))
    dup = order.copy()
    dup[0] = dup[1]                                    # a triangle listed twice, another missing
    with pytest.raises(BackendError, match="hierarchy"):
        backend.set_scene(dataclasses.replace(sc, bvh_nodes=nodes, bvh_tri_index=dup))
    backend.set_scene(dataclasses.replace(sc, bvh_nodes=nodes, bvh_tri_index=order))   # the good one is accepted


def test_progressive_batches_and_device_tonemap(backend):
    """Shape of BASELINE config 5 at test size: room + large mesh rendered progressively (sample
    batches added into one device buffer, which is a torch tensor), then tone-mapped on the device
    with upstream's main.py chain (exposure 2, brightness 0.02, contrast 1.1).  Batches must add up
    to the one-shot render (same global sample indices -> same paths; float32 summation order is
    the only difference), and the device tone-map must equal the numpy one on the read-back."""
    import torch

    from raytracing_amd.post_processing import PostProcessing

    base = h.packed_scene("room_horse")
    backend.set_scene(base.with_subdivided_mesh(base.n_meshes - 1, 4))
    W, H, spp, b = 160, 90, 12, 6
    one = torch.zeros((W, H, 3), dtype=torch.float32, device="cuda:0")
    backend.render(W, H, 0, spp, b, 77, accum_ptr=one.data_ptr())
    backend.synchronize()
    prog = torch.zeros_like(one)
    previews = []
    for s0 in range(0, spp, 4):
        backend.render(W, H, s0, 4, b, 77, accum_ptr=prog.data_ptr())
        backend.synchronize()
        previews.append((prog / float(s0 + 4)).flip(1))     # RenderResult.img_arr of the passes so far
    torch.cuda.synchronize()
    a, p = one.cpu().numpy(), prog.cpu().numpy()
    assert np.allclose(a, p, rtol=2e-6, atol=1e-6), np.abs(a - p).max()
    pp = PostProcessing(exposure=2, brightness=0.02, contrast=1.1)
    dev = pp.process(previews[-1])
    assert dev.is_cuda
    host = pp.process(previews[-1].cpu().numpy().astype(np.float64))
    assert np.abs(dev.cpu().numpy() - host).max() < 2e-5
    assert float(previews[0].mean()) > 0 and len(previews) == 3


def test_config5_at_full_size_eight_rank_shares(backend):
    """BASELINE config 5 AT SIZE on one GPU: room + the 1.18 M-triangle mesh at 3840x2160, 1024 spp,
    8 bounces.  The eight rank shares `shard_samples(1024, r, 8)` are rendered progressively (batches of
    16 samples) into eight buffers - what eight GPUs would hold before the merge - and summed in rank
    order, which is what the all-reduce computes; the sum must equal ONE 1024-spp render of the same
    frame (same global sample indices -> same paths; float32 summation order is the only difference).
    Every value finite and non-negative.  Then the device tone-map of upstream's main.py chain
    (exposure 2, brightness 0.02, contrast 1.1) straight from the merged buffer against the numpy
    chain on the read-back, 2e-5 absolute."""
    import torch

    from raytracing_amd.distributed import shard_samples
    from raytracing_amd.post_processing import PostProcessing

    base = h.packed_scene("room_horse")
    backend.set_scene(base.with_subdivided_mesh(base.n_meshes - 1, 6))
    W, H, spp, b, seed, ranks = 3840, 2160, 1024, 8, 2025, 8
    stream = torch.cuda.current_stream().cuda_stream
    one = torch.zeros((W, H, 3), dtype=torch.float32, device="cuda:0")
    backend.reset_stats()
    backend.render(W, H, 0, spp, b, seed, accum_ptr=one.data_ptr(), stream=stream)
    backend.synchronize(stream)
    st_one = backend.stats()
    merged = torch.zeros_like(one)
    share = torch.zeros_like(one)
    backend.reset_stats()
    for r in range(ranks):
        begin, count = shard_samples(spp, r, ranks)
        share.zero_()
        for s0 in range(begin, begin + count, 16):
            backend.render(W, H, s0, min(16, begin + count - s0), b, seed, accum_ptr=share.data_ptr(), stream=stream)
        backend.synchronize(stream)
        merged += share                      # rank order, as a ring all-reduce's partial sums would not be - hence the tolerance
    torch.cuda.synchronize()
    st_shares = backend.stats()
    assert st_one["paths"] == st_shares["paths"] == W * H * spp
    assert st_one["rays"] == st_shares["rays"]                      # the same paths, segment for segment
    assert bool(torch.isfinite(merged).all()) and float(merged.min()) >= 0.0
    scale = float(one.max())
    err = float((merged - one).abs().max())
    rel = float(((merged - one).abs() / one.clamp_min(1e-3 * scale)).max())
    print(f"\nconfig 5 at size: {st_one['rays']} rays per render, max |sum of shares - one render| = {err:.3e} (sums up to {scale:.1f}), relative {rel:.3e}")
    assert rel <= 2e-5, rel
    # tone-map on the device, straight from the merged sums (pass count and display flip folded in)
    pp = PostProcessing(exposure=2, brightness=0.02, contrast=1.1)
    dev = torch.empty_like(merged)
    backend.postprocess(W, H, dev.data_ptr(), src_ptr=merged.data_ptr(), scale=1.0 / spp, flip_y=True, exposure=2.0, brightness=0.02,
                        contrast=1.1, stream=stream)
    torch.cuda.synchronize()
    host = pp.process((merged.cpu().numpy().astype(np.float64) / spp)[:, ::-1])
    assert np.abs(dev.cpu().numpy() - host).max() < 2e-5
    del one, merged, share, dev


def test_wavefront_form_against_oracle(backend, monkeypatch):
    """The wavefront form - lean extend kernel, settle kernel for the queries float32 cannot close, material-sorted
    shade kernel, per-sample resolve - held to the ORACLE, not to the single kernel: a rectangle on the silhouette
    of the 4,608-triangle mesh, rendered in three batches, against the oracle's brute-force render of the same
    rectangle (origin rule, same sample sequence).  Pixels agree to float32 rounding except where a path lands ON an
    edge between two walls: in this rectangle the mirror reflections off the sphere do, for four pixels, by the
    scene's symmetry.  The hit there is the oracle's hit, segment by segment (scripts/wf_diag.py), and the next ray
    starts from the double-precision position of that hit (float32 + residue, carried through the queues), which
    brings three of the four into line; the fourth reaches the edge by a float32 reflection 1e-7 off the oracle's.
    Bound: 2 of 896 (measured 1)."""
    base = h.packed_scene("room_horse")
    sc = base.with_subdivided_mesh(base.n_meshes - 1, 2)
    backend.set_scene(sc)
    W, H, b, spp = 96, 54, 8, 6
    rect = (40, 72, 8, 36)
    monkeypatch.setenv("RAYTRACING_AMD_WAVEFRONT", "1")
    monkeypatch.setenv("RAYTRACING_AMD_WF_PATHS", "65536")
    backend.resize(W, H)
    backend.reset_stats()
    backend.render(W, H, 0, spp, b, h.SEED_RENDER, rect=rect)
    backend.synchronize()
    monkeypatch.delenv("RAYTRACING_AMD_WAVEFRONT")
    monkeypatch.delenv("RAYTRACING_AMD_WF_PATHS")
    img = backend.read().astype(np.float64)
    st = backend.stats()
    with h.origin_rule():
        ref, rays = h.oracle_render(sc, W, H, spp, b, rect=rect)
    inside = np.zeros((W, H), bool)
    inside[rect[0]:rect[1], rect[2]:rect[3]] = True
    first = backend.trace_paths(np.array([(x, y, 0) for x in range(rect[0], rect[1]) for y in range(rect[2], rect[3])], np.int32), W, H, 0,
                                h.SEED_RENDER)["obj"][:, 0]
    on_mesh = (first == int(sc.mesh_obj[-1])).mean()
    assert 0.1 < on_mesh < 0.9, f"the rectangle misses the mesh's silhouette: {on_mesh:.2f} of its pixels see the mesh"
    assert not img[~inside].any() and np.isfinite(img).all()
    off = (np.abs(img - ref).max(axis=2) > 1e-3 * (1.0 + ref.max(axis=2))) & inside
    print(f"\nwavefront rectangle {rect}: {int(off.sum())} of {int(inside.sum())} pixels differ; rays {st['rays']} (oracle {rays})")
    assert off.sum() <= 2, f"{int(off.sum())} pixels differ from the oracle"
    # (a path that picks the other face at an edge goes on differently: a few rays more or fewer)
    assert abs(st["rays"] - rays) <= max(16, 0.002 * rays) and st["bvh_node_tests"] > 0, (st["rays"], rays)


def test_wavefront_form_traces_the_same_paths(backend, monkeypatch):
    """The optional wavefront form (generate / extend / shade / resolve kernels) runs the same device
    functions as the single render kernel, so images agree up to float32 summation order and the
    statistics agree (test counts of the walk to within a percent)."""
    base = h.packed_scene("room_horse")
    backend.set_scene(base.with_subdivided_mesh(base.n_meshes - 1, 3))
    W, H, spp, b = 96, 54, 6, 5
    out = []
    for wf in (False, True):
        if wf:
            monkeypatch.setenv("RAYTRACING_AMD_WAVEFRONT", "1")
            monkeypatch.setenv("RAYTRACING_AMD_WF_PATHS", "65536")     # several batches
        backend.resize(W, H)
        backend.reset_stats()
        backend.render(W, H, 2, spp, b, 11)
        backend.synchronize()
        out.append((backend.read(), backend.stats()))
    monkeypatch.delenv("RAYTRACING_AMD_WAVEFRONT")
    monkeypatch.delenv("RAYTRACING_AMD_WF_PATHS")
    (a, sa), (w, sw) = out
    assert np.allclose(a, w, rtol=1e-5, atol=1e-6), np.abs(a - w).max()
    for k in ("rays", "paths", "sphere_tests", "exact_rays", "refined_hits", "sky_lookups", "shaded_hits"):
        assert sa[k] == sw[k], (k, sa[k], sw[k])
    # how many boxes and leaf triangles a walk tests depends a little on where its slices end (a leaf
    # met at the end of a slice is tested before or after the boxes that follow it, which moves the
    # pruning bound): the two forms cut their walks differently; and the default form walks the query of a path it
    # sets aside twice (once in the lean kernel, once in the full one)
    for k in ("tri_tests", "bvh_node_tests"):
        assert abs(sa[k] - sw[k]) <= 0.02 * sa[k], (k, sa[k], sw[k])


def test_refit_for_moved_mesh_equals_rebuild(backend):
    """Moved meshes (upstream: the mesh setters, mesh.py:178-209): with RT_SCENE_REFIT the library keeps the
    topology and leaf order of the hierarchy it built for the same meshes and only refits the boxes.
    On the 74 k-triangle mesh, turned, shifted and bent: the refitted scene's hits are those of a scene
    built from scratch and of the oracle (identity exact), bounce rays included; the time of both
    uploads is printed.  A scene with other meshes is built anew however the flag is set."""
    import time

    base = h.packed_scene("room_horse")
    sc0 = base.with_subdivided_mesh(base.n_meshes - 1, 4)
    m = base.n_meshes - 1
    b0, b1 = int(sc0.mesh_tri_begin[m]), int(sc0.mesh_tri_begin[m + 1])
    assert b1 - b0 > 70000

    def moved(sc, angle, shift, bend):
        tv = sc.tri_verts.copy()
        p = tv[b0:b1].reshape(-1, 3)
        c = p.mean(0)
        ca, sa = np.cos(angle), np.sin(angle)
        R = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]])
        q = (p - c) @ R.T
        q[:, 0] += bend * q[:, 1] ** 2          # not a rigid motion: the boxes really have to be refitted
        p[:] = q + c + np.asarray(shift)
        aabb = sc.mesh_aabb.copy()
        aabb[m] = np.concatenate([p.min(0), p.max(0)])
        return dataclasses.replace(sc, tri_verts=np.ascontiguousarray(tv), mesh_aabb=aabb)

    backend.set_scene(dataclasses.replace(sc0, library_bvh=True))
    assert not backend.scene_layout()["refitted"]
    o, d = _random_rays(8000, 77, scale=1.0)
    _, po, pd = _path_rays()
    o, d = np.concatenate([o, po[:4000]]), np.concatenate([d, pd[:4000]])
    for angle, shift, bend in [(0.4, (0.05, 0.0, -0.1), 0.0), (2.2, (-0.2, 0.02, 0.1), 0.3)]:
        sc1 = dataclasses.replace(moved(sc0, angle, shift, bend), library_bvh=True)
        t0 = time.perf_counter(); backend.set_scene(sc1, refit=True); t_refit = time.perf_counter() - t0
        lay = backend.scene_layout()
        assert lay["refitted"] and lay["bvh_nodes"] > 10000
        got = backend.intersect(o, d)
        hit = got["obj"] >= 0
        o2 = got["pos"][hit].astype(np.float32).astype(np.float64)
        g2 = got["gprim"][hit].astype(np.int32)
        d2 = np.random.default_rng(5).normal(size=o2.shape)
        d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
        got2 = backend.intersect(o2, d2, origin_prim=g2)
        t0 = time.perf_counter(); backend.set_scene(sc1); t_build = time.perf_counter() - t0
        assert not backend.scene_layout()["refitted"]
        fresh, fresh2 = backend.intersect(o, d), backend.intersect(o2, d2, origin_prim=g2)
        for a, b in ((got, fresh), (got2, fresh2)):
            assert np.array_equal(a["obj"], b["obj"]) and np.array_equal(a["prim"], b["prim"])
            assert np.array_equal(a["t"], b["t"])              # the same hit record, bit for bit
        want = h.oracle_intersect(sc1, o, d)
        _check_hits(got, want)
        with h.origin_rule():
            _check_hits(got2, h.oracle_intersect(sc1, o2, d2, origin_prim=g2))
        assert (want["obj"] == sc1.mesh_obj[m]).sum() > 300
        print(f"\n{b1 - b0} triangles moved: upload with refit {1e3 * t_refit:.0f} ms, with a new hierarchy {1e3 * t_build:.0f} ms "
              f"(ratio {t_refit / t_build:.2f})")
        backend.set_scene(sc1, refit=True)                     # the topology kept is the one built last
        assert backend.scene_layout()["refitted"]
    # other meshes: nothing to refit
    backend.set_scene(dataclasses.replace(base.with_subdivided_mesh(m, 3), library_bvh=True), refit=True)
    assert not backend.scene_layout()["refitted"]
    _check_hits(backend.intersect(o[:4000], d[:4000]), h.oracle_intersect(base.with_subdivided_mesh(m, 3), o[:4000], d[:4000]))
