"""Random scenes and rays built to provoke ties and edge cases, shared by tests/test_gpu_fuzz.py and
scripts/fuzz_parity.py.  Scenes: vertex coordinates snapped to a coarse grid (shared edges and vertices, coplanar
triangles), exact and reversed duplicates, degenerate triangles, parallelograms (general, axis-perpendicular, sheared),
closed boxes typed exactly or with a wobble, small spheres (concentric, touching), huge spheres, mixed materials.
Rays: free, aimed at vertices / edge points / interior points, along sphere silhouettes, axis-parallel and diagonal
from grid points; all float32-representable."""
import dataclasses

import numpy as np

import helpers as h


def materials_pool():
    return np.concatenate([h.packed_scene("room").materials, h.packed_scene("random_spheres_seed7").materials])


def gprim_of(sc, obj, prim):
    """Global primitive id (sphere index, or n_spheres + triangle index) of hit (object, primitive)."""
    obj = np.asarray(obj)
    kind, index = sc.obj_kind[np.maximum(obj, 0)], sc.obj_index[np.maximum(obj, 0)]
    tri = sc.n_spheres + sc.mesh_tri_begin[np.where(kind == 1, index, 0)] + prim
    return np.where(obj < 0, -1, np.where(kind == 0, index, tri)).astype(np.int32)


def unit32(v):
    v = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-300)
    return v.astype(np.float32).astype(np.float64)


def make_scene(seed, mats_pool):
    rng = np.random.default_rng(seed)
    base = h.packed_scene("room")
    snap = [None, 0.125, 0.25, 0.5][seed % 4]          # grid the vertex coordinates are snapped to (None: general position)
    q = (lambda v: v) if snap is None else (lambda v: np.round(np.asarray(v) / snap) * snap)
    meshes = []
    n_lone = int(rng.integers(20, 160))
    tris = []
    for _ in range(n_lone):
        c = rng.uniform(-1, 1, 3)
        scale = float(rng.choice([0.02, 0.3, 1.5]))
        t = [q(c + rng.normal(scale=scale, size=3)) for _ in range(3)]
        tris.append(np.concatenate(t))
    for _ in range(int(rng.integers(0, 6))):               # exact duplicates and reversed duplicates of earlier triangles
        t = tris[int(rng.integers(len(tris)))].copy()
        tris.append(t if rng.uniform() < 0.5 else np.concatenate([t[3:6], t[0:3], t[6:9]]))
    for _ in range(int(rng.integers(0, 4))):               # degenerate: two equal vertices, or three in line
        a, b = q(rng.uniform(-1, 1, 3)), q(rng.uniform(-1, 1, 3))
        tris.append(np.concatenate([a, b, b]) if rng.uniform() < 0.5 else np.concatenate([a, b, a + 2 * (b - a)]))
    meshes.append(np.array(tris))
    quads = []
    for _ in range(int(rng.integers(4, 30))):              # parallelograms, general or axis-perpendicular, some sheared
        a = q(rng.uniform(-1, 1, 3))
        if rng.uniform() < 0.5:
            e1, e2 = q(rng.normal(scale=0.4, size=3)), q(rng.normal(scale=0.4, size=3))
        else:
            ax = int(rng.integers(3)); b_ax, c_ax = [i for i in range(3) if i != ax]
            e1, e2 = np.zeros(3), np.zeros(3)
            e1[b_ax] = q(rng.uniform(0.2, 0.8)); e2[c_ax] = q(rng.uniform(0.2, 0.8))
            if rng.uniform() < 0.4:
                e2[b_ax] = q(rng.uniform(-0.3, 0.3))
        if not np.any(e1) or not np.any(e2):
            continue
        b, c, d = a + e1, a + e1 + e2, a + e2
        quads += [np.concatenate([a, b, c]), np.concatenate([a, c, d])] if rng.uniform() < 0.5 else \
                 [np.concatenate([b, c, a]), np.concatenate([d, a, c])]
    if quads:
        meshes.append(np.array(quads))
    if rng.uniform() < 0.7:                                  # a closed box, walls typed with a wobble or exactly
        lo = q(rng.uniform(-1.5, -0.5, 3)); hi = q(rng.uniform(0.5, 1.5, 3))
        wob = float(rng.choice([0.0, 1e-6, 1e-4]))
        faces = []
        for ax in range(3):
            b_ax, c_ax = [i for i in range(3) if i != ax]
            for side, val in ((0, lo[ax]), (1, hi[ax])):
                if rng.uniform() < 0.15:
                    continue
                p = np.zeros((4, 3)); p[:, ax] = val
                p[:, b_ax] = [lo[b_ax], hi[b_ax], hi[b_ax], lo[b_ax]]; p[:, c_ax] = [lo[c_ax], lo[c_ax], hi[c_ax], hi[c_ax]]
                p = p + rng.uniform(-wob, wob, p.shape)
                faces += [np.concatenate([p[0], p[1], p[2]]), np.concatenate([p[0], p[2], p[3]])]
        if faces:
            meshes.append(np.array(faces))
    tv = np.concatenate(meshes)
    begin = np.concatenate([[0], np.cumsum([len(m) for m in meshes])]).astype(np.int32)
    aabb = np.array([np.concatenate([m.reshape(-1, 3).min(0), m.reshape(-1, 3).max(0)]) for m in meshes])
    n_mesh = len(meshes)
    n_small = int(rng.integers(0, 25))
    sph = [np.concatenate([q(rng.uniform(-1, 1, 3)), [float(rng.choice([0.01, 0.125, 0.25, 0.5]))]]) for _ in range(n_small)]
    for _ in range(int(rng.integers(0, 3))):               # concentric and touching pairs
        if sph:
            s = sph[int(rng.integers(len(sph)))]
            sph.append(np.concatenate([s[:3], [s[3] * 0.5]]) if rng.uniform() < 0.5 else np.concatenate([s[:3] + [2 * s[3], 0, 0], [s[3]]]))
    if rng.uniform() < 0.6:
        sph.append(np.array([0.0, -1001.5, 0.0, 1000.0]))   # a huge one
    if rng.uniform() < 0.2:
        sph.append(np.array([0.0, -100001.0, 0.0, 100000.0]))
    sph = np.array(sph, np.float64).reshape(-1, 4)
    n_sph = len(sph)
    kinds = np.array([1] * n_mesh + [0] * n_sph, np.int32)
    index = np.array(list(range(n_mesh)) + list(range(n_sph)), np.int32)
    mats = mats_pool[rng.integers(len(mats_pool), size=n_mesh + n_sph)]
    sc = dataclasses.replace(base, obj_kind=kinds, obj_index=index, materials=np.ascontiguousarray(mats),
                             spheres=np.ascontiguousarray(sph), sphere_obj=np.arange(n_sph, dtype=np.int32) + n_mesh,
                             mesh_aabb=np.ascontiguousarray(aabb), mesh_obj=np.arange(n_mesh, dtype=np.int32),
                             mesh_tri_begin=begin, tri_verts=np.ascontiguousarray(tv), object_names=["x"] * (n_mesh + n_sph))
    return sc, rng


def make_rays(sc, rng, n):
    """Origins and directions (float32-representable doubles), a mix of families."""
    V = sc.tri_verts.reshape(-1, 3, 3)
    fam = []
    o = rng.uniform(-1.6, 1.6, (n, 3)); d = rng.normal(size=(n, 3)); fam.append((o, d))                    # free rays
    t = V[rng.integers(len(V), size=n)]
    o = rng.uniform(-2, 2, (n, 3)); tgt = t[np.arange(n), rng.integers(3, size=n)]; fam.append((o, tgt - o))   # at vertices
    s = rng.uniform(0, 1, (n, 1)); k = rng.integers(3, size=n)
    tgt = t[np.arange(n), k] * s + t[np.arange(n), (k + 1) % 3] * (1 - s); o = rng.uniform(-2, 2, (n, 3)); fam.append((o, tgt - o))   # at edge points
    w = rng.dirichlet([1, 1, 1], size=n); tgt = (t * w[:, :, None]).sum(1); o = rng.uniform(-2, 2, (n, 3)); fam.append((o, tgt - o))   # at interior points
    if len(sc.spheres):
        P = sc.spheres[rng.integers(len(sc.spheres), size=n)]
        o = rng.uniform(-2, 2, (n, 3)); c = P[:, :3] - o; dist = np.linalg.norm(c, axis=1, keepdims=True)
        side = np.cross(c, rng.normal(size=(n, 3))); side /= np.maximum(np.linalg.norm(side, axis=1, keepdims=True), 1e-30)
        # tangent direction: angle asin(r / dist) off the centre line (only from outside)
        sin_a = np.clip(P[:, 3:4] / np.maximum(dist, 1e-30), 0, 1); cos_a = np.sqrt(1 - sin_a ** 2)
        fam.append((o, c / np.maximum(dist, 1e-30) * cos_a + side * sin_a))                                   # sphere silhouettes
    g = np.round(rng.uniform(-1.5, 1.5, (n, 3)) * 8) / 8
    ax = rng.integers(3, size=n); d = np.zeros((n, 3)); d[np.arange(n), ax] = rng.choice([-1.0, 1.0], size=n)
    two = rng.uniform(size=n) < 0.5; d[two, (ax[two] + 1) % 3] = rng.choice([-1.0, 1.0, 0.5], size=int(two.sum()))
    fam.append((g, d))                                                                                        # grid origins, axis-parallel / diagonal rays
    o = np.concatenate([f[0] for f in fam]); d = np.concatenate([f[1] for f in fam])
    ok = np.linalg.norm(d, axis=1) > 1e-12
    return o[ok].astype(np.float32).astype(np.float64), unit32(d[ok])


TOL = 1e-5


def sphere_depth_noise(sc, o):
    """Absolute uncertainty of UPSTREAM'S OWN sphere depths for rays from `o`: sphere.py:18-19 and util.py:29-32 square
    through `**`, which is libm's pow() - correct to 0.52 ulp but not correctly rounded, so one time in some dozens
    b**2 is an ulp off b*b - and b**2 - 4ac cancels: the root moves by a few ulps of |o - c| + r (1.5e-11 for the
    reference's own ground sphere of radius 1e5).  A library built WITHOUT the C library's pow tables
    (raytracing_amd/backend/build.py says so when it cannot find or verify them) squares with a multiplication, so its
    sphere depths agree with upstream's to that noise and not bit for bit; with the tables the exact code restates pow()
    and no answer needs this allowance (the campaign counts the ones that do: zero).  2^-48 (|o| + |c| + r), largest
    sphere of the scene."""
    if len(sc.spheres) == 0:
        return np.zeros(len(o))
    reach = float((np.linalg.norm(sc.spheres[:, :3], axis=1) + sc.spheres[:, 3]).max())
    return 2.0 ** -48 * (np.linalg.norm(o, axis=1) + reach)


def check_hits(got, want, o, sc):
    """Per-ray comparison of rt_intersect_batch's answers with the oracle's.  Returns (off, ties): boolean masks of the
    rays that are off (identity, or depth / normal beyond 1e-5 relative, or position beyond 1e-5 of the ray's own scale
    |o| + t) and of the rays that name another primitive, or a depth beyond 1e-5 relative, within upstream's own sphere
    depth noise (sphere_depth_noise: two depths that close are a tie no restatement can be held to, and a root of 1e-8
    is nothing but that noise)."""
    noise = sphere_depth_noise(sc, o)
    bad = (got["obj"] != want["obj"]) | (got["prim"] != want["prim"])
    both = (got["obj"] >= 0) & (want["obj"] >= 0)
    with np.errstate(invalid="ignore"):
        dt = np.where(both, np.abs(got["t"] - want["t"]), np.inf)
    tie = bad & both & (dt <= np.maximum(noise, 1e-14 * np.abs(want["t"])))
    hit = (want["obj"] >= 0) & ~bad
    rel = np.zeros(len(o)); rel[hit] = np.maximum(dt[hit] - noise[hit], 0.0) / want["t"][hit]
    tie |= hit & (dt > TOL * np.abs(want["t"])) & (rel <= TOL)
    nerr = np.zeros(len(o)); nerr[hit] = np.abs(got["normal"][hit] - want["normal"][hit]).max(axis=1)
    scale = np.linalg.norm(o, axis=1) + np.where(hit, want["t"], 0.0)
    perr = np.zeros(len(o)); perr[hit] = np.abs(got["pos"][hit] - want["pos"][hit]).max(axis=1) / np.maximum(1.0, scale[hit])
    off = (bad & ~tie) | (rel > TOL) | (nerr > TOL) | (perr > TOL)
    return off, tie


def check_render(be, sc, W=48, H=27, spp=2, bounces=5):
    """A small render against the oracle's.  Images may differ where float32 shading (a ray 1e-7 off upstream's) decides
    between coincident surfaces; what must hold is that on every path that differs, the first segment that names another
    hit is answered as the oracle answers THE KERNEL'S OWN RAY.  Returns (pixels that differ, paths that differ, paths
    whose own ray the oracle answers differently - plus pixels that differ though all their paths agree -, paths with
    equal hits but another radiance, paths with equal hits that end in the sky on another texel)."""
    be.resize(W, H); be.reset_stats()
    be.render(W, H, 0, spp, bounces, h.SEED_RENDER); be.synchronize()
    img = be.read().astype(np.float64)
    assert np.isfinite(img).all()
    with h.origin_rule():
        ref, _ = h.oracle_render(sc, W, H, spp, bounces)
    px = np.argwhere(np.abs(img - ref).max(axis=2) > 1e-3 * (1.0 + ref.max(axis=2)))
    if len(px) == 0:
        return 0, 0, 0, 0, 0
    xys = np.array([(x, y, s) for x, y in px for s in range(spp)], np.int32)
    tr = be.trace_paths(xys, W, H, bounces, h.SEED_RENDER)
    differ = wrong = shading = sky = 0
    explained = set()   # pixels with at least one path that differs: the image's difference there has a cause
    with h.origin_rule():
        for i, (x, y, s) in enumerate(xys):
            want, _, _, seg = h.oracle_trace_pixel(sc, W, H, int(x), int(y), int(s), bounces, with_segments=True)
            if np.allclose(tr["rgb"][i], want, rtol=2e-3, atol=2e-4):
                continue
            differ += 1
            explained.add((int(x), int(y)))
            ns = min(int(tr["n_seg"][i]), len(seg["obj"]))
            k = next((k for k in range(ns) if tr["obj"][i, k] != seg["obj"][k] or tr["prim"][i, k] != seg["prim"][k]), None)
            if k is None:
                # equal hits all the way: a path that ends in the sky may read the neighbouring texel (its last direction
                # is 1e-8 off the oracle's; the 1907 x 1001 image has texel borders, e.g. the column border at z = 0)
                ends_in_sky = int(tr["n_seg"][i]) == len(seg["obj"]) and seg["obj"][-1] < 0
                sky += int(ends_in_sky); shading += int(not ends_in_sky)
                continue
            origin = gprim_of(sc, tr["obj"][i, k - 1:k], tr["prim"][i, k - 1:k]) if k > 0 else None
            own = h.oracle_intersect(sc, tr["org"][i, k:k + 1], tr["dir"][i, k:k + 1].astype(np.float64), origin_prim=origin)
            if own["obj"][0] != tr["obj"][i, k] or own["prim"][0] != tr["prim"][i, k]:
                wrong += 1
    # a pixel of the rendered image that differs although every one of its paths, traced one by one, agrees with the
    # oracle's: the render kernel and the path-by-path entry would disagree with each other - counted with `wrong`
    wrong += sum(1 for x, y in px if (int(x), int(y)) not in explained)
    return len(px), differ, wrong, shading, sky


def make_big_scene(seed, mats_pool, n_tris=12000):
    """Thousands of small triangles in clusters (a hierarchy several levels deep, leaves that share edges and vertices
    when the coordinates are snapped), a few strips of parallelograms, spheres among them, one huge sphere below."""
    rng = np.random.default_rng(1_000_003 * seed + 17)
    base = h.packed_scene("room")
    snap = [None, 1.0 / 64, 1.0 / 32][seed % 3]
    q = (lambda v: v) if snap is None else (lambda v: np.round(np.asarray(v) / snap) * snap)
    n_clusters = int(rng.integers(6, 40))
    centres = rng.uniform(-1.2, 1.2, (n_clusters, 3))
    radii = rng.uniform(0.05, 0.5, n_clusters)
    which = rng.integers(n_clusters, size=n_tris)
    c = centres[which] + rng.normal(size=(n_tris, 3)) * radii[which, None]
    size = rng.choice([0.01, 0.03, 0.1], size=(n_tris, 1, 1))
    tris = q(c[:, None, :] + rng.normal(size=(n_tris, 3, 3)) * size).reshape(n_tris, 9)
    strips = []
    for _ in range(int(rng.integers(2, 8))):            # strips of parallelograms that share edges
        a = q(rng.uniform(-1, 1, 3)); e1 = q(rng.normal(scale=0.1, size=3)); e2 = q(rng.normal(scale=0.1, size=3))
        if not np.any(e1) or not np.any(e2):
            continue
        for k in range(int(rng.integers(4, 40))):
            p0 = a + k * e1
            strips += [np.concatenate([p0, p0 + e1, p0 + e1 + e2]), np.concatenate([p0, p0 + e1 + e2, p0 + e2])]
    meshes = [tris] + ([np.array(strips)] if strips else [])
    tv = np.concatenate(meshes)
    begin = np.concatenate([[0], np.cumsum([len(m) for m in meshes])]).astype(np.int32)
    aabb = np.array([np.concatenate([m.reshape(-1, 3).min(0), m.reshape(-1, 3).max(0)]) for m in meshes])
    n_mesh = len(meshes)
    sph = [np.concatenate([q(rng.uniform(-1, 1, 3)), [float(rng.choice([0.05, 0.125, 0.25]))]]) for _ in range(int(rng.integers(0, 12)))]
    sph.append(np.array([0.0, -1002.0, 0.0, 1000.0]))
    sph = np.array(sph, np.float64).reshape(-1, 4)
    n_sph = len(sph)
    kinds = np.array([1] * n_mesh + [0] * n_sph, np.int32)
    index = np.array(list(range(n_mesh)) + list(range(n_sph)), np.int32)
    mats = mats_pool[rng.integers(len(mats_pool), size=n_mesh + n_sph)]
    sc = dataclasses.replace(base, obj_kind=kinds, obj_index=index, materials=np.ascontiguousarray(mats),
                             spheres=np.ascontiguousarray(sph), sphere_obj=np.arange(n_sph, dtype=np.int32) + n_mesh,
                             mesh_aabb=np.ascontiguousarray(aabb), mesh_obj=np.arange(n_mesh, dtype=np.int32),
                             mesh_tri_begin=begin, tri_verts=np.ascontiguousarray(tv), object_names=["x"] * (n_mesh + n_sph),
                             library_bvh=True)
    return sc, rng
