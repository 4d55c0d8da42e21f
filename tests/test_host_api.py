"""CPU tests of the host side: value types, OBJ loading, packing, the C ABI's symbol table, the
post-processing chain, and that nothing renders without the HIP library / a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers as h
from raytracing_amd import core, mesh, post_processing, raytracing
from raytracing_amd.backend import lib as rtlib
from raytracing_amd.backend.packer import PackedScene, RtSceneDesc, pack_scene


def test_vec3_interface():
    V = core.Vec3
    a, b = V(1, 2, 3), V(4)
    assert tuple(a + b) == (5, 6, 7) and tuple(a - b) == (-3, -2, -1) and tuple(-a) == (-1, -2, -3)
    assert a * b == 24 and tuple(a * 2) == (2, 4, 6) and tuple(2 * a) == (2, 4, 6) and tuple(a / 2) == (0.5, 1, 1.5)
    assert tuple(np.float64(2.0) * a) == (2, 4, 6)       # numpy scalar * Vec3 broadcasts over the components
    assert tuple(a.cross(b)) == (2 * 4 - 3 * 4, 3 * 4 - 1 * 4, 1 * 4 - 2 * 4)
    assert tuple(a.prod(b)) == (4, 8, 12) and tuple(a.elementwise() * V(1, 0, 1)) == (1, 0, 3)
    assert abs(V(3, 4, 0).normalize().magnitude() - 1) < 1e-15
    with pytest.raises(ValueError):
        V(0).normalize()
    assert tuple(V(0.6, -0.8, 0).reflect(V(0, 2, 0))) == pytest.approx((0.6, 0.8, 0))
    assert tuple(V(1, 0, 0).rotate(90, V(0, 0, 1))) == pytest.approx((0, 1, 0))
    c = V(-1, 2, -3)
    d = c.absolute()
    assert tuple(c) == (1, 2, 3) and tuple(d) == (1, 2, 3) and d is not c   # mutates, then copies
    arr = np.zeros((2, 2, 3))
    arr[0, 1] += a
    assert tuple(arr[0, 1]) == (1, 2, 3)


def test_shading_helpers_match_upstream_known_answers():
    for e in h.kat()["shade"]:
        d, n = core.Vec3(e["d"]), core.Vec3(e["n"])
        assert raytracing.util.schlick_approximation(-d, n, e["ior"]) == pytest.approx(e["schlick"], rel=1e-13)
        assert tuple(raytracing.util.refract(d, n, e["eta"])) == pytest.approx(tuple(e["refract"]), rel=1e-12, abs=1e-15)
        assert tuple(d.reflect(n)) == pytest.approx(tuple(e["reflect"]), rel=1e-12, abs=1e-15)
    for e in h.kat()["sky"]:
        assert raytracing.util.vec_to_sky_coords(e["d"]) == pytest.approx(tuple(e["uv"]), abs=1e-15)


def test_camera_matches_upstream_known_answers():
    cam = raytracing.Camera(core.Vec3(0, 0, -2), np.pi / 2, core.Vec3(0))
    assert cam.d == 1.0000000000000002 and tuple(cam.right) == (1, 0, 0) and tuple(cam.up) == (0, 1, 0)
    for e in h.kat()["camera_room_1080p"]:
        ray = cam.get_ray(e["u"], e["v"])
        assert tuple(ray.direction) == pytest.approx(tuple(e["d"]), abs=1e-15)


def test_obj_loader_semantics(tmp_path):
    p = tmp_path / "two.obj"
    p.write_text("# c\nmtllib x.mtl\no first\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\ns 0\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"
                 "o second\nv 0 0 1\nv 1 0 1\nv 0 1 1\nf 5//1 6//1 7//1\n")
    ms = mesh.Mesh.load_from_obj_file(str(p))
    assert [m.name for m in ms] == ["first", "second"]
    assert ms[0].triangles.tolist() == [[0, 1, 2], [0, 2, 3]]          # fan triangulation
    assert ms[1].triangles.tolist() == [[0, 1, 2]]                     # indices rebased per object
    assert ms[1].transformed_vertices.tolist() == [[0, 0, 1], [1, 0, 1], [0, 1, 1]]
    assert tuple(ms[0].bounding_box[0]) == (0, 0, 0) and tuple(ms[0].bounding_box[1]) == (1, 1, 0)
    assert ms[0].material.ior == 1.45


def test_mesh_transform_order():
    m = mesh.Mesh(core.Material.default_material(), core.Vec3(10, 0, 0), [(1, 0, 0), (0, 1, 0), (0, 0, 1)], [(0, 1, 2)],
                  scale=core.Vec3(2, 1, 1), rotation=core.Vec3(0, 0, 90))
    # scale (2,0,0) -> rotate 90 deg about z -> (0,2,0) -> translate
    assert m.transformed_vertices[0] == pytest.approx((10, 2, 0))
    m.set_origin(core.Vec3(0))
    assert m.transformed_vertices[0] == pytest.approx((0, 2, 0))
    m2 = mesh.Mesh(core.Material.default_material(), core.Vec3(1, 1, 1), [(3, 0, 0), (0, 3, 0), (0, 0, 3)], [(0, 1, 2)])
    before = m2.transformed_vertices.copy()
    m2.set_origin_to_center_of_mass()
    assert tuple(m2.origin) == (2, 2, 2) and np.allclose(m2.vertices.sum(axis=0), 0)
    assert np.allclose(m2.transformed_vertices, before)   # unscaled, unrotated: shape stays put


def test_packer_reproduces_upstream_scene_tables():
    """Scenes built with THIS package's classes pack to the very tables packed from upstream's
    objects (fixtures), object order and all - when the asset files are available."""
    assets = "/root/reference/assets"
    if not os.path.isdir(assets):
        pytest.skip("upstream assets are only mounted in the build container")
    import random

    from raytracing_amd import scenes

    ref = h.packed_scene("room")
    got = pack_scene(scenes.room(assets))
    for f in ("obj_kind", "obj_index", "materials", "spheres", "mesh_aabb", "mesh_tri_begin", "tri_verts", "sky_rgb"):
        assert np.array_equal(getattr(got, f), getattr(ref, f)), f
    assert got.object_names == ref.object_names and got.sky_scale == 0.4 and got.cam_d == ref.cam_d
    np.random.seed(7)
    random.seed(7)
    ref = h.packed_scene("random_spheres_seed7")
    got = pack_scene(scenes.random_spheres(assets))
    assert np.array_equal(got.spheres, ref.spheres) and np.array_equal(got.materials, ref.materials)
    assert (got.cam_dof_strength, got.cam_dof_dist) == (0.01, 10)


def test_packer_duck_typing_and_desc_layout():
    class O:  # a stand-in for an upstream object: only attributes are read
        pass

    s = O(); s.material = core.Material.default_material(); s.origin = (1, 2, 3); s.radius = 2.0
    sc = O(); sc.objects = [s]; sc.camera = raytracing.Camera(core.Vec3(0), 1.0, core.Vec3(0, 0, 1))
    p = pack_scene(sc)
    assert p.n_spheres == 1 and p.n_tris == 0 and p.sky_rgb is None
    d = p.desc()
    assert d.struct_size == C.sizeof(RtSceneDesc) and d.n_objects == 1 and d.spheres[3] == 2.0
    with pytest.raises(TypeError):
        sc.objects = [object.__new__(O)]
        sc.objects[0].material = s.material
        pack_scene(sc)


def test_packed_scene_roundtrip(tmp_path):
    ref = h.packed_scene("room")
    ref.save(str(tmp_path / "s.npz"))
    back = PackedScene.load(str(tmp_path / "s.npz"))
    assert np.array_equal(back.tri_verts, ref.tri_verts) and np.array_equal(back.sky_rgb, ref.sky_rgb)
    assert back.cam_forward == ref.cam_forward and back.object_names == ref.object_names


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports exactly what include/raytracing_amd.h declares."""
    header = open(os.path.join(h.REPO, "include", "raytracing_amd.h")).read()
    declared = set(re.findall(r"^(?:int|void \*|const char \*)\s*(rt_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared == set(rtlib.EXPORTS)
    handle = rtlib.load_library()
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.rt_abi_version() == 3
    assert C.sizeof(rtlib.RtRenderParams) == 56


def test_no_cpu_fallback():
    """Without a GPU the product path fails loudly instead of rendering on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rtlib.BackendError) as e:
        rtlib.HipBackend(0)
    assert e.value.status == 4
    r = raytracing.Renderer()
    with pytest.raises(rtlib.BackendError):
        r.init_render(raytracing.RenderSettings(raytracing.Scene(), 8, 8, 1, 1))
    with pytest.raises(rtlib.BackendError):   # per-ray tracing goes to the GPU as well
        r.raytrace(core.Ray(core.Vec3(0), core.Vec3(0, 0, 1)))
    import raytracing_amd

    for root, _, files in os.walk(os.path.dirname(raytracing_amd.__file__)):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                assert "oracle" not in open(os.path.join(root, f)).read().lower().replace("the cpu oracle", ""), f


def test_post_processing_matches_upstream():
    z, _ = h.load_npz("postproc.npz")
    pp = post_processing.PostProcessing(exposure=2, brightness=0.02, contrast=1.1, saturation=0.9, gamma=1.2)
    assert np.allclose(pp.process(z["img"]), z["out"], rtol=1e-12, atol=1e-14)
    import torch

    t = pp.process(torch.from_numpy(z["img"]))
    assert np.allclose(t.numpy(), z["out"], rtol=1e-10, atol=1e-12)


def test_render_result_views():
    r = raytracing.RenderResult(4, 3, 2)
    r.add_to_px(1, 2, core.Vec3(2, 4, 6))
    r.add_to_px(9, 9, core.Vec3(1))
    r.rendered_passes = 2
    r.update_views()
    assert tuple(r.img_arr[1, 0]) == (1, 2, 3) and tuple(r.img_arr_clipped[1, 0]) == (1, 1, 1)
    assert r.rendered_pixels == 1 and r.progress == pytest.approx(1 / 12 / 2)


def test_native_bvh_builder_output_is_a_valid_hierarchy():
    """rt_build_bvh runs on the CPU (no GPU needed): every triangle in exactly one leaf, boxes
    contain their triangles and nest, links point forward - checked by an independent numpy
    restatement of the acceptance rules."""
    from raytracing_amd.backend.bvh import NODE_DTYPE, build_flat_bvh, check_flat_bvh

    base = h.packed_scene("room_horse")
    for levels, leaf in ((0, 4), (0, 1), (3, 4), (3, 15)):
        sc = base.with_subdivided_mesh(base.n_meshes - 1, levels)
        assert sc.n_tris == 22 + 288 * 4 ** levels
        nodes, order = build_flat_bvh(sc.tri_verts, leaf_size=leaf)
        assert nodes.dtype == NODE_DTYPE and sorted(order.tolist()) == list(range(sc.n_tris))
        check_flat_bvh(nodes, order, sc.tri_verts)
        leaves = nodes["tri"] >= 0
        assert (nodes["tri"][leaves] & 15).max() <= 15 and int(leaves.sum()) * 2 - 1 == len(nodes)   # binary tree
    # degenerate input: coincident centroids, zero-area and non-finite triangles still give a valid tree
    tv = np.zeros((40, 9))
    tv[:, 3] = 1.0; tv[:, 7] = 1.0
    tv[5] = 0.0
    tv[6, 4] = np.inf
    nodes, order = build_flat_bvh(tv)
    assert sorted(order.tolist()) == list(range(40)) and len(nodes) >= 1
    # the checker itself rejects a broken tree
    bad = nodes.copy()
    bad["skip"][0] = 0
    with pytest.raises(ValueError):
        check_flat_bvh(bad, order, np.where(np.isfinite(tv), tv, 0.0))


def test_wide_hierarchy_is_the_same_hierarchy():
    """rt_collapse_bvh (CPU): the four-wide form the kernels walk, checked by an independent walk in numpy - from the
    root every wide node is reached exactly once through the first-child arithmetic, the leaf ranges are exactly
    the two-child hierarchy's leaves, every child's [c - h, c + h] contains the boxes of the two-child nodes it
    stands for, the per-octant orders are permutations that put nearer centres first, and the reported stack need
    is what the deepest walk can push."""
    from raytracing_amd.backend.bvh import build_flat_bvh, collapse_flat_bvh

    base = h.packed_scene("room_horse")
    for levels, leaf in ((0, 2), (2, 2), (3, 1), (3, 6)):
        sc = base.with_subdivided_mesh(base.n_meshes - 1, levels)
        nodes, order = build_flat_bvh(sc.tri_verts, leaf_size=leaf)
        wide, need = collapse_flat_bvh(nodes)
        leaves2 = sorted(int(t) for t in nodes["tri"][nodes["tri"] >= 0])
        # the two-child boxes by leaf range, for the containment check of leaf children
        box_of_leaf = {int(nd["tri"]): (nd["lo"].astype(np.float64), nd["hi"].astype(np.float64)) for nd in nodes if nd["tri"] >= 0}
        seen = np.zeros(len(wide), np.int64)
        got_leaves, deepest = [], 0
        stack = [(0, 0, None)]   # (wide node, stack entries in use when the walk reaches it, box the parent holds for it)
        while stack:
            w, pending, held = stack.pop()
            seen[w] += 1
            nd = wide[w]
            inner, used = int(nd["inner"]) & 15, (int(nd["inner"]) >> 4) & 15
            assert inner & ~used == 0 and used in (1, 3, 7, 15), (w, inner, used)   # slots are filled from the front
            c = np.stack([nd["cx"], nd["cy"], nd["cz"]], 1).astype(np.float64)
            hh = np.stack([nd["hx"], nd["hy"], nd["hz"]], 1).astype(np.float64)
            n_inner = bin(inner).count("1")
            deepest = max(deepest, pending + n_inner)
            lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
            for k in range(4):
                if not (used >> k) & 1:
                    assert np.all(hh[k] < 0)            # an empty slot can never be hit
                    continue
                lo, hi = c[k] - hh[k], c[k] + hh[k]
                lo_all, hi_all = np.minimum(lo_all, lo), np.maximum(hi_all, hi)
                if (inner >> k) & 1:
                    child = int(nd["first"]) + bin(inner & ((1 << k) - 1)).count("1")
                    stack.append((child, pending + n_inner - 1, (lo, hi)))
                else:
                    rng = int(nd["link"][k])
                    got_leaves.append(rng)
                    blo, bhi = box_of_leaf[rng]
                    assert np.all(lo <= blo) and np.all(hi >= bhi), (w, k)
            if held is not None:
                # what the parent holds for this node contains everything this node holds for its children (to the
                # rounding of the centres: a few units in the last place of the scene size)
                slack = 4e-7 * max(1.0, np.abs(c[used.bit_length() - 1]).max())
                assert np.all(held[0] <= lo_all + slack) and np.all(held[1] >= hi_all - slack), w
            for oct_ in range(8):
                byte = (int(nd["order"][oct_ >> 2]) >> (8 * (oct_ & 3))) & 0xFF
                perm = [(byte >> (2 * j)) & 3 for j in range(4)]
                assert sorted(perm) == [0, 1, 2, 3]
                sign = np.array([-1.0 if (oct_ >> a) & 1 else 1.0 for a in range(3)])
                keys = [float(np.dot(sign, c[k])) for k in perm if (used >> k) & 1]
                assert keys == sorted(keys), (w, oct_, perm, keys)
        assert np.all(seen == 1) and sorted(got_leaves) == leaves2
        assert need == deepest and need >= 1
    # one leaf only: the root holds it
    tv = np.zeros((1, 9)); tv[0, 3] = tv[0, 7] = 1.0
    nodes, order = build_flat_bvh(tv)
    wide, need = collapse_flat_bvh(nodes)
    assert len(wide) == 1 and int(wide[0]["inner"]) == 0x10 and need == 0


def test_subdivision_keeps_the_surface():
    """Midpoint subdivision: 4x the faces per level, same total area, same bounding box."""
    base = h.packed_scene("room_horse")
    m = base.n_meshes - 1

    def area(tv):
        t = tv.reshape(-1, 3, 3)
        return 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1).sum()

    fine = base.with_subdivided_mesh(m, 2)
    b0, b1 = base.mesh_tri_begin[m], base.mesh_tri_begin[m + 1]
    f0, f1 = fine.mesh_tri_begin[m], fine.mesh_tri_begin[m + 1]
    assert f1 - f0 == 16 * (b1 - b0) and fine.mesh_tri_begin[-1] == fine.n_tris
    assert abs(area(fine.tri_verts[f0:f1]) - area(base.tri_verts[b0:b1])) < 1e-12
    assert np.array_equal(fine.tri_verts[:f0], base.tri_verts[:b0])
    v = fine.tri_verts[f0:f1].reshape(-1, 3)
    assert np.all(v.min(0) >= base.mesh_aabb[m][:3]) and np.all(v.max(0) <= base.mesh_aabb[m][3:])


def test_native_obj_reader_equals_the_python_statement(tmp_path):
    """rt_obj_open / rt_obj_read (C++ host code in the library) against the plain-Python statement of
    upstream's loader semantics (mesh.py:211-251), quirks included: records before the first `o`
    land in the first object, face indices are rebased by the vertices of EARLIER objects only,
    corners a/b/c and a//c, polygons fanned, short faces dropped, CRLF and indentation stripped,
    `vn` / `vt` / comments ignored; a file without `o` gives no mesh."""
    text = ("# comment\r\nv 9 9 9\nf 1 2 3\n\n  o first thing  \nv 0 0 0\nv 1 0 0\n\tv 1 1 0\nvn 0 0 1\nvt 0.5 0.5\n"
            "f 1/1/1 2/2/1 3/3/1 4/4/1 2\nf 1 2\no second\nv 0 0 1e0\nv 1.5e-3 -0 1\nv -0.25 1 1\ns off\nf 5//1 6//1 7//1\n"
            "usemtl x\nf 7 6 5 5\no empty\n")
    p = tmp_path / "quirks.obj"
    p.write_text(text)
    native = mesh.Mesh._parse_obj_native(str(p))
    plain = mesh.Mesh._parse_obj_python(str(p))
    assert native is not None and [g[0] for g in native] == [g[0] for g in plain] == ["first thing", "second", "empty"]
    for (_, va, ta), (_, vb, tb) in zip(native, plain):
        assert np.array_equal(va, np.asarray(vb, np.float64).reshape(-1, 3))
        assert np.array_equal(ta, np.asarray(tb, np.int32).reshape(-1, 3))
    assert native[0][2].tolist() == [[0, 1, 2], [0, 1, 2], [0, 2, 3], [0, 3, 1]]   # the stray face, then the fanned pentagon
    assert native[1][2].tolist() == [[0, 1, 2], [2, 1, 0], [2, 0, 0]] and len(native[2][1]) == 0
    ms = mesh.Mesh.load_from_obj_file(str(p))
    assert [m.name for m in ms] == ["first thing", "second", "empty"] and ms[2].bounding_box is None
    q = tmp_path / "no_objects.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    assert mesh.Mesh.load_from_obj_file(str(q)) == [] and mesh.Mesh._parse_obj_native(str(q)) == []
    with pytest.raises(ValueError):
        mesh.Mesh._parse_obj_native(str(tmp_path / "missing.obj"))


def test_app_helpers(tmp_path):
    """Caption / time formatting of the preview loop (main.py:13-15, 225-228) and the PNG writer's
    orientation: (W, H, 3) in, image of width W and height H out."""
    from PIL import Image

    from raytracing_amd import app

    assert app.convert_seconds(0) == "00:00:00" and app.convert_seconds(3725.9) == "01:02:05"
    assert app.convert_seconds(25 * 3600 + 61) == "01:01:01"   # hours wrap at 24 as upstream's do

    class R:
        progress_percent = 12.345678
    assert app.caption(R(), 5, 40) == "rendering: 12.3457%, estimated 00:00:05 / 00:00:40"
    img = np.zeros((7, 4, 3))
    img[6, 0] = (1.0, 0.5, 2.0)      # x = 6, y = 0; values above 1 clip
    p = tmp_path / "o.png"
    app.save_image(img, str(p))
    back = np.asarray(Image.open(p))
    assert back.shape == (4, 7, 3) and tuple(back[0, 6]) == (255, 127, 255) and back.sum() == 255 + 127 + 255


def test_threaded_hierarchy_build_equals_single_threaded(monkeypatch):
    """Above 65,536 triangles rt_build_bvh splits the top of the tree on the calling thread (its
    passes over large ranges shared among threads) and builds the subtrees below on workers; the
    result is the single-threaded hierarchy node for node."""
    from raytracing_amd.backend.bvh import build_flat_bvh, check_flat_bvh

    base = h.packed_scene("room_horse")
    tv = base.with_subdivided_mesh(base.n_meshes - 1, 5).tri_verts      # 295 k triangles
    out = {}
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("RAYTRACING_AMD_BUILD_THREADS", threads)
        out[threads] = build_flat_bvh(tv)
    for threads in ("3", "8"):
        assert np.array_equal(out["1"][0], out[threads][0]) and np.array_equal(out["1"][1], out[threads][1])
    check_flat_bvh(out["8"][0], out["8"][1], tv)


def test_scene_fingerprint_sees_every_setter_sequence():
    """The change detector must not depend on array identities: the mesh setters free one array and
    allocate the next, and CPython hands the same address out again, so id()-based keys come out
    unchanged after a few setter calls.  Any number of setter calls between two looks must change
    the fingerprint; no call must leave it alone."""
    from raytracing_amd import raytracing
    from raytracing_amd.core import Material, Vec3
    from raytracing_amd.mesh import Mesh, Sphere

    sc = raytracing.Scene()
    m = Mesh(Material(Vec3(0.8), 0, 0, 0, 1.0), Vec3(0, 0, 3), [(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)], [(0, 1, 2), (0, 2, 3)])
    sc.add_object(m)
    sc.add_object(Sphere(Material(Vec3(1), 1.0, 0, 0, 1.0), Vec3(0, 5, 3), 1.0))
    seen = {sc.fingerprint()}
    assert sc.fingerprint() in seen                      # stable without edits
    for n in range(1, 9):
        for trial in range(25):
            for k in range(n):
                which = (trial + k) % 3
                if which == 0:
                    m.set_origin(Vec3(0.01 * (trial + k), 0, 3))
                elif which == 1:
                    m.set_rotation(Vec3(0, 1.0 + trial + k, 0))
                else:
                    m.set_scale(Vec3(1.0 + 0.01 * (trial + k + 1)))
            fp = sc.fingerprint()
            assert fp not in seen, (n, trial)
            seen.add(fp)
    v0 = m.version
    m.transformed_vertices = m.transformed_vertices.copy()   # direct assignment counts as an edit
    m.triangles = [(0, 2, 1), (0, 3, 2)]
    m.touch()
    assert m.version == v0 + 3 and sc.fingerprint() not in seen

    class ForeignMesh:   # a mesh-like object without a change counter (upstream's own class)
        def __init__(self):
            self.material = Material(Vec3(1), 0, 0, 0, 1.0)
            self.origin, self.rotation, self.scale = Vec3(0), Vec3(0), Vec3(1)
            self.transformed_vertices = [Vec3(0, 0, 0), Vec3(1, 0, 0), Vec3(0, 1, 0)]
            self.triangles = [(0, 1, 2)]
            self.bounding_box = (Vec3(0, 0, 0), Vec3(1, 1, 0))
    f = ForeignMesh()
    sc.add_object(f)
    a = sc.fingerprint()
    f.origin = Vec3(0, 0, 1)
    f.bounding_box = (Vec3(0, 0, 1), Vec3(1, 1, 1))
    assert sc.fingerprint() != a


def test_app_loop_ends_when_the_render_thread_dies(monkeypatch):
    """A library error on the background thread must not leave the poll loop spinning: the error is
    kept on the result, the loop ends and run() raises; a stop request ends it without an error."""
    import threading

    from raytracing_amd import app, raytracing
    from raytracing_amd.backend.lib import BackendError

    class FakeBackend:
        device = 0

        def __init__(self, status):
            self.status = status

        def render(self, *a, **k):
            raise BackendError(self.status, "injected")

        def synchronize(self, *a):
            pass

        def stats(self):
            return {}

        def request_stop(self):
            pass

    def fake_start(self, settings, *a):
        self.render_settings = settings
        self.render_result = raytracing.RenderResult(settings.WIDTH, settings.HEIGHT, settings.RENDER_PASSES)
        self.samples_per_launch = 1
        self._backend = FakeBackend(fake_start.status)
        self.start_executor(1, 1)
        return self.render_result

    monkeypatch.setattr(raytracing.Renderer, "start_render", fake_start)
    settings = raytracing.RenderSettings(raytracing.Scene(), 8, 8, 4, 2)
    done = {}

    def go():
        try:
            done["ret"] = app.run(settings, poll_hz=500.0)
        except RuntimeError as e:
            done["err"] = e

    for status, expect_error in ((3, True), (5, False)):   # RT_ERR_HIP, RT_ERR_STOPPED
        fake_start.status = status
        done.clear()
        t = threading.Thread(target=go, daemon=True)
        t.start()
        t.join(10)
        assert not t.is_alive(), "poll loop did not end"
        if expect_error:
            assert "err" in done and "injected" in str(done["err"])
        else:
            res, _ = done["ret"]
            assert not res.finished and not res.failed and res.error is None


def test_pow_restatement_equals_this_machines_pow():
    """The build reads the tables of pow() out of the machine's libm, compiles the restatement the exact code uses
    (csrc/rt_libm_pow.h) for the host and holds it against pow() on four million results; on this image (glibc 2.35)
    that must succeed.  Elsewhere the build falls back to x*x and sqrt and says so - then there is nothing to test."""
    from raytracing_amd.backend import build

    had = os.path.exists(build.GEN_HEADER)
    keep = open(build.GEN_HEADER).read() if had else None
    try:
        ok = build.generate_libm_tables(verbose=False)
        if not ok:
            pytest.skip("no verifiable pow tables on this machine")
        text = open(build.GEN_HEADER).read()
        assert "kLogTab[128][3]" in text and "kExpTab[256]" in text
    finally:
        if keep is not None:
            with open(build.GEN_HEADER, "w") as f:
                f.write(keep)
