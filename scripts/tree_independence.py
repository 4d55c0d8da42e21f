"""Do two different hierarchies over the same triangles give the same paths?  (developer tool)
Traces the same seeded paths under two leaf-size settings and lists the first segment at which a path differs,
with the oracle's answer for that ray.   python scripts/tree_independence.py [level] [batches]"""
import sys, os, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
lv = int(sys.argv[1]) if len(sys.argv) > 1 else 6
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
base = h.packed_scene("room_horse")
sc = dataclasses.replace(base.with_subdivided_mesh(base.n_meshes - 1, lv), library_bvh=True)
bes = []
for leaf in ("3", "2"):
    os.environ["RAYTRACING_AMD_BVH_LEAF"] = leaf
    be = HipBackend(0); be.set_scene(sc); bes.append(be)
os.environ.pop("RAYTRACING_AMD_BVH_LEAF")
W, H = 480, 270
xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")
n_diff = 0; n_rays = 0
for b in range(batches):
    xys = np.stack([xs.ravel(), ys.ravel(), np.full(W * H, b)], axis=1).astype(np.int32)
    A, B = (be.trace_paths(xys, W, H, 8, h.SEED_RENDER) for be in bes)
    n_rays += int(A["n_seg"].sum())
    bad = np.nonzero((A["n_seg"] != B["n_seg"]) | (A["obj"] != B["obj"]).any(axis=1) | (A["prim"] != B["prim"]).any(axis=1))[0]
    for i in bad:
        n_diff += 1
        k = int(np.argmax((A["obj"][i] != B["obj"][i]) | (A["prim"][i] != B["prim"][i])))
        o, d = A["org"][i, k].astype(np.float64), A["dir"][i, k].astype(np.float64)
        same_ray = np.array_equal(A["org"][i, k], B["org"][i, k]) and np.array_equal(A["dir"][i, k], B["dir"][i, k])
        origin = -1
        if k > 0:
            prev = h.oracle_intersect(sc, A["org"][i, k - 1:k].astype(np.float64), A["dir"][i, k - 1:k].astype(np.float64))
            origin = int(prev["gprim"][0])
        with h.origin_rule():
            want = h.oracle_intersect(sc, o[None], d[None], origin_prim=np.array([origin], np.int32))
        print(f"path {tuple(xys[i])} segment {k} (same ray: {same_ray}, origin {origin}): leaf3 obj {A['obj'][i, k]} prim {A['prim'][i, k]} t {A['t'][i, k]:.9g} how {A['how'][i, k]} | "
              f"leaf2 obj {B['obj'][i, k]} prim {B['prim'][i, k]} t {B['t'][i, k]:.9g} how {B['how'][i, k]} | oracle obj {want['obj'][0]} prim {want['prim'][0]} t {want['t'][0]:.9g}")
        print(f"   o {o.tolist()} d {d.tolist()}")
print(f"{n_rays} rays, {n_diff} paths differ")
