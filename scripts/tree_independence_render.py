"""Render-level companion of tree_independence.py: the same call under two leaf-size settings, default pipeline and
one-kernel form (developer tool)."""
import sys, os, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
lv = int(sys.argv[1]) if len(sys.argv) > 1 else 6
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
base = h.packed_scene("room_horse")
sc = dataclasses.replace(base.with_subdivided_mesh(base.n_meshes - 1, lv), library_bvh=True)
W, H = 1920, 1080
res = {}
for leaf in ("3", "2"):
    os.environ["RAYTRACING_AMD_BVH_LEAF"] = leaf
    be = HipBackend(0); be.set_scene(sc); be.resize(W, H)
    for nd in (False, True):
        be.clear(); be.reset_stats()
        be.render(W, H, 0, spp, 8, 1, no_defer=nd); be.synchronize()
        res[(leaf, nd)] = (be.read().copy(), be.stats()["rays"])
        print(f"leaf {leaf} no_defer {nd}: rays {res[(leaf, nd)][1]}", flush=True)
    be.close()
for nd in (False, True):
    a, b = res[("3", nd)], res[("2", nd)]
    diff = np.nonzero((a[0] != b[0]).any(axis=2))
    print(f"no_defer {nd}: rays {a[1]} vs {b[1]}, {len(diff[0])} pixels differ: {list(zip(diff[0][:8].tolist(), diff[1][:8].tolist()))}")
a, b = res[("3", False)], res[("3", True)]
print("lean vs one kernel, leaf 3:", int((a[0] != b[0]).any(axis=2).sum()), "pixels differ")
