"""Every segment of every sample path of one pixel, each checked against the oracle (developer tool; the library is the
one named by RAYTRACING_AMD_LIB).   python scripts/trace_pixel.py x y [spp] [level]"""
import sys, os, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
x, y = int(sys.argv[1]), int(sys.argv[2])
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
lv = int(sys.argv[4]) if len(sys.argv) > 4 else 6
base = h.packed_scene("room_horse")
sc = dataclasses.replace(base.with_subdivided_mesh(base.n_meshes - 1, lv), library_bvh=True)
W, H = 1920, 1080
be = HipBackend(0); be.set_scene(sc)
xys = np.array([[x, y, s] for s in range(spp)], np.int32)
T = be.trace_paths(xys, W, H, 8, 1)
for s in range(spp):
    origin = -1
    for k in range(int(T["n_seg"][s])):
        o, d = T["org"][s, k].astype(np.float64), T["dir"][s, k].astype(np.float64)
        with h.origin_rule():
            want = h.oracle_intersect(sc, o[None], d[None], origin_prim=np.array([origin], np.int32))
        ok = want["obj"][0] == T["obj"][s, k] and want["prim"][0] == T["prim"][s, k]
        print(f"s{s} k{k} obj {T['obj'][s, k]} prim {T['prim'][s, k]} t {T['t'][s, k]:.9g} how {T['how'][s, k]} | oracle obj {want['obj'][0]} prim {want['prim'][0]} t {want['t'][0]:.9g} gprim {want['gprim'][0]} {'' if ok else '  <-- DIFFERS'}")
        if not ok:
            print("   o", o.tolist(), "d", d.tolist(), "origin", origin)
        origin = int(want["gprim"][0])
print("rgb", T["rgb"].sum(axis=0))
