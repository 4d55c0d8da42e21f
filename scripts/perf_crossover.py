"""Linear tables against the hierarchy for small triangle counts (developer tool): room + the first
k triangles of the horse mesh, 1080p, 16 spp."""
import sys, os, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
be = HipBackend(0)
base = h.packed_scene("room_horse")
m = base.n_meshes - 1
b0, b1 = int(base.mesh_tri_begin[m]), int(base.mesh_tri_begin[m + 1])
W, H, spp = 1920, 1080, 16
be.resize(W, H)
for k in (16, 32, 64, 128, 288):
    tv = np.ascontiguousarray(np.concatenate([base.tri_verts[:b0], base.tri_verts[b0:b0 + k]]))
    begin = base.mesh_tri_begin.copy(); begin[m + 1] = b0 + k
    aabb = base.mesh_aabb.copy(); pts = tv[b0:].reshape(-1, 3); aabb[m] = np.concatenate([pts.min(0), pts.max(0)])
    sc = dataclasses.replace(base, tri_verts=tv, mesh_tri_begin=begin, mesh_aabb=aabb)
    out = []
    for lib_bvh in (False, True):
        be.set_scene(dataclasses.replace(sc, library_bvh=lib_bvh))
        be.render(W, H, 0, 2, 8, 1); be.synchronize()
        best = 1e9
        for _ in range(3):
            be.clear(); be.reset_stats(); be.render(W, H, 0, spp, 8, 1); be.synchronize()
            st = be.stats(); best = min(best, st["kernel_ms"])
        out.append(st["rays"] / best / 1e6)
    print(f"{22 + k:4d} triangles: linear {out[0]:.2f}  hierarchy {out[1]:.2f} G rays/s", flush=True)
