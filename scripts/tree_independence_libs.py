"""The same render under the library named by RAYTRACING_AMD_LIB; saves the image or compares it with a saved one
(developer tool: two builds whose hierarchy builders differ must give the same image).
    python scripts/tree_independence_libs.py save|compare file.npy [level] [spp]"""
import sys, os, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
mode, path = sys.argv[1], sys.argv[2]
lv = int(sys.argv[3]) if len(sys.argv) > 3 else 6
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 32
base = h.packed_scene("room_horse")
sc = dataclasses.replace(base.with_subdivided_mesh(base.n_meshes - 1, lv), library_bvh=True)
W, H = 1920, 1080
be = HipBackend(0); be.set_scene(sc); be.resize(W, H)
be.render(W, H, 0, spp, 8, 1); be.synchronize()
img = be.read().copy(); st = be.stats()
print("rays", st["rays"], "nodes/ray", st["bvh_node_tests"] / st["rays"])
if mode == "save":
    np.save(path, img)
else:
    ref = np.load(path)
    d = np.nonzero((ref != img).any(axis=2))
    print(len(d[0]), "pixels differ:", [(int(x), int(y), ref[x, y].tolist(), img[x, y].tolist()) for x, y in zip(d[0][:6], d[1][:6])])
