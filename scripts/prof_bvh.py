"""One short render of the room + 1.18 M-triangle mesh scene (profiling target; developer tool)."""
import sys, os, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as h
from raytracing_amd.backend.lib import HipBackend
lv = int(sys.argv[1]) if len(sys.argv) > 1 else 6
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
be = HipBackend(0)
base = h.packed_scene("room_horse")
sc = base.with_subdivided_mesh(base.n_meshes - 1, lv) if lv else base
be.set_scene(dataclasses.replace(sc, library_bvh=True))
W, H = 1920, 1080
be.resize(W, H)
be.render(W, H, 0, spp, 8, 1); be.synchronize()
st = be.stats()
print(f"tris {sc.n_tris} kernel {st['kernel_ms']:.2f} ms rays {st['rays']} nodes/ray {st['bvh_node_tests']/st['rays']:.1f} tris/ray {st['tri_tests']/st['rays']:.1f}")
print("set aside:", be.render_counters())
