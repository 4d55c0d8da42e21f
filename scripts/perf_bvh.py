"""Times the render kernel on room + subdivided horse meshes (developer tool)."""
import sys, os, time, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
be = HipBackend(0)
W, H = 1920, 1080
base = h.packed_scene("room_horse")
cases = [(0, False, 16), (0, True, 16), (2, True, 8), (4, True, 8), (6, True, 8)]
if len(sys.argv) > 1:
    cases = [(int(a), True, int(os.environ.get('SPP', 8))) for a in sys.argv[1:]]
for lv, bvh, spp in cases:
    sc = base.with_subdivided_mesh(base.n_meshes - 1, lv) if lv else base
    sc = dataclasses.replace(sc, library_bvh=bvh)
    t0 = time.time(); be.set_scene(sc); t_set = time.time() - t0
    be.resize(W, H)
    be.render(W, H, 0, 1, 8, 1); be.synchronize()
    best = None
    for rep in range(2):
        be.clear(); be.reset_stats()
        be.render(W, H, 0, spp, 8, 1); be.synchronize()
        st = be.stats()
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    st = best
    print(f"tris {sc.n_tris:8d} bvh={int(bvh)} set_scene {t_set:.2f}s {spp}spp kernel {st['kernel_ms']:.2f} ms  {st['rays']/st['kernel_ms']/1e6:.3f} Grays/s  "
          f"stack need {be.scene_layout()['walk_stack_need']} fallbacks {be.walk_fallbacks()} nodes/ray {st['bvh_node_tests']/st['rays']:.1f} tris/ray {st['tri_tests']/st['rays']:.1f} exact={st['exact_rays']/st['rays']:.2e} refine={st['refined_hits']/st['rays']:.3f}", flush=True)
