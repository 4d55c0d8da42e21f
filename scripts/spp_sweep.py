"""Time of one rt_render call against its samples per pixel (room 1080p; developer tool): the per-call floor."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as h
from raytracing_amd.backend.lib import HipBackend
be = HipBackend(0)
be.set_scene(h.packed_scene("room"))
W, H = 1920, 1080
be.resize(W, H)
be.render(W, H, 0, 4, 8, 1); be.synchronize()
for spp in [int(a) for a in sys.argv[1:]] or [8, 16, 24, 32, 48, 64, 128, 256]:
    best = 1e9
    for rep in range(3):
        be.reset_stats()
        be.render(W, H, 0, spp, 8, 1); be.synchronize()
        best = min(best, be.stats()["kernel_ms"])
    print(f"spp {spp:4d}: {best:7.3f} ms per call = {best / spp * 32:6.3f} ms per 32 spp", flush=True)
