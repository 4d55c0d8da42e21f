"""What a render call costs whatever its size (developer tool): per-call time over image sizes and sample counts, room scene.
    python scripts/fixed_cost.py"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as h
from raytracing_amd.backend.lib import HipBackend
be = HipBackend(0)
be.set_scene(h.packed_scene("room"))
for (W, H) in ((480, 270), (960, 540), (1920, 1080), (3840, 2160)):
    be.resize(W, H)
    be.render(W, H, 0, 4, 8, 1); be.synchronize()
    for spp in (1, 2, 4, 8, 16, 32):
        be.reset_stats()
        calls = 8
        for k in range(calls):
            be.render(W, H, k * spp, spp, 8, 1)
        be.synchronize()
        st = be.stats()
        print(f"{W}x{H} {spp:3d} spp: {st['kernel_ms'] / calls:7.3f} ms per call (min {st['kernel_ms_min']:.3f}) = {st['kernel_ms_min'] * 1e6 / (W * H * spp):7.2f} ns per path", flush=True)
