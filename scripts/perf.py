"""Times the render kernel on the two 1080p workloads (developer tool)."""
import sys, os, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
be = HipBackend(0)
W, H = 1920, 1080
for name, spp in (("room", 64), ("random_spheres_seed7", 32)):
    be.set_scene(h.packed_scene(name)); be.resize(W, H)
    be.render(W, H, 0, 4, 8, 1); be.synchronize()
    best = None
    for rep in range(3):
        be.clear(); be.reset_stats()
        be.render(W, H, 0, spp, 8, 1); be.synchronize()
        st = be.stats()
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    st = best
    print(f"{os.environ.get('RAYTRACING_AMD_LIB','default'):>40s} {name:22s} {spp}spp kernel {st['kernel_ms']:.2f} ms  {st['rays']/st['kernel_ms']/1e6:.2f} Grays/s  exact={st['exact_rays']/st['rays']:.2e} refine={st['refined_hits']/st['rays']:.3f}", flush=True)
