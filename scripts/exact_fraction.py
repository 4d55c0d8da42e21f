import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import helpers as h
from raytracing_amd.backend.lib import HipBackend
for env in (None, "RAYTRACING_AMD_NO_BOXES"):
    if env: os.environ[env] = "1"
    be = HipBackend(0); be.set_scene(h.packed_scene("room"))
    for (W, H, spp) in ((48, 27, 4), (96, 54, 8), (100, 57, 8), (1920, 1080, 4)):
        be.resize(W, H); be.reset_stats(); be.render(W, H, 0, spp, 8, h.SEED_RENDER); be.synchronize()
        st = be.stats()
        print(env, W, H, spp, "rays", st["rays"], "exact", st["exact_rays"], f"{st['exact_rays']/st['rays']:.5f}", "refined", st["refined_hits"], flush=True)
