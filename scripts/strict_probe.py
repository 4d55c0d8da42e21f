"""Strict against loose depth acceptance on the lean pipeline and the one-kernel form (developer tool):
    python scripts/strict_probe.py [spp]"""
import sys, os, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as h
from raytracing_amd.backend.lib import HipBackend
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["room", "random_spheres_seed7", "horse6"]
be = HipBackend(0)
W, H = 1920, 1080
for name in which:
    if name.startswith("horse"):
        base = h.packed_scene("room_horse")
        be.set_scene(dataclasses.replace(base.with_subdivided_mesh(base.n_meshes - 1, int(name[5:] or 6)), library_bvh=True))
    else:
        be.set_scene(h.packed_scene(name))
    be.resize(W, H)
    be.render(W, H, 0, 4, 8, 1); be.synchronize()
    for strict in (True, False):
        for no_defer in (False, True):
            best = None
            for rep in range(2):
                be.clear(); be.reset_stats()
                be.render(W, H, 0, spp, 8, 1, strict_depth=strict, no_defer=no_defer); be.synchronize()
                st = be.stats()
                if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
            st = best
            cnt = be.render_counters()
            print(f"{name:22s} {spp}spp strict={int(strict)} one_kernel={int(no_defer)} {st['kernel_ms']:.2f} ms {st['rays']/st['kernel_ms']/1e6:.2f} Grays/s "
                  f"exact={st['exact_rays']/st['rays']:.2e} refined={st['refined_hits']/st['rays']:.2e} aside={cnt['paths_set_aside']} "
                  f"({cnt['paths_set_aside']/st['paths']:.2e} of paths) given_up={cnt['jobs_given_up']}", flush=True)
