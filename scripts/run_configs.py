"""Measures the BASELINE configurations other than the bench.py headline on one GPU (developer tool).  Usage: python scripts/run_configs.py [c2] [c4] [c5] [--spp-scale F]"""
import sys, os, time, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
import helpers as h
from raytracing_amd.backend.lib import HipBackend
from raytracing_amd.post_processing import PostProcessing

which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c2", "c4", "c5"]
scale = 1.0
for a in sys.argv[1:]:
    if a.startswith("--spp-scale="): scale = float(a.split("=")[1])
be = HipBackend(0)
stream = torch.cuda.current_stream().cuda_stream


def run(name, scene, W, H, spp, bounces, batch=None, tonemap=False):
    spp = max(1, int(spp * scale))
    be.set_scene(scene)
    acc = torch.zeros((W, H, 3), dtype=torch.float32, device="cuda:0")
    be.render(W, H, 0, 1, bounces, 1, accum_ptr=acc.data_ptr(), stream=stream); be.synchronize(stream)
    acc.zero_(); be.reset_stats(); torch.cuda.synchronize()
    pp = PostProcessing(exposure=2, brightness=0.02, contrast=1.1)
    t0 = time.perf_counter()
    step = batch or spp
    for s0 in range(0, spp, step):
        n = min(step, spp - s0)
        be.render(W, H, s0, n, bounces, 1, accum_ptr=acc.data_ptr(), stream=stream)
        if tonemap:
            img = pp.process((acc / float(s0 + n)).flip(1))
    be.synchronize(stream); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = be.stats()
    print(f"{name}: {W}x{H} {spp} spp {bounces} bounces: {dt*1e3:.1f} ms wall, kernel {st['kernel_ms']:.1f} ms, "
          f"{st['rays']/dt/1e9:.2f} Grays/s, {st['paths']/dt/1e9:.3f} Gsamples/s, rays/path {st['rays']/max(st['paths'],1):.2f}, "
          f"nodes/ray {st['bvh_node_tests']/max(st['rays'],1):.1f}, tris/ray {st['tri_tests']/max(st['rays'],1):.1f}", flush=True)


if "c2" in which:
    run("C2 random_spheres", h.packed_scene("random_spheres_seed7"), 1920, 1080, 64, 8)
base = h.packed_scene("room_horse")
big = base.with_subdivided_mesh(base.n_meshes - 1, 6)
if "c4" in which:
    run("C4 room + 1.18M-triangle mesh", big, 1920, 1080, 128, 8)
if "c5" in which:
    run("C5 room + mesh 4K progressive + tonemap (1 GPU share: 128 of 1024 spp)", big, 3840, 2160, 128, 8, batch=16, tonemap=True)
