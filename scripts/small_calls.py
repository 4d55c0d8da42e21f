"""Back-to-back render calls of few samples per pixel (what one GPU of several does per step; developer tool):
    python scripts/small_calls.py [spp_per_call] [calls]"""
import sys, os, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as h
from raytracing_amd.backend.lib import HipBackend
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
be = HipBackend(0)
be.set_scene(h.packed_scene("room"))
W, H = 1920, 1080
be.resize(W, H)
be.render(W, H, 0, 4, 8, 1); be.synchronize()
be.reset_stats()
t0 = time.perf_counter()
for k in range(calls):
    be.render(W, H, k * spp, spp, 8, 1)
be.synchronize()
dt = (time.perf_counter() - t0) * 1e3
st = be.stats()
print(f"{calls} calls of {spp} spp: wall {dt:.2f} ms = {dt / calls:.2f} per call; by HIP events {st['kernel_ms']:.2f} ms = {st['kernel_ms'] / calls:.2f} per call "
      f"(min {st['kernel_ms_min']:.2f}, max {st['kernel_ms_max']:.2f}); {st['rays'] / dt / 1e6:.2f} G rays/s")
