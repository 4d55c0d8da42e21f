"""When the waves of the lean kernel run dry and end (developer tool; needs a library built with -DRT_WAVE_END):
    RAYTRACING_AMD_LIB=variants/ends.so python scripts/wave_ends.py [spp ...]"""
import ctypes as C, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as h
from raytracing_amd.backend.lib import HipBackend
be = HipBackend(0)
be.set_scene(h.packed_scene(os.environ.get("SCENE", "room")))
W, H = 1920, 1080
be.resize(W, H)
be.render(W, H, 0, 4, 8, 1); be.synchronize()
be.lib.rt_debug_wave_log.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
for spp in [int(a) for a in sys.argv[1:]] or [8, 32, 256]:
    be.reset_stats()
    be.render(W, H, 0, spp, 8, 1); be.synchronize()
    st = be.stats()
    raw = (C.c_uint64 * (6 * 8192))()
    be.lib.rt_debug_wave_log(be.handle, raw, 6 * 8192)
    v = np.array(raw, dtype=np.uint64).reshape(8192, 6).astype(np.float64)
    res = v[:1280 * 4]                     # the blocks resident from the start
    t0 = res[:, 0].min()
    dry, end = (res[:, 1] - t0) / 100.0, (res[:, 2] - t0) / 100.0
    worked = res[:, 3] > 0
    d0 = dry[worked].min()
    print(f"{spp:4d} spp: call {st['kernel_ms']:.3f} ms; resident waves: ran dry first {d0:.0f} mean {dry[worked].mean():.0f} last {dry[worked].max():.0f} us; "
          f"ended mean {end.mean():.0f} last {end.max():.0f} us; idle wave time behind the mean end {100 * (end.max() - end.mean()) / end.max():.1f} % of the kernel")
    print(f"      after running dry: {res[:, 3].mean():.1f} trips (most {res[:, 3].max():.0f}), {res[:, 4].mean():.1f} lanes busy then, {(end - dry)[worked].mean():.0f} us to the end (most {(end - dry)[worked].max():.0f}): "
          f"{((end - dry)[worked] / np.maximum(res[worked, 3], 1)).mean():.1f} us per trip")
    print("      ended, by 100 us after the first ran dry:", np.histogram(end - d0, bins=[-1e9] + [100 * k for k in range(1, 13)] + [1e9])[0].tolist(), flush=True)
