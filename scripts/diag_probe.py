"""The 1080p pixels whose camera rays run exactly along the room's edges (x - y = 420, x + y = 1499): per-path
radiance of the kernel against the oracle (developer tool)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
be = HipBackend(0)
sc = h.packed_scene("room"); be.set_scene(sc)
W, H, b, spp = 1920, 1080, 8, 4
px = [(x, x - 420) for x in range(W) if 0 <= x - 420 < H] + [(x, 1499 - x) for x in range(W) if 0 <= 1499 - x < H]
xys = np.array([(x, y, s) for x, y in px for s in range(spp)], np.int32)
tr = be.trace_paths(xys, W, H, b, h.SEED_RENDER)
bad = []
with h.origin_rule():
    for i, (x, y, s) in enumerate(xys):
        want, _, _, seg = h.oracle_trace_pixel(sc, W, H, int(x), int(y), int(s), b, with_segments=True)
        if not np.allclose(tr["rgb"][i], want, rtol=2e-3, atol=2e-4):
            n = int(tr["n_seg"][i]); k = 0
            while k < min(n, len(seg["obj"])) and tr["obj"][i, k] == seg["obj"][k] and tr["prim"][i, k] == seg["prim"][k]: k += 1
            bad.append((i, k))
print(f"{len(px)} pixels, {len(xys)} paths, {len(bad)} differ ({len(bad)/len(xys):.3e})")
from collections import Counter
print("first differing segment histogram", Counter(k for _, k in bad))
for i, k in bad[:6]:
    x, y, s = xys[i]
    want, _, _, seg = h.oracle_trace_pixel(sc, W, H, int(x), int(y), int(s), b, with_segments=True)
    print(f"path {i} px ({x},{y}) s {s}: differs at segment {k}; gpu obj/prim/how {tr['obj'][i,:k+1]}/{tr['prim'][i,:k+1]}/{tr['how'][i,:k+1]} oracle {seg['obj'][:k+1]}/{seg['prim'][:k+1]}")
    if k > 0:
        print("    gpu org", tr["org"][i, k], "oracle org", seg["org"][k], "gpu dir", tr["dir"][i, k], "oracle dir", seg["dir"][k])
