"""The paths through the room cube's edges at 1920x1080 that part company with the oracle's: for each, the first
segment that names another hit, the distance between the two origins and directions there, how the hit before it was
produced, and what the oracle answers for the kernel's own ray (developer tool)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
sc = h.packed_scene("room")
be = HipBackend(0); be.set_scene(sc)
W, H, b, spp = 1920, 1080, 8, 4
px = [(x, x - 420) for x in range(W) if 0 <= x - 420 < H] + [(x, 1499 - x) for x in range(W) if 0 <= 1499 - x < H]
xys = np.array([(x, y, s) for x, y in px for s in range(spp)], np.int32)
tr = be.trace_paths(xys, W, H, b, h.SEED_RENDER)
n = 0
with h.origin_rule():
    for i, (x, y, s) in enumerate(xys):
        want, _, _, seg = h.oracle_trace_pixel(sc, W, H, int(x), int(y), int(s), b, with_segments=True)
        if np.allclose(tr["rgb"][i], want, rtol=2e-3, atol=2e-4):
            continue
        n += 1
        ns = min(int(tr["n_seg"][i]), len(seg["obj"]))
        k = next((k for k in range(ns) if tr["obj"][i, k] != seg["obj"][k] or tr["prim"][i, k] != seg["prim"][k]), None)
        if k is None:
            print(f"path {i} ({x},{y},{s}): same hits over {ns} segments, n_seg {tr['n_seg'][i]} vs {len(seg['obj'])}; rgb {tr['rgb'][i]} vs {want}")
            continue
        do = np.abs(tr["org"][i, k] - np.asarray(seg["org"][k])).max()
        dd = np.abs(tr["dir"][i, k].astype(np.float64) - np.asarray(seg["dir"][k])).max()
        prev = h.oracle_intersect(sc, tr["org"][i, k - 1:k], tr["dir"][i, k - 1:k].astype(np.float64)) if k > 0 else None
        origin = np.array([int(prev["gprim"][0])], np.int32) if k > 0 else None
        own = h.oracle_intersect(sc, tr["org"][i, k:k + 1], tr["dir"][i, k:k + 1].astype(np.float64), origin_prim=origin)
        print(f"path {i} ({x},{y},{s}) seg {k}: gpu ({tr['obj'][i, k]},{tr['prim'][i, k]}) t {tr['t'][i, k]:.7g} how {tr['how'][i, k]} | oracle ({seg['obj'][k]},{seg['prim'][k]}) t {seg['t'][k]:.7g} "
              f"| origins apart {do:.2e}, directions {dd:.2e}, hit before: how {tr['how'][i, k - 1] if k else -1} residue {np.abs(tr['org_res'][i, k]).max():.1e} "
              f"| oracle on the kernel's ray ({own['obj'][0]},{own['prim'][0]}) t {own['t'][0]:.7g}")
print(n, "of", len(xys), "paths differ")
