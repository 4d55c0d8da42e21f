"""Why a fuzz scene's small render differs from the oracle's: for every differing pixel and sample, the first path
segment that names another hit (or, when every hit agrees, the radiance), the distance between the two rays there,
and what the oracle answers for the kernel's own ray (developer tool):  python scripts/fuzz_render_diag.py seed..."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "scripts"))
import numpy as np
import helpers as h
import fuzz_scenes as fz
from raytracing_amd.backend.lib import HipBackend

pool = fz.materials_pool()
be = HipBackend(0)
W, H, spp, b = 48, 27, 2, 5
for seed in [int(a) for a in sys.argv[1:]]:
    sc, _ = fz.make_scene(seed, pool)
    with h.linear_tables():
        be.set_scene(sc)
    be.resize(W, H); be.render(W, H, 0, spp, b, h.SEED_RENDER); be.synchronize()
    img = be.read().astype(np.float64)
    with h.origin_rule():
        ref, _ = h.oracle_render(sc, W, H, spp, b)
    px = np.argwhere(np.abs(img - ref).max(axis=2) > 1e-3 * (1.0 + ref.max(axis=2)))
    xys = np.array([(x, y, s) for x, y in px for s in range(spp)], np.int32)
    tr = be.trace_paths(xys, W, H, b, h.SEED_RENDER)
    kinds = {}
    shown = 0
    with h.origin_rule():
        for i, (x, y, s) in enumerate(xys):
            want, _, _, seg = h.oracle_trace_pixel(sc, W, H, int(x), int(y), int(s), b, with_segments=True)
            if np.allclose(tr["rgb"][i], want, rtol=2e-3, atol=2e-4):
                continue
            ns = min(int(tr["n_seg"][i]), len(seg["obj"]))
            k = next((k for k in range(ns) if tr["obj"][i, k] != seg["obj"][k] or tr["prim"][i, k] != seg["prim"][k]), None)
            if k is None:
                kind = "same hits, other radiance" if int(tr["n_seg"][i]) == len(seg["obj"]) else "same hits, other length"
                last = ns - 1
                detail = (f"rgb {tr['rgb'][i]} vs {want}; n_seg {tr['n_seg'][i]} vs {len(seg['obj'])}; objs {list(seg['obj'])}; last dir gpu {tr['dir'][i, last]} oracle {np.asarray(seg['dir'][last])}; "
                          f"materials {[sc.materials[o].tolist() if o >= 0 else None for o in seg['obj']]}")
            else:
                do = np.abs(tr["org"][i, k] - np.asarray(seg["org"][k])).max()
                dd = np.abs(tr["dir"][i, k].astype(np.float64) - np.asarray(seg["dir"][k])).max()
                origin = None
                if k > 0:   # the primitive the kernel's ray left from (the same one the oracle's path left from: hits agree up to here)
                    origin = fz.gprim_of(sc, tr["obj"][i, k - 1:k], tr["prim"][i, k - 1:k])
                own = h.oracle_intersect(sc, tr["org"][i, k:k + 1], tr["dir"][i, k:k + 1].astype(np.float64), origin_prim=origin)
                agrees = own["obj"][0] == tr["obj"][i, k] and own["prim"][0] == tr["prim"][i, k]
                kind = ("another ray (float32 shading), oracle agrees on the kernel's ray" if agrees else "ORACLE DISAGREES ON THE KERNEL'S OWN RAY") \
                    + (" [rays apart > 1e-5]" if max(do, dd) > 1e-5 else "")
                detail = (f"seg {k}: gpu ({tr['obj'][i, k]},{tr['prim'][i, k]}) t {tr['t'][i, k]:.9g} how {tr['how'][i, k]} | oracle ({seg['obj'][k]},{seg['prim'][k]}) t {seg['t'][k]:.9g} | "
                          f"origins apart {do:.2e}, directions {dd:.2e} | oracle on the kernel's ray ({own['obj'][0]},{own['prim'][0]}) t {own['t'][0]:.9g}")
            kinds[kind] = kinds.get(kind, 0) + 1
            if shown < 6 or "ORACLE" in kind or "radiance" in kind and shown < 12:
                shown += 1
                print(f"  seed {seed} path ({x},{y},{s}): {kind}\n      {detail}")
    print(f"seed {seed}: {len(px)} pixels differ; paths: {kinds}", flush=True)
