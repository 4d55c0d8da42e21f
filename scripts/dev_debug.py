import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
be = HipBackend(0)
for name, f in (("room", "rays_room.npz"), ("random_spheres_seed7", "rays_random_spheres.npz")):
    sc = h.packed_scene(name); be.set_scene(sc)
    z, _ = h.load_npz(f)
    o32 = z["org"].astype(np.float32).astype(np.float64); d32 = z["dir"].astype(np.float32).astype(np.float64)
    ref = h.oracle_intersect(sc, o32, d32)
    seg = z["seg"]; prev_g = np.full(len(seg), -1, np.int32)
    refd = h.oracle_intersect(sc, z["org"], z["dir"])
    prev_g[1:] = np.where(seg[1:] > 0, refd["gprim"][:-1], -1)
    for tag, op in (("no-origin", None), ("origin", prev_g)):
        r = be.intersect(o32, d32, origin_prim=op)
        bad = np.nonzero((r["obj"] != ref["obj"]) | (r["prim"] != ref["prim"]))[0]
        if op is not None:
            selfhit = (ref["gprim"] == prev_g) & (ref["t"] < 1e-3)
            bad = np.array([b for b in bad if not selfhit[b]])
        print(name, tag, "bad", len(bad))
        for k in bad[:12]:
            print("  seg", seg[k], "origin", prev_g[k], "ref", ref["obj"][k], ref["prim"][k], ref["gprim"][k], "%.9g" % ref["t"][k], "nh", ref["n_hits"][k],
                  "| gpu", r["obj"][k], r["prim"][k], r["gprim"][k], "%.9g" % r["t"][k], "how", r["how"][k], "o", o32[k], "d", d32[k])
