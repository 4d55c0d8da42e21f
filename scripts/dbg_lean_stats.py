import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, helpers as h
from raytracing_amd.backend.lib import HipBackend
be = HipBackend(0); sc = h.packed_scene("room_horse"); be.set_scene(sc)
W, H, spp = 96, 54, 8
for nd in (True, False):
    be.resize(W, H); be.reset_stats(); be.render(W, H, 0, spp, 8, h.SEED_RENDER, no_defer=nd); be.synchronize()
    print("no_defer", nd, be.stats(), flush=True)
xys = np.array([(x, y, s) for x in range(W) for y in range(H) for s in range(spp)], np.int32)
tr = be.trace_paths(xys, W, H, 8, h.SEED_RENDER)
n = tr["n_seg"]; m = np.arange(tr["how"].shape[1])[None, :] < n[:, None]
print("trace: segments", int(n.sum()), "how==2", int(((tr["how"] == 2) & m).sum()), "how==1", int(((tr["how"] == 1) & m).sum()))
