"""Which sphere hits of the room still need double precision under the strict acceptance (developer tool)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend
be = HipBackend(0)
sc = h.packed_scene("room")
be.set_scene(sc)
W, H = 192, 108
xys = np.array([(x, y, s) for x in range(W) for y in range(H) for s in range(4)], np.int32)
tr = be.trace_paths(xys, W, H, 8, 1)
n = tr["n_seg"]
mask = np.arange(tr["obj"].shape[1])[None, :] < n[:, None]
obj, how, t = tr["obj"][mask], tr["how"][mask], tr["t"][mask]
org, drs = tr["org"][mask].astype(np.float64), tr["dir"][mask].astype(np.float64)
seg = np.broadcast_to(np.arange(tr["obj"].shape[1])[None, :], tr["obj"].shape)[mask]
print("segments", len(obj), "sphere hits", int((obj == 6).sum()), "how histogram all", np.bincount(how[how >= 0]))
sp = obj == 6
print("sphere hits by how", np.bincount(how[sp]))
bad = sp & (how == 1)
a = (drs * drs).sum(1); b = (org * drs).sum(1); tc = -b / a
l2 = ((org + tc[:, None] * drs) ** 2).sum(1); h2 = 0.25 - l2
prev_obj = np.where(seg > 0, np.roll(obj, 1), -1)
for name, m in (("refined", bad), ("accepted", sp & (how == 0))):
    if m.sum() == 0: continue
    print(name, int(m.sum()), "seg0 share", float((seg[m] == 0).mean()), "from sphere share", float((prev_obj[m] == 6).mean()),
          "median s", float(np.median(np.sqrt(np.maximum(h2[m], 0) / a[m]))), "median t", float(np.median(t[m])), "median |o|", float(np.median(np.linalg.norm(org[m], axis=1))),
          "median |d|", float(np.median(np.sqrt(a[m]))))
    print("   s quantiles", np.quantile(np.sqrt(np.maximum(h2[m], 0) / a[m]), [0.01, 0.1, 0.5, 0.9, 0.99]))
    print("   t quantiles", np.quantile(t[m], [0.01, 0.1, 0.5, 0.9, 0.99]))
