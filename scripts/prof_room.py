"""One render of the room at 1080p (profiling target; developer tool)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as h
from raytracing_amd.backend.lib import HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "room"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
be = HipBackend(0)
be.set_scene(h.packed_scene(name))
W, H = 1920, 1080
be.resize(W, H)
be.render(W, H, 0, spp, 8, 1); be.synchronize()
st = be.stats()
print(f"{name} kernel {st['kernel_ms']:.2f} ms rays {st['rays']} paths {st['paths']}")
print("set aside:", be.render_counters())
