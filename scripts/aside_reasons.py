"""Why the lean kernel sets paths aside (developer tool; needs a library built with -DRT_ASIDE_REASONS:
    RAYTRACING_AMD_LIB=variants/_rt_reasons.so python scripts/aside_reasons.py [room|horse6] [spp])."""
import ctypes as C, os, sys, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as h
from raytracing_amd.backend.lib import HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "room"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
be = HipBackend(0)
if name.startswith("horse"):
    base = h.packed_scene("room_horse")
    be.set_scene(dataclasses.replace(base.with_subdivided_mesh(base.n_meshes - 1, int(name[5:] or 6)), library_bvh=True))
else:
    be.set_scene(h.packed_scene(name))
W, H = 1920, 1080
be.resize(W, H); be.reset_stats()
be.render(W, H, 0, spp, 8, 1); be.synchronize()
st = be.stats()
out = (C.c_uint64 * 8)()
be.lib.rt_debug_section_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
be.lib.rt_debug_section_cycles(be.handle, out, 8)
names = ["ray outside the filter's range", "box edge (which face)", "pair diagonal (which triangle)", "winner uncertain", "sphere depth imprecise",
         "triangle depth imprecise + contender", "contender only"]
tot = sum(out[:7])
print(f"{name} {spp} spp: rays {st['rays']}, paths {st['paths']}, set aside {tot} ({100.0 * tot / st['paths']:.3f} % of paths)")
for n, v in zip(names, out):
    print(f"  {n:40s} {v:10d}  {100.0 * v / max(tot, 1):5.1f} %")
