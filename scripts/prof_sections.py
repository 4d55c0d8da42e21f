"""Section timers of the render loop (developer tool; needs a library built with -DRT_SECTION_TIMERS:
    hipcc <flags of raytracing_amd/backend/build.py> -DRT_SECTION_TIMERS -o build_ab/_rt_hip_prof.so raytracing_amd/csrc/rt_api.hip
    RAYTRACING_AMD_LIB=build_ab/_rt_hip_prof.so python scripts/prof_sections.py [scene] [spp])."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as h
from raytracing_amd.backend.lib import HipBackend
NAMES = ["handout", "begin", "ray_setup", "spheres", "big_spheres", "quad_rows", "slab_rows", "tri_rows", "judge", "settle",
         "record", "sky", "draws", "bounce", "accum", "stats", "walk", "leaves"]
COUNTS = ["trips", "lanes_with_job", "lanes_starting", "lanes_ending", "walk_iters", "walk_lanes", "leaf_iters", "leaf_lanes"]
name = sys.argv[1] if len(sys.argv) > 1 else "room"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
be = HipBackend(0)
if name.startswith("horse"):   # room + the chess horse, every face split in four `level` times: horse6 = 1.18 M triangles
    import dataclasses
    base = h.packed_scene("room_horse")
    lv = int(name[5:] or 6)
    sc = base.with_subdivided_mesh(base.n_meshes - 1, lv) if lv else base
    be.set_scene(dataclasses.replace(sc, library_bvh=True))
else:
    be.set_scene(h.packed_scene(name))
W, H = 1920, 1080
be.resize(W, H)
be.render(W, H, 0, 4, 8, 1); be.synchronize()
be.reset_stats()
be.render(W, H, 0, spp, 8, 1); be.synchronize()
st = be.stats()
N = 26
raw = (C.c_uint64 * (2 * N))()
be.lib.rt_debug_section_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
be.lib.rt_debug_section_cycles(be.handle, raw, 2 * N)
iters = st["rays"] / 64.0
print(f"{name} {spp} spp: kernel {st['kernel_ms']:.2f} ms, rays {st['rays']}, wave-iterations {iters:.3e}; set aside: {be.render_counters()}")
for bank, title in ((0, "lean kernel (+ jobs rendered again)"), (1, "full kernel over the paths set aside")):
    out = list(raw)[bank * N:bank * N + 18]; counts = list(raw)[bank * N + 18:(bank + 1) * N]
    tot = sum(out)
    if tot == 0: continue
    print(f" {title}: cycles/iteration (wave view) {tot / iters:.0f}")
    for n, v in zip(NAMES, out):
        print(f"  {n:12s} {100.0 * v / tot:6.2f} %   {v / iters:8.1f} cycles / iteration")
    trips = max(counts[0], 1)
    print(f"  trips of the render loop {counts[0]}; lanes per trip: with a job {counts[1] / trips:.1f}, starting a query {counts[2] / trips:.1f}, closing one {counts[3] / trips:.1f} (hierarchy scenes)")
    if counts[4]:
        print(f"  box loop: {counts[4] / trips:.2f} passes per trip, {counts[5] / counts[4]:.1f} lanes in a pass; triangle loop: {counts[6] / trips:.2f} passes per trip, "
              f"{counts[7] / max(counts[6], 1):.1f} lanes in a pass")
