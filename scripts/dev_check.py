"""Developer smoke script for the GPU box: parity numbers + a first timing (not a test)."""
import sys, os, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
from raytracing_amd.backend.lib import HipBackend

be = HipBackend(0)
# RNG
u = be.debug_uniforms(123, 4, h.SEED_RENDER, 6)
lib = h.oracle()
want = np.array([[lib.orc_uniform(123, 4, h.SEED_RENDER, c, i) for i in range(5)] for c in range(6)], np.float32)
print("rng equal:", np.array_equal(u, want))

def cmp(tag, r, ref, mask=None):
    m = np.ones(len(ref["obj"]), bool) if mask is None else mask
    om = (r["obj"][m] != ref["obj"][m]).sum(); pm = (r["prim"][m] != ref["prim"][m]).sum()
    hit = m & (ref["obj"] >= 0) & (r["obj"] == ref["obj"])
    rel = np.abs(r["t"][hit] - ref["t"][hit]) / ref["t"][hit]
    nerr = np.abs(r["normal"][hit] - ref["normal"][hit]).max(axis=1)
    hw = np.bincount(r["how"][m], minlength=3)
    print(f"{tag}: n={m.sum()} obj_mismatch={om} prim_mismatch={pm} t_rel_max={rel.max():.3e} n_err_max={nerr.max():.3e} how={hw.tolist()}")
    return rel, nerr

for name, f in (("room", "rays_room.npz"), ("random_spheres_seed7", "rays_random_spheres.npz")):
    sc = h.packed_scene(name)
    be.set_scene(sc)
    z, _ = h.load_npz(f)
    ref = dict(obj=z["obj"], prim=z["prim"], t=z["t"], normal=z["normal"])
    t0 = time.time(); r = be.intersect(z["org"], z["dir"]); dt = time.time() - t0
    cmp(f"{name} f64 rays ({dt*1e3:.0f} ms)", r, ref)
    # float32-representable versions of the same rays; the oracle answers for exactly these rays
    o32 = z["org"].astype(np.float32).astype(np.float64); d32 = z["dir"].astype(np.float32).astype(np.float64)
    ref32 = h.oracle_intersect(sc, o32, d32)
    r = be.intersect(o32, d32)
    rel, nerr = cmp(f"{name} f32 rays, no origin", r, ref32)
    # bounce rays with their originating primitive
    seg = z["seg"]; prev_g = np.full(len(seg), -1, np.int32)
    refd = h.oracle_intersect(sc, z["org"], z["dir"])
    prev_g[1:] = np.where(seg[1:] > 0, refd["gprim"][:-1], -1)
    r = be.intersect(o32, d32, origin_prim=prev_g)
    with h.origin_rule():
        ref32o = h.oracle_intersect(sc, o32, d32, origin_prim=prev_g)
    cmp(f"{name} f32 rays, origin prim vs origin-rule oracle", r, ref32o)
    prim_rays = seg == 0
    cmp(f"{name} f32 primary rays", r, ref32, prim_rays)

# render
for name, img in (("room", "image_room_96x54_8spp_8b.npz"), ("random_spheres_seed7", "image_random_spheres_320x180_16spp_4b.npz")):
    sc = h.packed_scene(name); be.set_scene(sc)
    z, meta = h.load_npz(img)
    W, H, spp, b = meta["W"], meta["H"], meta["spp"], meta["bounces"]
    be.resize(W, H); be.reset_stats()
    be.render(W, H, 0, spp, b, h.SEED_RENDER); be.synchronize()
    got = be.read().astype(np.float64)
    ref = z["arr"]
    st = be.stats()
    rmse = np.sqrt(np.mean((got - ref) ** 2)); peak = ref.max()
    relrmse = rmse / np.mean(ref)
    big = np.abs(got - ref).max(axis=2) > 1e-3 * spp
    print(f"{name} image {W}x{H}x{spp}: rmse={rmse:.4e} rel_rmse={relrmse:.4e} mean={ref.mean():.4f} differing_pixels={big.sum()}/{W*H} stats={st}")
    orc_arr, orc_rays = h.oracle_render(sc, W, H, spp, b)
    print("   oracle rays", orc_rays, "gpu rays", st["rays"])

sc = h.packed_scene("room"); be.set_scene(sc)
W, H = 1920, 1080
be.resize(W, H)
for spp in (4, 32):
    be.reset_stats(); t0 = time.time()
    be.render(W, H, 0, spp, 8, 1); be.synchronize(); dt = time.time() - t0
    st = be.stats()
    print(f"room 1080p {spp}spp: wall {dt*1e3:.1f} ms kernel {st['kernel_ms']:.1f} ms rays {st['rays']} -> {st['rays']/st['kernel_ms']/1e6:.2f} Grays/s exact={st['exact_rays']} refine={st['refined_hits']}")
sc = h.packed_scene("random_spheres_seed7"); be.set_scene(sc)
for spp in (4, 16):
    be.reset_stats(); t0 = time.time()
    be.render(W, H, 0, spp, 8, 1); be.synchronize(); dt = time.time() - t0
    st = be.stats()
    print(f"spheres 1080p {spp}spp: wall {dt*1e3:.1f} ms kernel {st['kernel_ms']:.1f} ms rays {st['rays']} -> {st['rays']/st['kernel_ms']/1e6:.2f} Grays/s exact={st['exact_rays']} refine={st['refined_hits']}")

# per-path comparison against upstream's per-path radiance
for name, img in (("room", "image_room_96x54_8spp_8b.npz"), ("random_spheres_seed7", "image_random_spheres_320x180_16spp_4b.npz")):
    sc = h.packed_scene(name); be.set_scene(sc)
    z, meta = h.load_npz(img)
    W, H, spp, b = meta["W"], meta["H"], meta["spp"], meta["bounces"]
    if name != "room":
        W2 = 64
        xs = np.arange(0, W, W // W2)
    else:
        xs = np.arange(W)
    xys = np.array([(x, y, s) for x in xs for y in range(H) for s in range(spp)], np.int32)
    tr = be.trace_paths(xys, W, H, b, h.SEED_RENDER)
    ref = z["per_path"][xys[:, 0], xys[:, 1], xys[:, 2]]
    with h.origin_rule():
        refo = np.array([h.oracle_trace_pixel(sc, W, H, int(x), int(y), int(s_), b)[0] for x, y, s_ in xys])
    erro = np.abs(tr["rgb"] - refo).max(axis=1)
    tolo = 1e-4 * np.maximum(1.0, np.abs(refo).max(axis=1))
    print(f"{name}: vs origin-rule oracle: differing {(erro > tolo).sum()} of {len(xys)}; max err among agreeing {erro[erro<=tolo].max():.2e}")
    err = np.abs(tr["rgb"] - ref).max(axis=1)
    tol = 1e-4 * np.maximum(1.0, np.abs(ref).max(axis=1))
    bad = np.nonzero(err > tol)[0]
    print(f"{name}: paths {len(xys)} differing {len(bad)} ({len(bad)/len(xys):.2e}); max err among agreeing {err[err<=tol].max():.2e}; how hist {np.bincount(tr['how'][tr['how']>=0], minlength=3).tolist()}")
    for k in bad[:6]:
        x, y, s = xys[k]
        rgb, rays, nd, seg = h.oracle_trace_pixel(sc, W, H, int(x), int(y), int(s), b, with_segments=True)
        n = tr["n_seg"][k]
        print("  path", x, y, s, "gpu rgb", tr["rgb"][k], "ref", ref[k], "orc", rgb)
        print("     gpu obj", tr["obj"][k][:n].tolist(), "prim", tr["prim"][k][:n].tolist(), "how", tr["how"][k][:n].tolist(), "t", np.round(tr["t"][k][:n], 5).tolist())
        print("     orc obj", seg["obj"].tolist(), "prim", seg["prim"].tolist(), "t", np.round(seg["t"], 5).tolist())
