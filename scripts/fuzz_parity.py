"""Fuzz campaign (developer tool; runs on the GPU box): the scenes and rays of tests/fuzz_scenes.py, seed after seed -
hit identity, depth, normal and position of rt_intersect_batch against the oracle through the linear tables and
through a library-built hierarchy (free rays, bounce rays with their origin primitive, rays given as arbitrary
doubles), then a small render of each scene against the oracle's.

    python scripts/fuzz_parity.py [first_seed] [n_seeds] [seconds] [big]

`big`: fz.make_big_scene instead (12,000 clustered triangles, hierarchy only).

Every disagreement is printed and the offending rays are written to gpurun_out/fuzz_seed<k>_<stage>.npz."""
import dataclasses, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers as h
import fuzz_scenes as fz
from raytracing_amd.backend.lib import HipBackend

OUT = os.path.join(REPO, "gpurun_out")


def compare(tag, got, want, o, d, seed, sc, origin=None):
    off, tie = fz.check_hits(got, want, o, sc)
    if off.any():
        print(f"  !! seed {seed} {tag}: {int(off.sum())} of {len(o)} rays off", flush=True)
        i = np.nonzero(off)[0][:200]
        np.savez(os.path.join(OUT, f"fuzz_seed{seed}_{tag}.npz"), o=o[i], d=d[i], origin=(origin[i] if origin is not None else np.zeros(0)),
                 got_obj=got["obj"][i], got_prim=got["prim"][i], got_t=got["t"][i], got_how=got["how"][i],
                 want_obj=want["obj"][i], want_prim=want["prim"][i], want_t=want["t"][i])
    return int(off.sum()), int(tie.sum())


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 240.0
    big = len(sys.argv) > 4 and sys.argv[4] == "big"
    os.makedirs(OUT, exist_ok=True)
    pool = fz.materials_pool()
    be = HipBackend(0)
    t0 = time.time()
    rays = off = ties = px_all = px_off = paths_off = paths_wrong = paths_shading = paths_sky = seeds = 0
    for seed in range(first, first + count):
        if time.time() - t0 > budget:
            break
        seeds += 1
        sc, rng = fz.make_big_scene(seed, pool) if big else fz.make_scene(seed, pool)
        o, d = fz.make_rays(sc, rng, 6000)
        want = h.oracle_intersect(sc, o, d)
        hit = want["obj"] >= 0
        o2 = want["pos"][hit].astype(np.float32).astype(np.float64); g2 = want["gprim"][hit].astype(np.int32)
        d2 = fz.unit32(rng.normal(size=o2.shape))
        with h.origin_rule():
            want2 = h.oracle_intersect(sc, o2, d2, origin_prim=g2)
        od = rng.uniform(-1.6, 1.6, (4000, 3)); dd = rng.normal(size=(4000, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        wantd = h.oracle_intersect(sc, od, dd)                 # doubles as they come: the exact code answers
        for mode in (("hierarchy",) if big else ("linear", "hierarchy")):
            if mode == "linear":
                with h.linear_tables():
                    be.set_scene(sc)
            else:
                be.set_scene(dataclasses.replace(sc, library_bvh=True))
            for tag, got, w, oo, ddd, org in (("free", be.intersect(o, d), want, o, d, None),
                                              ("bounce", be.intersect(o2, d2, origin_prim=g2), want2, o2, d2, g2),
                                              ("double", be.intersect(od, dd), wantd, od, dd, None)):
                a, b = compare(f"{mode}_{tag}", got, w, oo, ddd, seed, sc, org)
                off += a; ties += b
            npx, ndiff, nwrong, nshade, nsky = fz.check_render(be, sc)
            px_all += 48 * 27; px_off += npx; paths_off += ndiff; paths_wrong += nwrong; paths_shading += nshade; paths_sky += nsky
            if nwrong or nshade:
                print(f"  !! seed {seed} {mode} render: {nwrong} paths whose own ray the oracle answers differently, {nshade} with equal hits and another radiance", flush=True)
        rays += (1 if big else 2) * (len(o) + len(o2) + len(od))
        if seeds % (10 if big else 50) == 0:
            print(f"{seeds} seeds, {rays} rays: {off} off, {ties} within sphere depth noise; pixels beyond 1e-3: {px_off} of {px_all} "
                  f"({paths_off} paths, {paths_wrong} wrong on their own ray, {paths_shading} shading, {paths_sky} sky texels); {time.time() - t0:.0f} s", flush=True)
    print(f"TOTAL seeds {first}..{first + seeds - 1}: {rays} rays, {off} off, {ties} answers within upstream's own sphere depth noise (libm pow); "
          f"rendered pixels beyond 1e-3: {px_off} of {px_all}, on {paths_off} paths - {paths_wrong} answered differently by the oracle on the kernel's own ray, "
          f"{paths_shading} with equal hits and another radiance, {paths_sky} with equal hits and a neighbouring sky texel")


if __name__ == "__main__":
    main()
