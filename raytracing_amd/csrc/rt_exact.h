// Double-precision "exact" intersection code: the reference semantics of upstream's
// Sphere.intersect (src/mesh/sphere.py:13-41), Mesh.intersect_aabb / intersect_tri
// (src/mesh/mesh.py:61-143) and Scene.intersect (src/raytracing/scene.py:23-27), evaluated
// operation by operation in upstream's order with no fused multiply-add (the translation unit is
// built with -ffp-contract=off).  It runs only for rays or primitives the float32 filter could
// not decide, so it is written for clarity, not speed.
//
// Ray-origin rule ("origin primitive"): a bounce ray starts on the primitive it just hit.  Upstream
// relies on `depth > 1e-10` in double precision to drop that self-intersection (scene.py:25);
// positions here are float32, so the originating primitive is carried with the ray instead:
// the originating triangle is skipped, and of the originating sphere only the far root counts,
// and only when the ray heads into the sphere.  origin < 0 means "no such primitive" and gives
// upstream's semantics unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include "rt_scene.h"

#include "rt_libm_pow.h"

namespace rt {

#define RT_DEV __device__ __forceinline__

struct D3 { double x, y, z; };
RT_DEV D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV D3 dsub(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV D3 dcross(D3 a, D3 b) { return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_DEV D3 dat(D3 o, D3 d, double t) { return d3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t); }
RT_DEV D3 dnormalize(D3 a) { double l = __builtin_sqrt(ddot(a, a)); return d3(a.x / l, a.y / l, a.z / l); }

struct ExactHit {
    bool hit;
    int32_t prim;   // sphere: 0 near root, 1 far root; triangle: unused
    double t;
    D3 n, pos;
};

// Sphere.intersect restricted to its first hit that survives Scene.intersect's depth filter.
// want_geometry = false skips position and normal (callers that only rank by depth).
// LIBM: radius**2, b**2 and (...)**0.5 as upstream's interpreter computes them, through the C library's pow()
// (rt_libm_pow.h) - the settling code's form, whose answers are final.  The search loops' form (LIBM false) squares
// with a product and takes the root with sqrt: within an ulp of the other, a quarter of its instructions, and what it
// feeds are float32 bounds with margins a million times that.
// `r2`: radius**2 as the host's C library gives it (ColdScene::sph64 holds it behind the spheres), for LIBM callers.
template <bool LIBM = true>
RT_DEV ExactHit exact_sphere(const double *sp, D3 o, D3 d, bool is_origin, bool want_geometry = true, const double *r2 = nullptr) {
    ExactHit h; h.hit = false; h.prim = 0; h.t = 0; h.n = d3(0, 0, 0); h.pos = h.n;
    D3 c = d3(sp[0], sp[1], sp[2]);
    double radius = sp[3];
    D3 ro = dsub(o, c);
    double a = ddot(d, d);
    double b = 2 * ddot(ro, d);
    double cc = ddot(ro, ro) - (LIBM ? (r2 != nullptr ? *r2 : libm_square(radius)) : radius * radius);
    double rad = (LIBM ? libm_square(b) : b * b) - 4 * a * cc;
    if (!(rad >= 0)) return h;
    double sq = LIBM ? libm_root(rad) : __builtin_sqrt(rad);
    double t1 = (-b - sq) / (2 * a), t2 = (-b + sq) / (2 * a);
    double t; int root;
    if (!is_origin) {
        if (t1 > 1e-10) { t = t1; root = 0; }
        else if (t2 > 1e-10) { t = t2; root = 1; }
        else return h;
    } else {
        if (b < 0 && t2 > 1e-10) { t = t2; root = 1; }
        else return h;
    }
    h.hit = true; h.prim = root; h.t = t;
    if (want_geometry) {
        h.pos = dat(o, d, t);
        h.n = dnormalize(dsub(h.pos, c));
    }
    return h;
}

// Mesh.intersect_aabb
RT_DEV bool exact_aabb(const double *bb, D3 o, D3 d) {
    double ix = d.x != 0 ? 1 / d.x : 1e10, iy = d.y != 0 ? 1 / d.y : 1e10, iz = d.z != 0 ? 1 / d.z : 1e10;
    double ax = (bb[0] - o.x) * ix, ay = (bb[1] - o.y) * iy, az = (bb[2] - o.z) * iz;
    double bx = (bb[3] - o.x) * ix, by = (bb[4] - o.y) * iy, bz = (bb[5] - o.z) * iz;
    double lx = bx < ax ? bx : ax, ly = by < ay ? by : ay, lz = bz < az ? bz : az;
    double hx = bx > ax ? bx : ax, hy = by > ay ? by : ay, hz = bz > az ? bz : az;
    double tn = lx; if (ly > tn) tn = ly; if (lz > tn) tn = lz;
    double tf = hx; if (hy < tf) tf = hy; if (hz < tf) tf = hz;
    double lo = tn; if (0 > lo) lo = 0;
    return tf >= lo;
}

// Mesh.intersect_tri + the depth filter
RT_DEV ExactHit exact_tri(const double *tv, D3 o, D3 d, bool want_geometry = true) {
    ExactHit h; h.hit = false; h.prim = 0; h.t = 0; h.n = d3(0, 0, 0); h.pos = h.n;
    D3 pa = d3(tv[0], tv[1], tv[2]), pb = d3(tv[3], tv[4], tv[5]), pc = d3(tv[6], tv[7], tv[8]);
    D3 ab = dsub(pb, pa), ac = dsub(pc, pa);
    D3 nrm = dcross(ab, ac);
    double det = -ddot(d, nrm);
    if (__builtin_fabs(det) < 1e-6) return h;
    double inv_det = 1 / det;
    D3 ao = dsub(o, pa);
    D3 dao = dcross(ao, d);
    double dst = ddot(nrm, ao) * inv_det;
    double u = ddot(ac, dao) * inv_det;
    double v = -ddot(ab, dao) * inv_det;
    double w = 1 - u - v;
    if (dst >= 0 && u >= 0 && v >= 0 && w >= 0 && dst > 1e-10) {
        h.hit = true; h.t = dst;
        if (want_geometry) { h.pos = dat(o, d, dst); h.n = dnormalize(nrm); }
    }
    return h;
}

// Depth only, for a triangle the float32 filter has already proven to be hit (mesh.py:117-130).
RT_DEV double exact_tri_depth(const double *tv, D3 o, D3 d) {
    D3 pa = d3(tv[0], tv[1], tv[2]), pb = d3(tv[3], tv[4], tv[5]), pc = d3(tv[6], tv[7], tv[8]);
    D3 nrm = dcross(dsub(pb, pa), dsub(pc, pa));
    double det = -ddot(d, nrm);
    return ddot(nrm, dsub(o, pa)) * (1 / det);
}

// A sphere that is CERTAINLY hit at a known root (the float32 filter proved both) but whose float32 depth or normal
// is not tight enough: depth, hit position and centre-to-hit vector in double precision.  Unlike the functions above
// this is not upstream's operation sequence - it uses fused multiply-adds, and Newton steps from the float32
// reciprocal and reciprocal-square-root estimates instead of the long division and square-root sequences - so it
// agrees with upstream's doubles to ~1e-15 relative, not bit for bit (its results are rounded to float32 anyway).
// Root as in sphere.py:22 / core/util.py:29-32: (-b -+ sqrt(b^2 - 4ac)) / 2a, here with b halved.
struct SphRefined { float t, px, py, pz, qx, qy, qz; };   // q = pos - centre (length ~ radius; the caller normalises it)

RT_DEV SphRefined refine_sphere64(const double *sp, D3 o, D3 d, int root) {
    const double cx = sp[0], cy = sp[1], cz = sp[2], rad = sp[3];
    const double rx = o.x - cx, ry = o.y - cy, rz = o.z - cz;
    const double a = __builtin_fma(d.z, d.z, __builtin_fma(d.y, d.y, d.x * d.x));
    const double b = __builtin_fma(rz, d.z, __builtin_fma(ry, d.y, rx * d.x));
    const double c = __builtin_fma(rz, rz, __builtin_fma(ry, ry, __builtin_fma(rx, rx, -(rad * rad))));
    double disc = __builtin_fma(b, b, -(a * c));
    disc = disc > 1e-300 ? disc : 1e-300;                         // (positive for a certain hit; stay finite whatever comes)
    double y = (double)__builtin_amdgcn_rsqf((float)disc);        // 1 / sqrt(disc): estimate, two Newton steps
    y = y * __builtin_fma(-0.5 * (disc * y), y, 1.5);
    y = y * __builtin_fma(-0.5 * (disc * y), y, 1.5);
    const double s = disc * y;
    double ia = (double)__builtin_amdgcn_rcpf((float)a);          // 1 / a likewise
    ia = ia * __builtin_fma(-a, ia, 2.0);
    ia = ia * __builtin_fma(-a, ia, 2.0);
    const double t = (root ? (s - b) : (-b - s)) * ia;
    const double px = __builtin_fma(t, d.x, o.x), py = __builtin_fma(t, d.y, o.y), pz = __builtin_fma(t, d.z, o.z);
    SphRefined q;
    q.t = (float)t; q.px = (float)px; q.py = (float)py; q.pz = (float)pz;
    q.qx = (float)(px - cx); q.qy = (float)(py - cy); q.qz = (float)(pz - cz);
    return q;
}

struct ScanHit { int32_t gprim; double t; };   // gprim < 0: miss

// One primitive by global id, with its mesh's bounding-box gate.
RT_DEV ExactHit exact_single(SceneRef S, int32_t gprim, D3 o, D3 d, int32_t origin) {
    const ColdScene &C = *S.cold;
    if (gprim < S.n_spheres) return exact_sphere(C.sph64 + 4 * (size_t)gprim, o, d, gprim == origin, true, C.sph64 + 4 * (size_t)S.n_spheres + gprim);
    int32_t ti = gprim - S.n_spheres;
    ExactHit miss; miss.hit = false; miss.prim = 0; miss.t = 0; miss.n = d3(0, 0, 0); miss.pos = miss.n;
    if (gprim == origin) return miss;
    if (!exact_aabb(C.aabb64 + 6 * (size_t)C.obj_index[S.tri_aux[ti].obj], o, d)) return miss;
    return exact_tri(C.tri64 + 9 * (size_t)ti, o, d);
}

// Depth only of one primitive by global id (same arithmetic as exact_single, so the same depth);
// hit = false when upstream would not report it.
struct ExactDepth { bool hit; double t; };

RT_DEV ExactDepth exact_single_depth(SceneRef S, int32_t gprim, D3 o, D3 d, int32_t origin) {
    const ColdScene &C = *S.cold;
    ExactDepth r; r.hit = false; r.t = 0;
    ExactHit h;
    if (gprim < S.n_spheres) h = exact_sphere(C.sph64 + 4 * (size_t)gprim, o, d, gprim == origin, false, C.sph64 + 4 * (size_t)S.n_spheres + gprim);
    else {
        const int32_t ti = gprim - S.n_spheres;
        if (gprim == origin) return r;
        if (!exact_aabb(C.aabb64 + 6 * (size_t)C.obj_index[S.tri_aux[ti].obj], o, d)) return r;
        h = exact_tri(C.tri64 + 9 * (size_t)ti, o, d, false);
    }
    r.hit = h.hit; r.t = h.t;
    return r;
}

// Scene.intersect(ray)[0]: first minimum of depth in object order, then per-object hit order.
// Returns the identity and depth only; the caller re-evaluates the winner for position / normal
// (same arithmetic, so the same depth).
RT_DEV ScanHit exact_scan(SceneRef S, D3 o, D3 d, int32_t origin) {
    const ColdScene &C = *S.cold;
    ScanHit best; best.gprim = -1; best.t = __builtin_inf();
    for (int oi = 0; oi < C.n_objects; ++oi) {
        int idx = C.obj_index[oi];
        if (C.obj_kind[oi] == 0) {
            ExactHit h = exact_sphere(C.sph64 + 4 * (size_t)idx, o, d, idx == origin, false, C.sph64 + 4 * (size_t)S.n_spheres + idx);
            if (h.hit && h.t < best.t) { best.gprim = idx; best.t = h.t; }
        } else {
            if (!exact_aabb(C.aabb64 + 6 * (size_t)idx, o, d)) continue;
            int t0 = C.mesh_tri_begin[idx], t1 = C.mesh_tri_begin[idx + 1];
            for (int ti = t0; ti < t1; ++ti) {
                if (S.n_spheres + ti == origin) continue;
                ExactHit h = exact_tri(C.tri64 + 9 * (size_t)ti, o, d, false);
                if (h.hit && h.t < best.t) { best.gprim = S.n_spheres + ti; best.t = h.t; }
            }
        }
    }
    return best;
}

}  // namespace rt
