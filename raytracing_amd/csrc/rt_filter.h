// float32 closest-hit search with conservative error bounds ("filter"), and the logic that turns
// its answer into a hit that equals upstream's Scene.intersect(ray)[0].
//
// Contract of the filter: a primitive that upstream's double-precision arithmetic would report
// as hit is ALWAYS a candidate here, and every candidate carries a lower bound t_lo <= its true
// depth.  Candidates are ordered by a 32-bit key = (bits of t_lo with the low bits cleared) | id;
// positive floats order like their bit patterns, so one v_min_u32 keeps the best candidate and
// one v_med3_u32 the runner-up.  After the loops the winner is re-evaluated with per-lane loads:
//   * certainly hit, certainly alone in front, precise enough  -> accepted as is (float32);
//   * anything else -> that primitive (or, for a possible tie / a rejected winner, the whole
//     scene) is re-evaluated by the double-precision code in rt_exact.h.
// So hit identity never depends on float32 rounding; float32 only decides how often the slow
// path runs.
#pragma once
#include <hip/hip_runtime.h>
#include "rt_exact.h"
#include "rt_scene.h"

namespace rt {

RT_DEV uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RT_DEV float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
RT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RT_DEV float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
RT_DEV float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
RT_DEV float rsq_(float x) { return __builtin_amdgcn_rsqf(x); }
RT_DEV float min_(float a, float b) { return __builtin_fminf(a, b); }
RT_DEV float max_(float a, float b) { return __builtin_fmaxf(a, b); }
RT_DEV float abs_(float a) { return __builtin_fabsf(a); }
RT_DEV uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }
RT_DEV uint32_t umed3_(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Wave-uniform table reads.  A load through a constant-address-space pointer at a wave-uniform
// address is issued on the scalar unit (s_load_dwordx4/x8/x16): the row lands in SGPRs and feeds
// the VALU as a scalar operand, costing no vector instruction and no VGPR.  (Through an ordinary
// global pointer the compiler cannot prove the tables are not written by this kernel's own stores
// and falls back to per-lane vector loads.)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const f4 __attribute__((address_space(4))) *const_f4_ptr;
RT_DEV const_f4_ptr as_const_f4(const void *p) { return (const_f4_ptr)(const f4 *)p; }

RT_DEV TriRow load_tri_row_uniform(const TriRow *rows, int i) {
    const_f4_ptr q = as_const_f4(rows) + 4 * (size_t)i;
    f4 a = q[0], b = q[1], c = q[2], d = q[3];
    TriRow T;
    T.nx = a.x; T.ny = a.y; T.nz = a.z; T.kn = a.w;
    T.e1x = b.x; T.e1y = b.y; T.e1z = b.z; T.k1 = b.w;
    T.e2x = c.x; T.e2y = c.y; T.e2z = c.z; T.k2 = c.w;
    T.cb = d.x; T.ct = d.y; T.det_lo = d.z;
    const float id_bits = d.w;   // copy out first: bit-casting a vector element expression reads lane 0
    T.gprim = (int32_t)f2u(id_bits);
    return T;
}

RT_DEV SphRow load_sph_row_uniform(const SphRow *rows, int i) {
    const_f4_ptr q = as_const_f4(rows) + 2 * (size_t)i;
    f4 a = q[0], b = q[1];
    SphRow P;
    P.cx = a.x; P.cy = a.y; P.cz = a.z; P.r2 = a.w;
    P.qs = b.x; P.cn = b.y; P.r = b.z;
    const float id_bits = b.w;
    P.gprim = (int32_t)f2u(id_bits);
    return P;
}

constexpr uint32_t kKeyNone = 0x7F800000u;   // keys at or above +inf's pattern mean "no candidate"
constexpr float kInf = __builtin_inff();
constexpr float kTCertain = 1e-9f;           // a depth bound above this certainly passes upstream's 1e-10 filter
// A float32 depth / normal is accepted as is when its error BOUND is below this fraction; otherwise the
// winning primitive is redone in double precision.  The closest-hit entry point uses the strict value
// (depth and normals then agree with upstream's doubles to 1e-5 relative, the stated contract).  The
// render kernel uses the loose one: hit identity is exact either way, the bound is only exceeded at
// grazing incidence (|cos| < ~0.1), and a depth good to 2e-4 of itself there moves a bounce origin
// by less than a tenth of a pixel footprint - while redoing it costs the WHOLE wave a
// double-precision division whenever any of its 64 lanes grazes (measured: 13 % of kernel time).
constexpr float kRelAcceptStrict = 9.5e-6f;
constexpr float kRelAcceptRender = 2e-4f;

// A ray prepared for the filter.  All bounds assume 1/2 <= |d| <= 2 and no exactly-zero component
// of d (such rays go straight to the exact code: see ray_needs_exact).
struct Ray {
    float ox, oy, oz, dx, dy, dz;
    float a, inv_a, dlen;  // d.d, 1/(d.d), |d|
    float o2;              // |o| (rounded up)
    float c4, c7;          // triangle bounds: q = c7 |t| + c4 with c4 = 4u S, c7 = 7u |d|, S = |o| + scene_radius
    float band0;           // crude per-ray discriminant bound for the in-loop sphere test
    int32_t origin;        // global id of the primitive the ray starts on, -1 = none
};

// `rounded` = the ray is the float32 rounding of a double-precision ray the exact code will use
// (camera rays): every bound then also covers u per component of o and d.
RT_DEV Ray make_ray(const DevScene &S, float ox, float oy, float oz, float dx, float dy, float dz, int32_t origin,
                    bool rounded) {
    Ray r;
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
    r.a = fma_(dz, dz, fma_(dy, dy, dx * dx));
    r.inv_a = rcp_(r.a);
    r.dlen = sqrt_(r.a);
    r.o2 = sqrt_(fma_(oz, oz, fma_(oy, oy, ox * ox))) * (1.0f + 4.0f * kU);
    float Sr = r.o2 + S.scene_radius;
    const float k = rounded ? 1.3f * kU : 1.01f * kU;
    r.c4 = 4.0f * k * Sr;
    r.c7 = 7.0f * k * r.dlen;
    r.band0 = 72.0f * kU * r.a * Sr * Sr;
    r.origin = origin;
    return r;
}

RT_DEV bool ray_needs_exact(const Ray &r) {
    bool ok = (r.a >= 0.25f) && (r.a <= 4.0f) && (r.dx != 0.0f) && (r.dy != 0.0f) && (r.dz != 0.0f);
    return !(ok && r.o2 < 1e18f);
}

// ---------------------------------------------------------------------------------------------
// Triangles
//
// Error model (u = 2^-24, every operation rounds once, table entries are rounded doubles):
//   nd = N.d            |err| <= 4u |N||d|
//   no = N.o + kN       |err| <= 4u |N| (|a| + |o|)
//   t  = -no * rcp(nd)  |err| <= u k [4 (|a|+|o|)/|d| + 7 |t|],   k = |N||d| / |nd|  (1 / |cos|)
//   u,v at p = o + t d  |err| <= |e| u k [9 (|a|+|o|) + 12 |t||d|]
// With q = 4u S + 7u |t||d| (S >= |a| + |o|) and g = q / |nd|:   t bound = g |N|,
// barycentric bound (for min(u, v, 1-u-v)) = g * cb,  cb = 2.25 (|e1|+|e2|) |N| + 2 |N| / S_min.
// ---------------------------------------------------------------------------------------------
struct TriEval { float m, t, g, nd; };

RT_DEV TriEval tri_eval(const TriRow &T, const Ray &r) {
    TriEval e;
    e.nd = fma_(T.nz, r.dz, fma_(T.ny, r.dy, T.nx * r.dx));
    float no = fma_(T.nz, r.oz, fma_(T.ny, r.oy, fma_(T.nx, r.ox, T.kn)));
    float inv = rcp_(e.nd);
    e.t = -no * inv;
    float px = fma_(e.t, r.dx, r.ox), py = fma_(e.t, r.dy, r.oy), pz = fma_(e.t, r.dz, r.oz);
    float u = fma_(T.e1z, pz, fma_(T.e1y, py, fma_(T.e1x, px, T.k1)));
    float v = fma_(T.e2z, pz, fma_(T.e2y, py, fma_(T.e2x, px, T.k2)));
    float w = (1.0f - u) - v;
    e.m = min_(min_(u, v), w);
    e.g = fma_(r.c7, abs_(e.t), r.c4) * abs_(inv);
    return e;
}

struct Best { uint32_t k0, k1; };  // best and runner-up keys

RT_DEV void best_insert(Best &b, uint32_t key) {
    b.k1 = umed3_(b.k0, b.k1, key);
    b.k0 = umin_(b.k0, key);
}

// Second-pass sink.  When the first pass cannot name the hit on its own (a runner-up whose lower
// bound is not behind the winner's upper bound, or a winner the double-precision code rejected),
// the tables are walked once more and every candidate whose key is at or below `limit` is
// evaluated by the exact code on the spot; the best of those under upstream's ordering (depth,
// then object order, then face order: scene.py:23-27, mesh.py:101-110) is the hit.  A primitive
// whose key is above `limit` has a true depth beyond a proven hit and cannot win or tie.
struct Resolve {
    const DevScene *S;
    D3 o, d;
    int32_t origin, skip;      // skip: the first-pass winner, already evaluated
    uint32_t limit, keep_mask;
    ExactHit e; int32_t gprim; uint64_t order;   // best so far; gprim < 0: none
};

RT_DEV uint64_t scene_order(const DevScene &S, int32_t gprim) {
    const int32_t obj = gprim < S.n_spheres ? S.sph_obj[gprim] : S.tri_aux[gprim - S.n_spheres].obj;
    return ((uint64_t)(uint32_t)obj << 32) | (uint32_t)gprim;
}

RT_DEV void resolve_insert(Resolve &R, uint32_t key) {
    if (key <= R.limit) {
        const DevScene &S = *R.S;
        const int32_t g = (int32_t)(key & S.key_mask);
        if (g != R.skip) {
            const ExactHit x = exact_single(S, g, R.o, R.d, R.origin);
            if (x.hit) {
                const uint64_t ord = scene_order(S, g);
                if (R.gprim < 0 || x.t < R.e.t || (x.t == R.e.t && ord < R.order)) {
                    R.e = x; R.gprim = g; R.order = ord;
                    const float up = (float)x.t * (1.0f + 2.0f * kU);   // >= x.t
                    const uint32_t lim = (f2u(up) & R.keep_mask) | S.key_mask;
                    R.limit = umin_(R.limit, lim);
                }
            }
        }
    }
}

struct BestSink { Best b; RT_DEV void operator()(uint32_t key) { best_insert(b, key); } };
struct ResolveSink { Resolve R; RT_DEV void operator()(uint32_t key) { resolve_insert(R, key); } };

RT_DEV uint32_t tri_key(const TriRow &T, const Ray &r, uint32_t keep_mask) {
    TriEval e = tri_eval(T, r);
    float mp = fma_(e.g, T.cb, e.m);
    float thi = fma_(e.g, T.ct, e.t);
    float tlo = fma_(-e.g, T.ct, e.t);
    float x = min_(mp, thi);
    float y = x * (-kInf);                 // x < 0 -> +inf (not a candidate); x >= 0 -> -inf / NaN
    float tl = max_(max_(tlo, 0.0f), y);   // NaNs drop out of max: a NaN anywhere keeps the candidate
    uint32_t key = (f2u(tl) & keep_mask) | (uint32_t)T.gprim;
    // det_lo = +inf marks the padding rows at the end of the table: never candidates (NaN included)
    const bool kill = (T.gprim == r.origin) || !(abs_(e.nd) >= T.det_lo);
    return kill ? 0xFFFFFFFFu : key;
}

// Brute-force loop over the live triangles, two rows per trip.  The table is padded with dead rows
// to an even count plus two, so the loop needs no tail and may read one trip ahead: each row's
// scalar load is issued a full test (~37 vector instructions) before the row is used.
template <class Sink>
RT_DEV void search_tris(const DevScene &S, const Ray &r, Sink &sink, uint32_t keep_mask) {
    const int n = S.n_tri_rows_padded;   // even; rows [n, n+2) exist and are dead
    if (n <= 0) return;
    const TriRow *rows = S.tri_rows;
    TriRow a0 = load_tri_row_uniform(rows, 0), a1 = load_tri_row_uniform(rows, 1);
    for (int i = 0; i < n; i += 2) {
        sink(tri_key(a0, r, keep_mask));
        a0 = load_tri_row_uniform(rows, i + 2);
        sink(tri_key(a1, r, keep_mask));
        a1 = load_tri_row_uniform(rows, i + 3);
    }
}

// Winner re-evaluation.  status: 0 accept float32 result, 1 identity certain but imprecise,
// 2 undecided (ask the exact code about this primitive).
struct Verdict { int status; float t, thi; };

RT_DEV Verdict judge_tri(const DevScene &S, int32_t ti, const Ray &r, float rel_accept) {
    const TriRow T = S.tri_by_id[ti];
    TriEval e = tri_eval(T, r);
    float bb = e.g * T.cb, bt = e.g * T.ct;
    bool certain = (e.m - bb >= 0.0f) && (e.t - bt > kTCertain) && (abs_(e.nd) >= S.tri_aux[ti].det_hi);
    Verdict v;
    v.t = e.t; v.thi = e.t + bt;
    v.status = !certain ? 2 : ((bt <= rel_accept * e.t) ? 0 : 1);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Small spheres
// ---------------------------------------------------------------------------------------------
struct SphBounds { float t1lo, t1hi, t2lo, t2hi, b, eb; bool maybe, surely; float t1, t2; };

// Bounds on both roots from the cancellation-free form (t - tc)^2 = (r^2 - |ro + tc d|^2) / a.
RT_DEV SphBounds sphere_bounds(const SphRow &P, const Ray &r) {
    SphBounds o;
    float rox = r.ox - P.cx, roy = r.oy - P.cy, roz = r.oz - P.cz;
    float b = fma_(roz, r.dz, fma_(roy, r.dy, rox * r.dx));
    float m2 = fma_(roz, roz, fma_(roy, roy, rox * rox));
    float mlen = sqrt_(m2);
    float tc = -b * r.inv_a;
    float lx = fma_(tc, r.dx, rox), ly = fma_(tc, r.dy, roy), lz = fma_(tc, r.dz, roz);
    float l2 = fma_(lz, lz, fma_(ly, ly, lx * lx));
    float h2 = P.r2 - l2;
    // per-component error of l: rounding of ro (u|ro|), of the centre (u|c|), of tc*d and its sum (~4u|ro|)
    const float o1 = r.o2;   // a rounded camera-ray origin is off by u|o|
    float dl = kU * (7.0f * mlen + 1.5f * P.cn + o1);
    float dh2 = 3.5f * sqrt_(l2) * dl + 3.0f * dl * dl + 5.0f * kU * (l2 + P.r2);
    float hhi = h2 + dh2, hlo = h2 - dh2;
    o.maybe = hhi >= 0.0f;
    o.surely = hlo > 0.0f;
    float shi = sqrt_(max_(hhi, 0.0f) * r.inv_a) * (1.0f + 8.0f * kU);
    float slo = sqrt_(max_(hlo, 0.0f) * r.inv_a) * (1.0f - 8.0f * kU);
    float smid = sqrt_(max_(h2, 0.0f) * r.inv_a);
    // b = ro.d has abs error <= u|d|(5|ro| + 2|c|); a and 1/a carry <= 6u relative
    float ebb = kU * r.dlen * (6.0f * mlen + 2.0f * P.cn + o1);
    float eb = ebb * r.inv_a * 1.001f + 7.0f * kU * abs_(tc);
    o.t1lo = tc - shi - eb; o.t1hi = tc - slo + eb;
    o.t2lo = tc + slo - eb; o.t2hi = tc + shi + eb;
    o.t1 = tc - smid; o.t2 = tc + smid;
    o.b = b; o.eb = ebb;
    return o;
}

template <class Sink>
RT_DEV void search_spheres(const DevScene &S, const Ray &r, Sink &sink, uint32_t keep_mask) {
    const int n = S.n_sph_rows;
    const SphRow *rows = S.sph_rows;
#pragma unroll 2
    for (int i = 0; i < n; ++i) {
        const SphRow P = load_sph_row_uniform(rows, i);
        float rox = r.ox - P.cx, roy = r.oy - P.cy, roz = r.oz - P.cz;
        float b = fma_(roz, r.dz, fma_(roy, r.dy, rox * r.dx));
        float cq = fma_(roz, roz, fma_(roy, roy, fma_(rox, rox, -P.r2)));
        float disc = fma_(b, b, -(r.a * cq));
        if (disc >= -r.band0) {
            // Candidate.  Only a cheap, loose lower bound on the depth is formed here (band0 bounds the
            // discriminant error for every sphere of the scene); the winner's precise bounds are
            // worked out once, after the loops.
            const float sh = sqrt_(disc + r.band0) * (1.0f + 4.0f * kU);
            const float eb = kU * r.dlen * (6.0f * (abs_(rox) + abs_(roy) + abs_(roz)) + 2.0f * P.cn + r.o2);
            const float nb = -b;
            const float t1lo = (nb - sh - eb) * r.inv_a, t2hi = (nb + sh + eb) * r.inv_a, t2lo = (nb - eb) * r.inv_a;
            bool valid; float tl;
            if (P.gprim == r.origin) {           // only the far root, only when heading inwards
                valid = b < eb;
                tl = t2lo;
            } else {
                valid = t2hi > 0.0f;
                tl = t1lo;
            }
            tl = max_(tl, 0.0f) * (1.0f - 8.0f * kU);
            uint32_t key = valid ? ((f2u(tl) & keep_mask) | (uint32_t)P.gprim) : 0xFFFFFFFFu;
            sink(key);
        }
    }
}

// The few huge spheres: double precision throughout.
template <class Sink>
RT_DEV void search_big_spheres(const DevScene &S, const Ray &r, D3 o64, D3 d64, Sink &sink, uint32_t keep_mask) {
    for (int i = 0; i < S.n_big_sph; ++i) {
        int si = S.big_sph[i];
        ExactHit eb = exact_sphere(S.cold->sph64 + 4 * (size_t)si, o64, d64, si == r.origin, false);
        float tl = (float)eb.t * (1.0f - 4.0f * kU);
        uint32_t key = eb.hit ? ((f2u(tl) & keep_mask) | (uint32_t)si) : 0xFFFFFFFFu;
        sink(key);
    }
}

struct SphVerdict { int status; int root; float t, thi; };

RT_DEV SphVerdict judge_sphere(const DevScene &S, int32_t si, const Ray &r, float rel_accept) {
    const SphRow P = S.sph_by_id[si];
    SphBounds o = sphere_bounds(P, r);
    SphVerdict v; v.status = 2; v.root = 0; v.t = 0; v.thi = 0;
    float lo, hi, t; int root; bool certain;
    if (si == r.origin) {
        certain = o.surely && (o.b < -o.eb) && (o.t2lo > kTCertain);
        lo = o.t2lo; hi = o.t2hi; t = o.t2; root = 1;
    } else if (o.t1lo > kTCertain) {
        certain = o.surely; lo = o.t1lo; hi = o.t1hi; t = o.t1; root = 0;
    } else {
        certain = o.surely && (o.t1hi < 0.0f) && (o.t2lo > kTCertain);
        lo = o.t2lo; hi = o.t2hi; t = o.t2; root = 1;
    }
    if (!certain) return v;
    v.root = root; v.t = t; v.thi = hi;
    // precise enough?  depth to rel_accept relative, and the normal (pos - c)/r to rel_accept
    float half = 0.5f * (hi - lo);
    float perr = half * r.dlen + 3.0f * kU * (r.o2 + P.cn + t * r.dlen) * 1.75f;
    bool precise = (half <= rel_accept * t) && (perr <= rel_accept * P.r);
    v.status = precise ? 0 : 1;
    return v;
}

// ---------------------------------------------------------------------------------------------
// Closest hit
// ---------------------------------------------------------------------------------------------
struct Hit {
    int32_t obj, prim, gprim;   // obj < 0: miss
    float t, nx, ny, nz, px, py, pz;
    int32_t how;                // 0 float32 filter alone, 1 winner re-evaluated in double, 2 exact scene scan
};

struct HitCounters { uint32_t exact, refine; };

// The exact code's view of a ray.  Bounce rays and caller-supplied rays ARE their float32 values.
// A pinhole camera ray is the float32 rounding of the ray upstream computes in double precision
// (camera.py:39-42); k_raygen keeps the rounding residue (double - float32, as a float32) per pixel,
// and the exact code adds it back, so that camera rays which graze an edge - with a 90 degree
// field of view the image diagonals of the room scene run exactly along the cube's edges - resolve
// as upstream resolves them.  `table` is null for every other ray.
struct RayResidue { const float *table; uint32_t pixel; float ox, oy, oz; };

RT_DEV RayResidue no_residue() { RayResidue r; r.table = nullptr; r.pixel = 0; r.ox = r.oy = r.oz = 0.0f; return r; }

// The rare passes live out of line: inlined, their register needs would be charged to the hot loops.
template <bool HAS_TRI, bool HAS_SPH>
RT_DEV void resolve_pass(const DevScene &S, const Ray &r, D3 o64, D3 d64, ResolveSink &rs) {
    if (HAS_SPH) {
        search_spheres(S, r, rs, rs.R.keep_mask);
        search_big_spheres(S, r, o64, d64, rs, rs.R.keep_mask);
    }
    if (HAS_TRI) search_tris(S, r, rs, rs.R.keep_mask);
}

RT_DEV void exact_scan_pass(const DevScene &S, D3 o64, D3 d64, int32_t origin, ExactHit &e, int32_t &gprim) {
    ScanHit sh = exact_scan(S, o64, d64, origin);
    gprim = sh.gprim;
    e.hit = false;
    if (gprim >= 0) e = exact_single(S, gprim, o64, d64, origin);
    if (!e.hit) gprim = -1;
}

// `e64`, when given, receives the double-precision record whenever the exact code produced the hit
// (e64->hit says whether it did).
template <bool HAS_TRI, bool HAS_SPH, bool STRICT>
RT_DEV Hit closest_hit(const DevScene &S, const Ray &r, HitCounters &cnt, ExactHit *e64, const RayResidue rr) {
    Hit h; h.obj = -1; h.prim = -1; h.gprim = -1; h.t = kInf; h.nx = h.ny = h.nz = 0; h.px = h.py = h.pz = 0; h.how = 0;
    // only the rare double-precision branches below read these; the loads and conversions sink into them
    float rdx = 0.0f, rdy = 0.0f, rdz = 0.0f, rox = 0.0f, roy = 0.0f, roz = 0.0f;
    if (rr.table) {
        const f4 q = *(const f4 *)(rr.table + 4 * (size_t)rr.pixel);
        rdx = q.x; rdy = q.y; rdz = q.z; rox = rr.ox; roy = rr.oy; roz = rr.oz;
    }
    const D3 o64 = d3((double)r.ox + (double)rox, (double)r.oy + (double)roy, (double)r.oz + (double)roz);
    const D3 d64 = d3((double)r.dx + (double)rdx, (double)r.dy + (double)rdy, (double)r.dz + (double)rdz);
    ExactHit e; e.hit = false; e.prim = 0; e.t = 0; e.n = d3(0, 0, 0); e.pos = e.n;
    bool scan = ray_needs_exact(r);
    int32_t gprim = -1; int root = 0; float t = kInf;
    if (!scan) {
        const uint32_t keep_mask = ~S.key_mask;
        BestSink first; first.b.k0 = 0xFFFFFFFFu; first.b.k1 = 0xFFFFFFFFu;
        if (HAS_SPH) {
            search_spheres(S, r, first, keep_mask);
            search_big_spheres(S, r, o64, d64, first, keep_mask);
        }
        if (HAS_TRI) search_tris(S, r, first, keep_mask);
        const Best b = first.b;
        bool second = false; uint32_t limit = 0; int32_t skip = -1;
        if (b.k0 < kKeyNone) {
            gprim = (int32_t)(b.k0 & S.key_mask);
            const float second_lo = u2f(b.k1 & keep_mask);
            const bool has_second = b.k1 < kKeyNone;
            int status; float thi;
            if (gprim < S.n_spheres) {
                SphVerdict v = judge_sphere(S, gprim, r, STRICT ? kRelAcceptStrict : kRelAcceptRender);
                status = v.status; t = v.t; thi = v.thi; root = v.root;
            } else {
                Verdict v = judge_tri(S, gprim - S.n_spheres, r, STRICT ? kRelAcceptStrict : kRelAcceptRender);
                status = v.status; t = v.t; thi = v.thi;
            }
            skip = gprim;
            if (status == 1 && gprim >= S.n_spheres) {
                // a triangle that is certainly the hit but whose float32 depth is too coarse (grazing
                // incidence): only the depth is redone in double precision
                cnt.refine++;
                t = (float)exact_tri_depth(S.cold->tri64 + 9 * (size_t)(gprim - S.n_spheres), o64, d64);
                thi = t * (1.0f + 2.0f * kU);
                h.how = 1;
            } else if (status != 0) {   // undecided, or an imprecise sphere: double precision for this primitive
                cnt.refine++;
                e = exact_single(S, gprim, o64, d64, r.origin);
                if (e.hit) { t = (float)e.t; thi = t * (1.0f + 2.0f * kU); root = e.prim; }
            }
            if (status == 2 && !e.hit) {
                // the winner was not a hit after all: every other candidate is open
                second = has_second; limit = kKeyNone - 1u; gprim = -1;
            } else if (has_second && !(second_lo > thi)) {
                // someone else may be in front, or tie: the contenders are settled in double precision
                second = true; limit = (f2u(thi) & keep_mask) | S.key_mask;
                if (!e.hit) e = exact_single(S, gprim, o64, d64, r.origin);
                if (!e.hit) { limit = kKeyNone - 1u; gprim = -1; }   // cannot happen for a certain winner; stay safe
            }
        }
        if (second) {
            cnt.exact++;
            h.how = 2;
            ResolveSink rs;
            rs.R.S = &S; rs.R.o = o64; rs.R.d = d64; rs.R.origin = r.origin; rs.R.skip = skip;
            rs.R.limit = limit; rs.R.keep_mask = keep_mask;
            rs.R.e = e; rs.R.gprim = gprim; rs.R.order = gprim >= 0 ? scene_order(S, gprim) : 0;
            resolve_pass<HAS_TRI, HAS_SPH>(S, r, o64, d64, rs);
            e = rs.R.e; gprim = rs.R.gprim;
            if (gprim < 0) e.hit = false;
        }
    }
    if (scan) {   // upstream's Scene.intersect, literally, in double precision
        cnt.exact++;
        h.how = 2;
        exact_scan_pass(S, o64, d64, r.origin, e, gprim);
    }
    if (e64) *e64 = e;
    if (gprim < 0) return h;
    h.gprim = gprim;
    if (gprim < S.n_spheres) {
        h.obj = S.sph_obj[gprim]; h.prim = e.hit ? e.prim : root;
    } else {
        const TriAux A = S.tri_aux[gprim - S.n_spheres];
        h.obj = A.obj; h.prim = A.face;
    }
    if (e.hit) {
        if (h.how == 0) h.how = 1;
        h.t = (float)e.t; h.nx = (float)e.n.x; h.ny = (float)e.n.y; h.nz = (float)e.n.z;
        h.px = (float)e.pos.x; h.py = (float)e.pos.y; h.pz = (float)e.pos.z;
        return h;
    }
    h.t = t;
    h.px = fma_(t, r.dx, r.ox); h.py = fma_(t, r.dy, r.oy); h.pz = fma_(t, r.dz, r.oz);
    if (gprim < S.n_spheres) {
        const SphRow P = S.sph_by_id[gprim];
        float ir = rcp_(P.r);
        float nx = (h.px - P.cx) * ir, ny = (h.py - P.cy) * ir, nz = (h.pz - P.cz) * ir;
        float s = rsq_(fma_(nz, nz, fma_(ny, ny, nx * nx)));
        h.nx = nx * s; h.ny = ny * s; h.nz = nz * s;
    } else {
        const TriAux A = S.tri_aux[gprim - S.n_spheres];
        h.nx = A.nx; h.ny = A.ny; h.nz = A.nz;
    }
    return h;
}

}  // namespace rt
