/*
 * rt_oracle.c — TEST INFRASTRUCTURE, not product code.
 *
 * A plain-C, double-precision, scalar, brute-force restatement of the per-pixel render path of the
 * upstream Python path tracer (m-J-K-b/raytracing).  It exists to check the HIP kernels and to be
 * timed as the CPU baseline; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product library never links, loads or calls anything in this directory.
 *
 * Every function cites the upstream file:line it restates.  Arithmetic is written operation by
 * operation in the upstream's order (Python floats = IEEE doubles, no fused multiply-add), and the
 * file must be compiled with -ffp-contract=off so that the results are bit-identical to what the
 * Python code computes, up to the third-party Vector3 implementation's own rounding (see
 * oracle/README.md: "what pins this oracle").
 *
 * Random numbers: upstream draws from global MT19937 streams; this restatement (and the kernels)
 * instead take the draws of path (pixel, sample) from Philox4x32-10, one 128-bit block per path
 * segment, in upstream's draw ORDER (camera.py:46-47 first, then renderer.py:62,64 and
 * vec3.py:14-16 per bounce; see orc_uniform).  The golden
 * generator injects the very same stream into the upstream code, so both sides see one sequence.
 *
 * Optional "origin rule" (orc_set_origin_rule(1), OFF by default): the HIP kernels carry the
 * primitive a bounce ray starts on and never re-hit it at depth ~0 (the originating triangle is
 * skipped; of the originating sphere only the far root counts, and only for a ray heading into
 * it).  Upstream instead relies on `depth > 1e-10`, which on huge spheres lets round-off through
 * ("acne": ~2 % of the ground hits of random_spheres re-hit the ground at t ~ 1e-10).  With the
 * rule ON this file restates the kernels' semantics instead of upstream's, so that the two
 * effects (acne / float32 arithmetic) can be measured apart.
 *
 * Deliberate deviations from upstream (each counted in orc_counters):
 *   - sky lookup clamps asin's argument to [-1,1] and the texel index to the image; upstream raises
 *     (IndexError at u == 1.0, ValueError from int(nan)) and silently drops the tile
 *     (scenes/room.py:18-20, src/raytracing/util.py:25-28).
 */
#define _GNU_SOURCE
#include "../include/raytracing_amd.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vscale(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vprod(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); } /* vec3.py:25-26 */
static inline double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* Vector3.normalize: divide each component by sqrt(x*x+y*y+z*z) */
static inline v3 vnormalize(v3 a) {
    double l = sqrt(vdot(a, a));
    return V(a.x / l, a.y / l, a.z / l);
}
/* Vector3.reflect(n): n is re-normalised unless its squared length is exactly 1 */
static inline v3 vreflect(v3 d, v3 n) {
    double nl2 = vdot(n, n);
    if (nl2 != 1.0) {
        double nl = sqrt(nl2);
        n = V(n.x / nl, n.y / nl, n.z / nl);
    }
    double dp = vdot(d, n);
    return V(d.x - 2 * n.x * dp, d.y - 2 * n.y * dp, d.z - 2 * n.z * dp);
}

/* ---------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).
 * ------------------------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Uniform i (0..4) of segment c of path (pixel, sample): the i-th 24-bit field of the 128-bit
 * block Philox4x32-10(counter = (c,0,0,0), key = (pixel ^ seed_lo, sample ^ seed_hi)), taken from
 * the most significant end: x[31:8], x[7:0]:y[31:16], y[15:0]:z[31:24], z[23:0], w[31:8].
 * Segment 0 feeds the camera (depth-of-field jitter, camera.py:46-47); segment j+1 feeds the
 * shading decision after the j-th Scene.intersect call of the path (renderer.py:62,64 and
 * vec3.py:14-16 draw at most five numbers there), in upstream's call order. */
double orc_uniform(uint32_t pixel, uint32_t sample, uint64_t seed, uint32_t c, uint32_t i) {
    uint32_t ctr[4] = {c, 0, 0, 0};
    uint32_t key[2] = {pixel ^ (uint32_t)seed, sample ^ (uint32_t)(seed >> 32)};
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    uint32_t f;
    switch (i) {
    case 0: f = o[0] >> 8; break;
    case 1: f = ((o[0] & 0xFFu) << 16) | (o[1] >> 16); break;
    case 2: f = ((o[1] & 0xFFFFu) << 8) | (o[2] >> 24); break;
    case 3: f = o[2] & 0xFFFFFFu; break;
    default: f = o[3] >> 8; break;
    }
    return (double)f * (1.0 / 16777216.0);
}

typedef struct {
    uint32_t pixel, sample; uint64_t seed;
    uint32_t c, i;      /* current segment and the next field inside it */
    uint32_t total;     /* draws consumed by the path so far */
    const double *inject; int n_inject; /* optional explicit stream (tests), consumed sequentially */
} draw_stream;

static inline void begin_segment(draw_stream *s, uint32_t c) { s->c = c; s->i = 0; }

static inline double next_draw(draw_stream *s) {
    if (s->inject) {
        double v = (int)s->total < s->n_inject ? s->inject[s->total] : 0.5;
        s->total++;
        return v;
    }
    s->total++;
    return orc_uniform(s->pixel, s->sample, s->seed, s->c, s->i++);
}

/* ---------------------------------------------------------------------------------------------
 * Geometry
 * ------------------------------------------------------------------------------------------- */
typedef struct { int obj, prim, gprim; double t; v3 n, pos; } hit_rec;

typedef struct {
    uint64_t sky_clamped_index, sky_clamped_asin;
} orc_counters_t;
static orc_counters_t g_counters;
void orc_counters(uint64_t out[2]) { out[0] = g_counters.sky_clamped_index; out[1] = g_counters.sky_clamped_asin; }
void orc_counters_reset(void) { memset(&g_counters, 0, sizeof g_counters); }

/* Sphere.intersect (src/mesh/sphere.py:13-41) + quadratic_formula (src/core/util.py:29-32).
 * Writes up to two hits (near root first); returns their number. */
static int sphere_hits(const double *sp, v3 o, v3 d, double t_out[2], v3 n_out[2], v3 p_out[2], int root_out[2]) {
    v3 c = V(sp[0], sp[1], sp[2]);
    double radius = sp[3];
    v3 ro = vsub(o, c);
    double a = vdot(d, d);
    double b = 2 * vdot(ro, d);
    double cc = vdot(ro, ro) - pow(radius, 2.0); /* radius**2: float ** int is C pow() in CPython */
    double radicand = pow(b, 2.0) - 4 * a * cc; /* b**2 */
    if (radicand < 0) return 0;
    double sq = pow(pow(b, 2.0) - 4 * a * cc, 0.5); /* (...) ** 0.5 is C pow() in CPython, not sqrt() */
    double t1 = (-b - sq) / (2 * a), t2 = (-b + sq) / (2 * a);
    int n = 0;
    if (t1 > 0) {
        v3 p = vadd(o, vscale(d, t1)); /* Ray.at, src/core/ray.py:12-13 */
        t_out[n] = t1; p_out[n] = p; n_out[n] = vnormalize(vsub(p, c)); root_out[n] = 0; n++;
    }
    if (t2 > 0) {
        v3 p = vadd(o, vscale(d, t2));
        t_out[n] = t2; p_out[n] = p; n_out[n] = vnormalize(vsub(p, c)); root_out[n] = 1; n++;
    }
    return n;
}

/* Mesh.intersect_aabb (src/mesh/mesh.py:61-92) */
static int aabb_gate(const double *bb, v3 o, v3 d) {
    v3 inv = V(d.x != 0 ? 1 / d.x : 1e10, d.y != 0 ? 1 / d.y : 1e10, d.z != 0 ? 1 / d.z : 1e10);
    v3 tmin = V((bb[0] - o.x) * inv.x, (bb[1] - o.y) * inv.y, (bb[2] - o.z) * inv.z);
    v3 tmax = V((bb[3] - o.x) * inv.x, (bb[4] - o.y) * inv.y, (bb[5] - o.z) * inv.z);
    /* Python's min(a, b) returns a unless b < a; max(a, b) returns a unless b > a */
    v3 t1 = V(tmax.x < tmin.x ? tmax.x : tmin.x, tmax.y < tmin.y ? tmax.y : tmin.y, tmax.z < tmin.z ? tmax.z : tmin.z);
    v3 t2 = V(tmax.x > tmin.x ? tmax.x : tmin.x, tmax.y > tmin.y ? tmax.y : tmin.y, tmax.z > tmin.z ? tmax.z : tmin.z);
    double t_near = t1.x; if (t1.y > t_near) t_near = t1.y; if (t1.z > t_near) t_near = t1.z;
    double t_far = t2.x; if (t2.y < t_far) t_far = t2.y; if (t2.z < t_far) t_far = t2.z;
    double lo = t_near; if (0 > lo) lo = 0;
    return t_far >= lo;
}

/* Mesh.intersect_tri (src/mesh/mesh.py:109-143) */
static int tri_hit(const double *tv, v3 o, v3 d, double *t, v3 *n, v3 *p) {
    v3 pa = V(tv[0], tv[1], tv[2]), pb = V(tv[3], tv[4], tv[5]), pc = V(tv[6], tv[7], tv[8]);
    v3 ab = vsub(pb, pa), ac = vsub(pc, pa);
    v3 nrm = vcross(ab, ac);
    double det = -vdot(d, nrm);
    if (fabs(det) < 1e-6) return 0;
    double inv_det = 1 / det;
    v3 ao = vsub(o, pa);
    v3 dao = vcross(ao, d);
    double dst = vdot(nrm, ao) * inv_det;
    double u = vdot(ac, dao) * inv_det;
    double v = -vdot(ab, dao) * inv_det;
    double w = 1 - u - v;
    if (dst >= 0 && u >= 0 && v >= 0 && w >= 0) {
        *t = dst; *p = vadd(o, vscale(d, dst)); *n = vnormalize(nrm);
        return 1;
    }
    return 0;
}

static int g_origin_rule = 0;
void orc_set_origin_rule(int on) { g_origin_rule = on; }

/* Scene.intersect (src/raytracing/scene.py:23-27): all hits with depth > 1e-10, stable-sorted by
 * depth; only hits[0] is consumed (renderer.py:36), i.e. the first minimum in object order, then
 * per-object hit order.  Returns the number of hits that survive the depth filter.
 * origin = global id of the primitive the ray starts on (used only under the origin rule). */
static int scene_intersect(const rt_scene_desc *s, v3 o, v3 d, hit_rec *best, int origin) {
    if (!g_origin_rule) origin = -1;
    int n_hits = 0;
    best->obj = -1; best->prim = -1; best->gprim = -1; best->t = INFINITY;
    for (int oi = 0; oi < s->n_objects; ++oi) {
        if (s->obj_kind[oi] == RT_OBJ_SPHERE) {
            int si = s->obj_index[oi];
            double t[2]; v3 n[2], p[2]; int root[2];
            int k = sphere_hits(s->spheres + 4 * (size_t)si, o, d, t, n, p, root);
            for (int j = 0; j < k; ++j) {
                if (!(t[j] > 1e-10)) continue;
                if (si == origin) { /* far root only, and only when heading into the sphere */
                    const double *sp = s->spheres + 4 * (size_t)si;
                    v3 ro = vsub(o, V(sp[0], sp[1], sp[2]));
                    if (root[j] != 1 || !(2 * vdot(ro, d) < 0)) continue;
                }
                n_hits++;
                if (t[j] < best->t) {
                    best->t = t[j]; best->obj = oi; best->prim = root[j]; best->gprim = si;
                    best->n = n[j]; best->pos = p[j];
                }
            }
        } else {
            int mi = s->obj_index[oi];
            if (!aabb_gate(s->mesh_aabb + 6 * (size_t)mi, o, d)) continue;
            for (int ti = s->mesh_tri_begin[mi]; ti < s->mesh_tri_begin[mi + 1]; ++ti) {
                double t; v3 n, p;
                if (s->n_spheres + ti == origin) continue;
                if (!tri_hit(s->tri_verts + 9 * (size_t)ti, o, d, &t, &n, &p)) continue;
                if (!(t > 1e-10)) continue;
                n_hits++;
                if (t < best->t) {
                    best->t = t; best->obj = oi; best->prim = ti - s->mesh_tri_begin[mi];
                    best->gprim = s->n_spheres + ti; best->n = n; best->pos = p;
                }
            }
        }
    }
    return n_hits;
}

/* exported: closest hit of n rays.  gprim = sphere index, or n_spheres + global triangle index. */
void orc_intersect_batch(const rt_scene_desc *s, int64_t n, const double *org, const double *dir,
                         const int32_t *origin_prim, int32_t *obj, int32_t *prim, int32_t *gprim,
                         double *t, double *normal, double *pos, int32_t *n_hits) {
    /* rays are independent; threads only make big brute-force batches bearable */
#pragma omp parallel for schedule(dynamic, 16) if (n >= 256)
    for (int64_t i = 0; i < n; ++i) {
        hit_rec h;
        int k = scene_intersect(s, V(org[3 * i], org[3 * i + 1], org[3 * i + 2]),
                                V(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), &h, origin_prim ? origin_prim[i] : -1);
        obj[i] = h.obj; prim[i] = h.prim; if (gprim) gprim[i] = h.gprim;
        t[i] = h.obj >= 0 ? h.t : INFINITY;
        if (normal) { normal[3 * i] = h.obj >= 0 ? h.n.x : 0; normal[3 * i + 1] = h.obj >= 0 ? h.n.y : 0; normal[3 * i + 2] = h.obj >= 0 ? h.n.z : 0; }
        if (pos) { pos[3 * i] = h.obj >= 0 ? h.pos.x : 0; pos[3 * i + 1] = h.obj >= 0 ? h.pos.y : 0; pos[3 * i + 2] = h.obj >= 0 ? h.pos.z : 0; }
        if (n_hits) n_hits[i] = k;
    }
}

/* ---------------------------------------------------------------------------------------------
 * Shading helpers (src/raytracing/util.py)
 * ------------------------------------------------------------------------------------------- */
/* schlick_approximation, util.py:11-15 */
double orc_schlick(const double incident[3], const double normal[3], double ior) {
    double c = incident[0] * normal[0] + incident[1] * normal[1] + incident[2] * normal[2];
    double r0 = (1 - ior) / (1 + ior);
    r0 = r0 * r0;
    return r0 + (1 - r0) * pow(1 - c, 5);
}

/* refract, util.py:18-22 (result is NOT normalised) */
static v3 refract_dir(v3 incident, v3 normal, double eta) {
    double cos_theta = vdot(normal, vneg(incident));
    if (1 < cos_theta) cos_theta = 1; /* min(x, 1) */
    v3 perp = vscale(vadd(incident, vscale(normal, cos_theta)), eta);
    double par = -sqrt(fabs(1.0 - vdot(perp, perp)));
    return vadd(perp, vscale(normal, par));
}
void orc_refract(const double inc[3], const double nrm[3], double eta, double out[3]) {
    v3 r = refract_dir(V(inc[0], inc[1], inc[2]), V(nrm[0], nrm[1], nrm[2]), eta);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_reflect(const double d[3], const double n[3], double out[3]) {
    v3 r = vreflect(V(d[0], d[1], d[2]), V(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* vec_to_sky_coords, util.py:25-28 */
void orc_sky_coords(const double d[3], double uv[2]) {
    uv[0] = 0.5 + atan2(d[2], d[0]) / (2 * M_PI);
    uv[1] = 1 - (0.5 + asin(d[1]) / M_PI);
}

/* sample_sky_map closure, scenes/room.py:15-22 / scenes/random_spheres.py:13-20; black when the
 * scene has no sky image (Scene.get_environment default, scene.py:35-36). */
static v3 environment(const rt_scene_desc *s, v3 d) {
    if (!s->sky_rgb || s->sky_w <= 0 || s->sky_h <= 0) return V(0, 0, 0);
    double y = d.y;
    if (!(y >= -1.0 && y <= 1.0)) { /* upstream: asin -> nan -> ValueError */
        __atomic_fetch_add(&g_counters.sky_clamped_asin, 1, __ATOMIC_RELAXED);
        y = y > 1.0 ? 1.0 : -1.0;
    }
    double u = 0.5 + atan2(d.z, d.x) / (2 * M_PI);
    double v = 1 - (0.5 + asin(y) / M_PI);
    int px = (int)(u * s->sky_w), py = (int)(v * s->sky_h);
    if (px < 0 || px >= s->sky_w || py < 0 || py >= s->sky_h) { /* upstream: IndexError */
        __atomic_fetch_add(&g_counters.sky_clamped_index, 1, __ATOMIC_RELAXED);
        px = px < 0 ? 0 : (px >= s->sky_w ? s->sky_w - 1 : px);
        py = py < 0 ? 0 : (py >= s->sky_h ? s->sky_h - 1 : py);
    }
    const uint8_t *c = s->sky_rgb + 3 * ((size_t)py * s->sky_w + px);
    v3 col = V(c[0] / 255.0, c[1] / 255.0, c[2] / 255.0);
    if (s->sky_scale != 1.0) col = vscale(col, s->sky_scale);
    return col;
}
void orc_environment(const rt_scene_desc *s, const double d[3], double rgb[3]) {
    v3 c = environment(s, V(d[0], d[1], d[2]));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* ---------------------------------------------------------------------------------------------
 * Camera.get_ray (src/raytracing/camera.py:39-50)
 * ------------------------------------------------------------------------------------------- */
static void camera_ray(const rt_scene_desc *s, double u, double v, draw_stream *rng, v3 *o, v3 *d) {
    v3 right = V(s->cam_right[0], s->cam_right[1], s->cam_right[2]);
    v3 up = V(s->cam_up[0], s->cam_up[1], s->cam_up[2]);
    v3 fwd = V(s->cam_forward[0], s->cam_forward[1], s->cam_forward[2]);
    v3 pos = V(s->cam_pos[0], s->cam_pos[1], s->cam_pos[2]);
    v3 dir = vnormalize(vadd(vadd(vscale(right, u), vscale(up, v)), vscale(fwd, s->cam_d)));
    if (s->cam_dof_dist == 0 || s->cam_dof_strength == 0) { *o = pos; *d = dir; return; }
    v3 target = vadd(vscale(dir, s->cam_dof_dist), pos);
    double r1 = next_draw(rng), r2 = next_draw(rng);
    v3 rp = vadd(vadd(pos, vscale(right, (r1 * 2 - 1) * s->cam_dof_strength)),
                 vscale(up, (r2 * 2 - 1) * s->cam_dof_strength));
    *o = rp; *d = vnormalize(vsub(target, rp));
}
void orc_camera_ray(const rt_scene_desc *s, double u, double v, const double *draws, int n_draws,
                    double o[3], double d[3]) {
    draw_stream rng = {0, 0, 0, 0, 0, 0, draws, n_draws};
    v3 oo, dd;
    camera_ray(s, u, v, &rng, &oo, &dd);
    o[0] = oo.x; o[1] = oo.y; o[2] = oo.z; d[0] = dd.x; d[1] = dd.y; d[2] = dd.z;
}

/* pixel -> (u, v), src/raytracing/renderer.py:96-103 with ASPECT = height / width
 * (src/raytracing/render_settings.py:17) */
static inline void pixel_uv(int x, int y, int W, int H, double *u, double *v) {
    double aspect = (double)H / (double)W;
    *u = (x + 0.5) / W * 2 - 1;
    *v = (y + 0.5) / H * 2 * aspect - aspect;
}

/* ---------------------------------------------------------------------------------------------
 * Renderer.raytrace (src/raytracing/renderer.py:21-79), iteratively.  Upstream recurses with a
 * fresh ray_color = (1,1,1) per level and returns colour_k (x) child, so the radiance of a path is
 * c_1 (x) (c_2 (x) (... (x) E)); the products are applied back to front to keep upstream's
 * association.
 * Optional per-segment trace (for the ray-dump fixtures): seg_* arrays of capacity max_bounces+1.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    double *org, *dir; int32_t *inside, *obj, *prim; double *t; int n;
} seg_trace;

static v3 trace_path_from(const rt_scene_desc *s, v3 o, v3 d, int inside, int first_depth, int max_bounces,
                          draw_stream *rng, uint64_t *rays, seg_trace *tr) {
    v3 colors[64];
    int n_colors = 0;
    int origin = -1;
    v3 terminal = V(0, 0, 0);
    for (int depth = first_depth;; ++depth) {
        if (depth > max_bounces) { terminal = V(0, 0, 0); break; } /* renderer.py:29-30 */
        hit_rec h;
        (*rays)++;
        scene_intersect(s, o, d, &h, origin); /* renderer.py:32 */
        begin_segment(rng, (uint32_t)depth + 1);
        if (tr) {
            int i = tr->n++;
            tr->org[3 * i] = o.x; tr->org[3 * i + 1] = o.y; tr->org[3 * i + 2] = o.z;
            tr->dir[3 * i] = d.x; tr->dir[3 * i + 1] = d.y; tr->dir[3 * i + 2] = d.z;
            tr->inside[i] = inside; tr->obj[i] = h.obj; tr->prim[i] = h.prim; tr->t[i] = h.obj >= 0 ? h.t : INFINITY;
        }
        if (h.obj < 0) { terminal = environment(s, d); break; } /* renderer.py:33-34 */
        const double *m = s->materials + 7 * (size_t)h.obj;
        v3 color = V(m[0], m[1], m[2]);
        double emission = m[3], smoothness = m[4], transmittance = m[5], m_ior = m[6];
        double ior = 1 / m_ior; /* renderer.py:40 */
        v3 N = h.n;
        if (inside) { N = vneg(N); ior = m_ior; } /* renderer.py:41-43 */
        if (emission > 0) { terminal = vscale(color, emission); break; } /* renderer.py:45-47 */
        if (n_colors < 64) colors[n_colors++] = color; /* renderer.py:49 */
        v3 refl = vreflect(d, N); /* renderer.py:51 */
        v3 minus_d = vneg(d);
        double inc[3] = {minus_d.x, minus_d.y, minus_d.z}, nn[3] = {N.x, N.y, N.z};
        double fresnel = orc_schlick(inc, nn, m_ior); /* renderer.py:52-54 */
        double cos_theta = vdot(N, minus_d);
        if (!(cos_theta > 0.0)) cos_theta = 0.0; /* max(x, 0.0) */
        double one_m = 1.0 - cos_theta * cos_theta;
        double sin_theta = sqrt(one_m > 0.0 ? one_m : 0.0);
        int cannot_refract = ior * sin_theta > 1.0; /* renderer.py:56-58 */
        v3 nd;
        if (cannot_refract || next_draw(rng) * transmittance < fresnel) { /* renderer.py:60-63 */
            if (next_draw(rng) < smoothness) { /* renderer.py:64-65 */
                nd = refl;
            } else { /* random_hemisphere_sample, util.py:31-35 + Vec3.random_unit, vec3.py:11-17 */
                double rx = next_draw(rng) * 2 - 1, ry = next_draw(rng) * 2 - 1, rz = next_draw(rng) * 2 - 1;
                nd = vnormalize(V(rx, ry, rz));
                if (vdot(nd, N) < 0) nd = vneg(nd);
            }
        } else { /* renderer.py:68-70 */
            nd = refract_dir(d, N, ior);
            inside = !inside;
        }
        o = h.pos; d = nd; origin = h.gprim; /* renderer.py:73-76: new ray starts exactly at the hit position */
    }
    v3 L = terminal;
    for (int i = n_colors - 1; i >= 0; --i) L = vprod(colors[i], L); /* renderer.py:78 */
    return L;
}

static v3 trace_path(const rt_scene_desc *s, v3 o, v3 d, int max_bounces, draw_stream *rng,
                     uint64_t *rays, seg_trace *tr) {
    return trace_path_from(s, o, d, 0, 0, max_bounces, rng, rays, tr);
}

/* Renderer.raytrace(ray, depth=first_depth) (renderer.py:21-79) for n given rays; ray i draws from
 * the stream of (pixel = stream_begin + i, sample). */
void orc_raytrace_batch(const rt_scene_desc *s, int64_t n, const double *org, const double *dir,
                        const uint8_t *inside, int first_depth, int max_bounces, uint64_t seed,
                        uint32_t sample, uint32_t stream_begin, double *rgb) {
    for (int64_t i = 0; i < n; ++i) {
        draw_stream rng = {stream_begin + (uint32_t)i, sample, seed, 0, 0, 0, NULL, 0};
        uint64_t r = 0;
        v3 L = trace_path_from(s, V(org[3 * i], org[3 * i + 1], org[3 * i + 2]), V(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
                               inside ? inside[i] != 0 : 0, first_depth, max_bounces, &rng, &r, NULL);
        rgb[3 * i] = L.x; rgb[3 * i + 1] = L.y; rgb[3 * i + 2] = L.z;
    }
}

/* One path of pixel (x, y), sample index `sample`; optionally records its segments. */
void orc_trace_pixel(const rt_scene_desc *s, int W, int H, int x, int y, uint32_t sample,
                     int max_bounces, uint64_t seed, const double *inject, int n_inject,
                     double rgb[3], uint64_t *rays, int32_t *n_draws,
                     double *seg_org, double *seg_dir, int32_t *seg_inside, int32_t *seg_obj,
                     int32_t *seg_prim, double *seg_t, int32_t *n_seg) {
    double u, v;
    pixel_uv(x, y, W, H, &u, &v);
    draw_stream rng = {(uint32_t)(x * H + y), sample, seed, 0, 0, 0, inject, n_inject};
    v3 o, d;
    camera_ray(s, u, v, &rng, &o, &d);
    seg_trace tr = {seg_org, seg_dir, seg_inside, seg_obj, seg_prim, seg_t, 0};
    uint64_t r = 0;
    v3 L = trace_path(s, o, d, max_bounces, &rng, &r, seg_org ? &tr : NULL);
    rgb[0] = L.x; rgb[1] = L.y; rgb[2] = L.z;
    if (rays) *rays = r;
    if (n_draws) *n_draws = (int32_t)rng.total;
    if (n_seg) *n_seg = tr.n;
}

/* Renderer.render_area (renderer.py:90-106) for passes sample_begin .. sample_begin+sample_count-1,
 * adding into arr (W,H,3) doubles (RenderResult.add_to_px, render_result.py:24-27).
 * Parallel over x columns when built with OpenMP and nthreads != 1. */
void orc_render(const rt_scene_desc *s, int W, int H, int x0, int x1, int y0, int y1,
                int sample_begin, int sample_count, int max_bounces, uint64_t seed,
                double *arr, uint64_t *rays_out, int nthreads) {
    uint64_t total_rays = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : total_rays)
#endif
    for (int x = x0; x < x1; ++x) {
        for (int y = y0; y < y1; ++y) {
            for (int sidx = sample_begin; sidx < sample_begin + sample_count; ++sidx) {
                double u, v;
                pixel_uv(x, y, W, H, &u, &v);
                draw_stream rng = {(uint32_t)(x * H + y), (uint32_t)sidx, seed, 0, 0, 0, NULL, 0};
                v3 o, d;
                camera_ray(s, u, v, &rng, &o, &d);
                uint64_t r = 0;
                v3 L = trace_path(s, o, d, max_bounces, &rng, &r, NULL);
                double *px = arr + 3 * ((size_t)x * H + y);
                px[0] += L.x; px[1] += L.y; px[2] += L.z;
                total_rays += r;
            }
        }
    }
    if (rays_out) *rays_out = total_rays;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
