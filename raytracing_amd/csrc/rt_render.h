// The per-pixel render path as one persistent kernel: ray generation (camera.py:39-50), the
// bounce loop of Renderer.raytrace (renderer.py:21-79), environment lookup (scenes/room.py:15-22)
// and accumulation (render_result.py:24-27).
//
// Work decomposition.  A job is (pixel, chunk of consecutive samples).  Waves pull blocks of jobs
// from one global counter; inside a wave, a lane that finishes its job takes the next one of the
// wave's block (ballot + prefix count), so all 64 lanes stay busy in the brute-force intersection
// loops, whose cost per ray does not depend on the ray.  Each job owns one output slot, written
// with plain stores (no float atomics): the image is bit-reproducible for a given launch
// geometry, and a small reduction kernel folds the chunk slots into the accumulation buffer in
// chunk order.
#pragma once
#include <hip/hip_runtime.h>
#include "rt_filter.h"
#include "rt_scene.h"

namespace rt {

// ---------------------------------------------------------------------------------------------
// Random numbers: Philox4x32-10 (Salmon et al., SC'11), key = (pixel ^ seed_lo, sample ^ seed_hi),
// counter = (segment, 0, 0, 0).  One 128-bit block per path segment yields five 24-bit uniforms
// (fields taken from the most significant end); segment 0 feeds the camera, segment j+1 the
// shading step after the j-th closest-hit query, consumed in upstream's call order
// (camera.py:46-47; renderer.py:62,64; vec3.py:14-16 - at most five draws per step).
// ---------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

RT_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    U4 r; r.x = c0; r.y = c1; r.z = c2; r.w = c3;
    return r;
}

struct Draws { float f0, f1, f2, f3, f4; };

RT_DEV Draws segment_draws(uint32_t key0, uint32_t key1, uint32_t segment) {
    U4 o = philox4x32_10(segment, 0, 0, 0, key0, key1);
    const float k = 1.0f / 16777216.0f;
    Draws d;
    d.f0 = (float)(o.x >> 8) * k;
    d.f1 = (float)(((o.x & 0xFFu) << 16) | (o.y >> 16)) * k;
    d.f2 = (float)(((o.y & 0xFFFFu) << 8) | (o.z >> 24)) * k;
    d.f3 = (float)(o.z & 0xFFFFFFu) * k;
    d.f4 = (float)(o.w >> 8) * k;
    return d;
}

// ---------------------------------------------------------------------------------------------
// Environment: vec_to_sky_coords (util.py:25-28) + nearest texel (scenes/room.py:19-22).
// Deviation from upstream, shared with the CPU oracle: asin's argument and the texel index are
// clamped where upstream raises and silently loses the tile.
// ---------------------------------------------------------------------------------------------
// atan2(y, x) / (2 pi) in turns, in (-1/2, 1/2]: octant reduction + odd polynomial of degree 13 on
// [0, 1], fitted to atan(a) / (2 pi a) (max error 7e-8 turn in float32 = 1.3e-4 of a sky texel at
// 1907 texels per turn).
RT_DEV float atan2_turns(float y, float x) {
    const float ax = abs_(x), ay = abs_(y);
    const float mx = max_(ax, ay), mn = min_(ax, ay);
    const float a = mn * rcp_(max_(mx, 1e-30f));   // in [0, 1]
    const float s = a * a;
    // atan(a) / (2 pi) ~= a * P(a^2)
    float p = fma_(s, 0.0010841137263923883f, -0.005348223727196455f);
    p = fma_(s, p, 0.01267243456095457f);
    p = fma_(s, p, -0.021061478182673454f);
    p = fma_(s, p, 0.031525105237960815f);
    p = fma_(s, p, -0.0530262365937233f);
    p = fma_(s, p, 0.15915432572364807f);
    float r = a * p;
    r = ay > ax ? 0.25f - r : r;
    r = x < 0.0f ? 0.5f - r : r;
    return y < 0.0f ? -r : r;
}

// asin(x) / pi for |x| <= 1: pi/2 - sqrt(1 - |x|) * P(|x|) (Abramowitz & Stegun 4.4.46, |error| <= 2e-8 rad).
RT_DEV float asin_over_pi(float x) {
    const float a = abs_(x);
    float p = fma_(a, -0.0012624911f, 0.0066700901f);
    p = fma_(a, p, -0.0170881256f);
    p = fma_(a, p, 0.0308918810f);
    p = fma_(a, p, -0.0501743046f);
    p = fma_(a, p, 0.0889789874f);
    p = fma_(a, p, -0.2145988016f);
    p = fma_(a, p, 1.5707963050f);
    const float r = 0.5f - sqrt_(max_(1.0f - a, 0.0f)) * p * 0.3183098861837907f;
    return x < 0.0f ? -r : r;
}

RT_DEV void environment(SceneRef S, float dx, float dy, float dz, float &r, float &g, float &b) {
    if (S.sky == nullptr) { r = g = b = 0.0f; return; }
    float y = min_(max_(dy, -1.0f), 1.0f);
    float u = 0.5f + atan2_turns(dz, dx);
    float v = 1.0f - (0.5f + asin_over_pi(y));
    int px = (int)(u * (float)S.sky_w), py = (int)(v * (float)S.sky_h);
    px = px < 0 ? 0 : (px >= S.sky_w ? S.sky_w - 1 : px);
    py = py < 0 ? 0 : (py >= S.sky_h ? S.sky_h - 1 : py);
    uint32_t c = S.sky[(size_t)py * S.sky_w + px];
    const float k = S.sky_scale * (1.0f / 255.0f);
    r = (float)(c & 0xFFu) * k; g = (float)((c >> 8) & 0xFFu) * k; b = (float)((c >> 16) & 0xFFu) * k;
}

// 1 / sqrt(x) in double precision from the float32 hardware estimate and two Newton steps
// (relative error ~1e-16; avoids the long double-precision sqrt and divide sequences).
RT_DEV double rsqrt64(double x) {
    double s = (double)rsq_((float)x);
    s = s * (1.5 - 0.5 * x * s * s);
    s = s * (1.5 - 0.5 * x * s * s);
    return s;
}

// Ray generation: Camera.get_ray's pinhole part (camera.py:40) for every pixel of a W x H image
// (renderer.py:96-103), in double precision and in upstream's exact operation order - sqrt and
// divisions included, so the doubles are bit-identical to upstream's.  Writes, per pixel
// (x * H + y), the float32 direction (16 bytes) and the residue double - float32 as three doubles
// (32 bytes).  Runs once per (camera, W, H); the render kernel only loads from the table.
__global__ __launch_bounds__(256) void k_raygen(const DevScene S_arg, int W, int H, float *ray_dir, float *ray_res) {
    SceneRef S = *kernarg_scene();   // the scene argument, read in place (scalar loads)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const int x = i / H, y = i - x * H;
    const double *c = S.cold->cam64;
    const double aspect = (double)H / (double)W;
    const double u = (x + 0.5) / W * 2 - 1;
    const double v = (y + 0.5) / H * 2 * aspect - aspect;
    const D3 t = d3(c[3] * u + c[6] * v + c[9] * c[12], c[4] * u + c[7] * v + c[10] * c[12], c[5] * u + c[8] * v + c[11] * c[12]);
    const D3 d = dnormalize(t);
    const float fx = (float)d.x, fy = (float)d.y, fz = (float)d.z;
    f4 dir; dir.x = fx; dir.y = fy; dir.z = fz; dir.w = 0.0f;
    *(f4 *)(ray_dir + 4 * (size_t)i) = dir;
    double *res = (double *)ray_res + 4 * (size_t)i;   // (double - float32: exact in double precision)
    res[0] = d.x - (double)fx; res[1] = d.y - (double)fy; res[2] = d.z - (double)fz; res[3] = 0.0;
}

// Camera ray of one path.  Pinhole cameras: the table entry.  Depth of field (camera.py:43-49): the
// aperture offset and the re-aimed direction are worked out here from this path's two draws, in
// double precision with a Newton reciprocal square root (~1e-16 of upstream's doubles), and the
// float32 rounding of that ray is THE ray (no residue: the jitter decorrelates edge cases anyway).
RT_DEV void camera_ray(SceneRef S, const f4 q, const float *ray_res, uint32_t pixel, uint32_t key0,
                       uint32_t key1, float &ox, float &oy, float &oz, float &dx, float &dy, float &dz) {
    dx = q.x; dy = q.y; dz = q.z;
    ox = S.cam_pos[0]; oy = S.cam_pos[1]; oz = S.cam_pos[2];
    if (S.dof_on) {
        const double *c = S.cold->cam64;
        const double *e = (const double *)ray_res + 4 * (size_t)pixel;
        const double ddx = (double)q.x + e[0], ddy = (double)q.y + e[1], ddz = (double)q.z + e[2];
        const Draws dr = segment_draws(key0, key1, 0);
        const double j1 = ((double)dr.f0 * 2 - 1) * c[13], j2 = ((double)dr.f1 * 2 - 1) * c[13];
        const double gx = ddx * c[14] + c[0], gy = ddy * c[14] + c[1], gz = ddz * c[14] + c[2];
        const double oox = c[0] + c[3] * j1 + c[6] * j2, ooy = c[1] + c[4] * j1 + c[7] * j2, ooz = c[2] + c[5] * j1 + c[8] * j2;
        const double tx = gx - oox, ty = gy - ooy, tz = gz - ooz;
        const double s = rsqrt64(tx * tx + ty * ty + tz * tz);
        ox = (float)oox; oy = (float)ooy; oz = (float)ooz;
        dx = (float)(tx * s); dy = (float)(ty * s); dz = (float)(tz * s);
    }
}

// One bounce of Renderer.raytrace after a hit (renderer.py:39-70).  Returns false when the path
// ends at this surface (emissive).  Draw order: U1 unless total internal reflection, then U2, then
// the three cube-sample coordinates (vec3.py:14-16); `u` holds this segment's five uniforms and
// they are consumed positionally, so skipping U1 shifts the rest down by one.
RT_DEV bool shade_bounce(const MatRow &m, const Hit &h, const Draws &u, float &dx, float &dy, float &dz,
                         bool &inside, float &tr, float &tg, float &tb, float &er, float &eg, float &eb) {
    // (sign flips are exclusive-ors of the sign bit: full-rate instructions, where a select runs at half rate)
    const uint32_t flip = inside ? 0x80000000u : 0u;
    const float nx = u2f(f2u(h.nx) ^ flip), ny = u2f(f2u(h.ny) ^ flip), nz = u2f(f2u(h.nz) ^ flip);
    const float eta = inside ? m.ior : rcp_(m.ior);
    if (m.emission > 0.0f) {
        er = m.r * m.emission; eg = m.g * m.emission; eb = m.b * m.emission;
        return false;
    }
    tr *= m.r; tg *= m.g; tb *= m.b;
    float dn = fma_(dz, nz, fma_(dy, ny, dx * nx));   // d . N   (N unit)
    float cs = -dn;                                    // (-d) . N, not clamped (util.py:12)
    float om = 1.0f - cs, om2 = om * om;
    float fres = fma_(1.0f - m.r0, om2 * om2 * om, m.r0);
    float ct = max_(cs, 0.0f);
    float st = sqrt_(max_(0.0f, fma_(-ct, ct, 1.0f)));
    bool tir = eta * st > 1.0f;
    const float u2 = tir ? u.f0 : u.f1;
    const bool reflect_branch = tir || (u.f0 * m.transmittance < fres);
    if (reflect_branch) {
        if (u2 < m.smoothness) {
            float k = 2.0f * dn;
            dx = fma_(-k, nx, dx); dy = fma_(-k, ny, dy); dz = fma_(-k, nz, dz);
        } else {
            float rx = (tir ? u.f1 : u.f2) * 2.0f - 1.0f, ry = (tir ? u.f2 : u.f3) * 2.0f - 1.0f, rz = (tir ? u.f3 : u.f4) * 2.0f - 1.0f;
            float s = rsq_(fma_(rz, rz, fma_(ry, ry, rx * rx)));
            rx *= s; ry *= s; rz *= s;
            // into the normal's hemisphere (util.py:33-34: negated when the dot product is below zero; a
            // dot product of -0.0 leaves upstream's sample as it is, so the sign bit alone will not do)
            const float dn2 = fma_(rz, nz, fma_(ry, ny, rx * nx));
            const uint32_t sgn = dn2 < 0.0f ? 0x80000000u : 0u;
            dx = u2f(f2u(rx) ^ sgn); dy = u2f(f2u(ry) ^ sgn); dz = u2f(f2u(rz) ^ sgn);
        }
    } else {   // refract (util.py:18-22): result is not normalised
        float cr = min_(cs, 1.0f);
        float qx = eta * fma_(cr, nx, dx), qy = eta * fma_(cr, ny, dy), qz = eta * fma_(cr, nz, dz);
        float par = -sqrt_(abs_(1.0f - fma_(qz, qz, fma_(qy, qy, qx * qx))));
        dx = fma_(par, nx, qx); dy = fma_(par, ny, qy); dz = fma_(par, nz, qz);
        inside = !inside;
    }
    return true;
}

// Per-path state carried between segments.
struct Path {
    float ox, oy, oz, dx, dy, dz;   // current ray
    float tr, tg, tb;               // product of surface colours so far (renderer.py:49)
    uint32_t key0, key1;            // Philox key of this (pixel, sample)
    int32_t depth;                  // closest-hit queries done so far
    int32_t origin;                 // primitive the current ray starts on, -1 = camera
    uint32_t tag;                   // that primitive's TriAux::tag (0: none)
    float own_ratio;                // Hit::own_ratio of the hit the current ray leaves from
    bool inside;                    // Ray.inside (ray.py:10), toggled by refraction
    uint32_t pixel;                 // x * H + y: Philox key and row of the camera-ray table
    float rox, roy, roz;            // residue of the current ray's origin (RayResidue): non-zero only behind a hit the exact code produced
};

// `q`: the pixel's entry of the camera-ray table (float32 direction)
RT_DEV void path_begin(SceneRef S, const f4 q, const float *ray_res, Path &p, uint32_t pixel,
                       uint32_t sample, uint32_t seed_lo, uint32_t seed_hi) {
    p.key0 = pixel ^ seed_lo; p.key1 = sample ^ seed_hi;
    camera_ray(S, q, ray_res, pixel, p.key0, p.key1, p.ox, p.oy, p.oz, p.dx, p.dy, p.dz);
    p.pixel = pixel;
    p.tr = p.tg = p.tb = 1.0f; p.depth = 0; p.inside = false; p.origin = -1; p.tag = 0u; p.own_ratio = kInf;
    p.rox = p.roy = p.roz = 0.0f;
}
RT_DEV void path_begin(SceneRef S, const float *ray_dir, const float *ray_res, Path &p, uint32_t pixel,
                       uint32_t sample, uint32_t seed_lo, uint32_t seed_hi) {
    path_begin(S, *(const f4 *)(ray_dir + 4 * (size_t)pixel), ray_res, p, pixel, sample, seed_lo, seed_hi);
}


RT_DEV bool path_origin_rounded(const Path &p) { return p.rox != 0.0f || p.roy != 0.0f || p.roz != 0.0f; }

RT_DEV RayResidue path_residue(SceneRef S, const float *ray_res, const Path &p) {
    // a pinhole camera ray is the rounding of a double-precision ray whose residue is in the table
    RayResidue rr;
    rr.table = ray_res; rr.pixel = p.pixel; rr.on = (p.depth == 0) && !S.dof_on;
    rr.ox = p.rox; rr.oy = p.roy; rr.oz = p.roz; rr.o_on = path_origin_rounded(p);
    return rr;
}

// What follows a closest-hit answer: environment or one bounce.  Returns true while the path goes
// on; when it returns false (er, eg, eb) is the terminal radiance to be multiplied by (tr, tg, tb).
// `sky_count` is the calling lane's own count of environment lookups: one add inside the branch that
// runs anyway, instead of a flag, a compare and a ballot per trip; the kernel sums the lanes at its end.
RT_DEV bool path_shade(SceneRef S, Path &p, int max_bounces, const Hit &h, float &er, float &eg, float &eb, uint32_t &sky_count) {
    er = eg = eb = 0.0f;
    PROF(kProfRecord);
    if (h.obj < 0) {
        sky_count += 1u;
        environment(S, p.dx, p.dy, p.dz, er, eg, eb);
        return false;
    }
    PROF(kProfSky);
    const MatRow m = S.materials[h.obj];
    const Draws u = segment_draws(p.key0, p.key1, (uint32_t)p.depth + 1u);
    PROF(kProfDraws);
    bool alive = shade_bounce(m, h, u, p.dx, p.dy, p.dz, p.inside, p.tr, p.tg, p.tb, er, eg, eb);
    p.ox = h.px; p.oy = h.py; p.oz = h.pz; p.origin = h.gprim; p.tag = h.tag; p.own_ratio = h.own_ratio;
    p.rox = h.rpx; p.roy = h.rpy; p.roz = h.rpz;
    p.depth++;
    if (alive && p.depth > max_bounces) { alive = false; er = eg = eb = 0.0f; }   // renderer.py:29-30
    return alive;
}

// One segment: closest hit, then environment or one bounce.
// (DEFER: see hit_finish; a query handed back unanswered ends the path HERE with no radiance, h.how says so.)
template <int TRI, bool HAS_SPH, bool STRICT = false, bool DEFER = false>
RT_DEV bool path_step(SceneRef S, const float *ray_res, Path &p, int max_bounces, float &er, float &eg, float &eb, Hit &h,
                      uint32_t &sky_count) {
    Ray ray = make_ray(S, p.ox, p.oy, p.oz, p.dx, p.dy, p.dz, p.origin, p.depth == 0, p.tag, p.own_ratio, path_origin_rounded(p));
    PROF(kProfRaySetup);
    h = closest_hit<TRI, HAS_SPH, STRICT, DEFER>(S, ray, nullptr, path_residue(S, ray_res, p));
    if (DEFER && h.how == kHowDeferred) { er = eg = eb = 0.0f; return false; }
    const bool alive = path_shade(S, p, max_bounces, h, er, eg, eb, sky_count);
    PROF(kProfBounce);
    return alive;
}

RT_DEV uint32_t fast_div(uint32_t n, const FastDiv &f) {
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}


// The job-bookkeeping members of RenderArgs (rectangle, sample range, divisors, output pointers) are
// needed a few times per JOB, not per path segment.  Read as ordinary by-value kernel arguments they
// would sit in ~30 scalar registers for the whole launch, and the compiler pays for that by parking
// other scalars in vector-register lanes inside the hot loop (v_readlane / v_writelane are vector
// instructions, which is what this kernel is short of).  They are read where they are used instead,
// straight from the kernel-argument block (constant memory, scalar loads): the pointer is made
// opaque once per trip of the main loop so the loads cannot be hoisted back out of it.
typedef const RenderArgs __attribute__((address_space(4))) *KernargRenderArgs;
RT_DEV KernargRenderArgs kernarg_render_args() {
    typedef const char __attribute__((address_space(4))) *cbyte;
    return (KernargRenderArgs)((cbyte)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(DevScene) + 7) & ~(size_t)7));
}

#ifdef RT_WAVE_END   // diagnostic builds: per wave of the lean kernel, when it started, ran dry and ended (100 MHz clock), trips and busy lanes after running dry
__device__ unsigned long long g_wave_log[6 * 8192];
#endif
#ifndef RT_JOB_BLOCK
#define RT_JOB_BLOCK 64
#endif
// Jobs a wave takes from the global counter at a time.  When the counter runs out, a wave that has just taken a window
// still holds that much work while its neighbours go idle: measured with windows of 256 / 128 / 64 / 32 jobs, 1.18 M
// triangles 32 spp 30.2 / 28.6 / 28.6 / 28.7 ms, spheres scene 32 spp 9.55 / 9.02 / 8.91 / 9.30 ms, room 64 spp 12.85 /
// 12.75 / 12.80 / 13.04 ms.
constexpr int kJobBlock = RT_JOB_BLOCK;   // jobs a wave takes from the global counter at a time
constexpr int kWalkSlice = 4;   // wide nodes (four boxes each) a lane visits per trip of the render loop (hierarchy scenes)

// STRICT: a float32 hit depth is accepted only under the closest-hit entry's error bound (9.5e-6
// relative) instead of the render's looser one (see kRelAcceptRender); hit identity is the same.
//
// LEAN: the kernel holds none of the double-precision settling code (hit_finish with DEFER).  A path
// one of whose queries needs that code is SET ASIDE: the lane appends (pixel, sample, job) to
// RenderArgs::aside, counts the sample as done with no radiance, and goes on.  The FULL kernel is then
// launched over that list (RenderArgs::path_mode: a job is one record = one path, which goes on from the
// state the record holds - the same ray, the same random stream - and its radiance goes into the record), and
// k_reduce_chunks adds the records to their pixels.  When the list is full the lane gives its whole
// JOB up instead (RenderArgs::dirty_list),
// and a launch of the full kernel over that list (RenderArgs::job_list) renders those jobs again.  Every job's sum goes to its own slot of RenderArgs::partial in double precision: the
// sum of a few hundred float32 terms is then (all but) exact, so a pixel's value does not depend on
// which of its samples went which way.
#ifndef RT_LEAN_WAVES
#define RT_LEAN_WAVES 5   // waves per SIMD the lean kernel is compiled for (96 vector registers; room, 64 spp: 11.5 ms against 12.9 for the one-kernel form at four waves of 128)
#endif
#ifndef RT_BVH_FOUR_WAVES
#define RT_BVH_FOUR_WAVES (-1)   // set to kTriBvh: the lean kernel of hierarchy scenes is compiled for four waves per SIMD too
#endif
template <int TRI, bool HAS_SPH, bool STRICT, bool LEAN>
__global__ __launch_bounds__(256, (LEAN && TRI != RT_BVH_FOUR_WAVES) ? RT_LEAN_WAVES : 4) void k_render(const DevScene S_arg, const RenderArgs A) {
    SceneRef S = *kernarg_scene();   // the scene argument, read in place (scalar loads)
    // wave-uniform job window
    uint32_t win_next = 0, win_end = 0;
    bool drained = false;
    // per-lane job state
    bool have_job = false, in_path = false;
    uint32_t slot = 0;
    int s_cur = 0, s_end = 0;
    double ar = 0, ag = 0, ab = 0;
    uint32_t job_rays0 = 0, job_sky0 = 0;   // LEAN: this lane's counters when its job began (a job given up is not counted)
    int32_t job_last_rec = -1;              // LEAN: the record of the path this lane's job set aside last (-1: none yet)
    Path p;
    p.ox = p.oy = p.oz = 0; p.dx = p.dy = 0; p.dz = 1; p.tr = p.tg = p.tb = 1; p.key0 = p.key1 = 0;
    p.depth = 0; p.origin = -1; p.tag = 0u; p.own_ratio = kInf; p.inside = false; p.pixel = 0; p.rox = p.roy = p.roz = 0.0f;
    // Statistics are kept per WAVE in scalar registers (lane counts of ballots taken where the wave
    // is converged), not per lane: six vector registers less for the whole life of the kernel.
    uint32_t w_exact = 0, w_refine = 0;
    uint32_t l_rays = 0, l_sky = 0;   // this lane's own counts
    BvhCounters bc = bvh_counters();   // hierarchy only: per-lane test counts, reduced at the end
    // Every sample of a job starts with the job's pixel's camera-ray table entry: fetched once per job into the
    // lane's LDS slot, read from there at every path start (an LDS read instead of a trip to the vector cache:
    // room -1 %).  Linear-table scenes with triangles only: the spheres kernel measures the same either way, and
    // the hierarchy kernel's LDS is spoken for (4 KB more cost it a workgroup per CU: +11 %).
    constexpr bool kJobRayLds = TRI == kTriBrute;
    __shared__ f4 job_ray[kJobRayLds ? 256 : 1];
    // ... and the entries of a whole job window are fetched when the wave takes the window, by all 64 lanes at once (one
    // wait for memory per 64 jobs; fetched at the hand-out, every hand-out - a few lanes' worth of jobs - waited for
    // its own cold lines while the whole wave stood by)
    __shared__ f4 win_ray[kJobRayLds ? 256 : 1];
    uint32_t win_base = 0;
    bool win_staged = false;
    // hierarchy only: a closest-hit query in progress (walked a slice per trip of the main loop)
    typename FirstPass<TRI>::type first;
    first.init(S);
    int walk_node = -1;
    bool walking = false;
#ifdef RT_SECTION_TIMERS
    if (lane_id() == 0) {
        for (int k = 0; k < kProfCount; ++k) g_prof_acc[threadIdx.x >> 6][k] = 0;
        g_prof_last[threadIdx.x >> 6] = (unsigned)__builtin_amdgcn_s_memtime();
    }
#endif
#ifdef RT_WAVE_END   // diagnostic builds: when the waves of a launch start, run dry and end (100 MHz clock), in the reason slots 8-13
    const unsigned long long wave_t0 = wall_clock64();
    unsigned long long wave_dry = 0;
    unsigned trips_dry = 0, lanes_at_dry = 0;
#endif
    KernargRenderArgs J = kernarg_render_args();
    ScenePtr Sp = kernarg_scene();
    // (spheres-only kernels measured 2 % faster with the plain by-value arguments)
#define JOB(member) (TRI != kTriNone ? J->member : A.member)

    for (;;) {
        asm volatile("" : "+s"(J));
        if (TRI != kTriNone) asm volatile("" : "+s"(Sp));   // (spheres-only kernels measured faster without)
        SceneRef S = *Sp;   // (shadows the launch-wide binding: this trip's reads start here)
        // ---- 1. hand jobs to idle lanes ------------------------------------------------------
        unsigned long long idle = __ballot(!have_job);
        // The hand-out code is ~50 vector instructions that the whole wave pays for; a lane finishes
        // its job every trip or two.  Idle lanes therefore wait until a few of them can be served
        // at once (measured: -1 % kernel time for the linear-table scenes at four, neutral elsewhere).
        {
            const int n_idle = __builtin_popcountll(idle);
            if (n_idle < (TRI == kTriBrute ? 4 : 2) && n_idle != 64) idle = 0;
            // Path mode: a wave takes 64 records and sees them through TOGETHER before it takes more.  Every record
            // begins with the query that needs the double-precision code - thousands of instructions, worth running
            // once for 64 lanes, not once per refilled lane (measured: 2,200 instructions per path at 7 active lanes
            // with lane-by-lane refill; refilling once 16 / 32 / 48 lanes are idle instead of all 64: room 256 spp 46.75 /
            // 46.82 / 46.80 against 46.62 ms, 1.18 M triangles 32 spp 28.8 / 28.75 / 28.65 against 28.25).
            if (!LEAN && JOB(path_mode) && n_idle != 64) idle = 0;
        }
        while (idle != 0 && !(drained && win_next == win_end)) {
            if (win_next == win_end) {
                uint32_t base = 0;
                // (a job list is short - thousands of jobs, not millions: windows of one job per lane spread it over the chip)
                uint32_t n_jobs = JOB(job_list_count) != nullptr ? *JOB(job_list_count) : JOB(total_jobs);
                if (!LEAN && JOB(path_mode) && n_jobs > JOB(aside_cap)) n_jobs = JOB(aside_cap);
                uint32_t window = JOB(job_list_count) != nullptr ? 64u : (uint32_t)kJobBlock;
                // Near the end of the range a wave takes half a window: what it still holds when the counter runs dry is what
                // the rest of the chip waits for.  (win_end: where this wave's last window ended; the last two windows'
                // worth per resident wave.  Measured, room: 256 spp 46.78 -> 46.53 ms, 64 spp 12.68 -> 12.64, 32 spp 6.74
                // either way; a quarter window instead: 256 spp 46.67; windows of 16 or 8 at 32 spp: slower - every hand-out then waits
                // for the counter.)
                if (LEAN && win_end + gridDim.x * (4u * 2u * (uint32_t)kJobBlock) >= n_jobs) window = (uint32_t)kJobBlock / 2u;
                if (lane_id() == 0) base = atomicAdd(JOB(job_counter), window);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= n_jobs) {
#ifdef RT_WAVE_END
                    if (!drained) { wave_dry = wall_clock64(); lanes_at_dry = (unsigned)__builtin_popcountll(__ballot(have_job)); }
#endif
                    drained = true; break;
                }
                win_next = base;
                win_end = base + window < n_jobs ? base + window : n_jobs;
                if (kJobRayLds) {
                    win_base = base;
                    win_staged = kJobBlock <= 64 && JOB(job_list) == nullptr && !(!LEAN && JOB(path_mode));   // (a lane per job of the window)
                    if (win_staged) {
                        const uint32_t wj = base + lane_id();
                        if (wj < win_end) {
                            const FastDiv div_ch = {JOB(div_chunks.d), JOB(div_chunks.m), JOB(div_chunks.sh1), JOB(div_chunks.sh2)};
                            const FastDiv div_h = {JOB(div_rect_h.d), JOB(div_rect_h.m), JOB(div_rect_h.sh1), JOB(div_rect_h.sh2)};
                            const uint32_t wpl = fast_div(wj, div_ch), wcol = fast_div(wpl, div_h);
                            const int wpx = JOB(x0) + (int)wcol, wpy = JOB(y0) + (int)(wpl - wcol * (uint32_t)JOB(rect_h));
                            win_ray[threadIdx.x] = *(const f4 *)(A.ray_dir + 4 * (size_t)(uint32_t)(wpx * JOB(height) + wpy));
                        }
                        __builtin_amdgcn_wave_barrier();   // (other lanes of this wave read these entries below: LDS serves a wave's accesses in order)
                    }
                }
            }
            uint32_t avail = win_end - win_next;
            uint32_t r = rank_in(idle);
            if (!have_job && r < avail) {
                uint32_t job = win_next + r;
                if (!LEAN && JOB(path_mode)) {
                    // one record = one path: it goes on from the query the lean kernel handed back, result into the record
                    const AsideRec rec = JOB(aside)[job];
                    slot = job; s_cur = (int)rec.sample; s_end = s_cur + 1;
                    p.pixel = rec.pixel; p.key0 = rec.pixel ^ A.seed_lo; p.key1 = rec.sample ^ A.seed_hi;
                    p.ox = rec.ox; p.oy = rec.oy; p.oz = rec.oz; p.dx = rec.dx; p.dy = rec.dy; p.dz = rec.dz;
                    p.tr = rec.tr; p.tg = rec.tg; p.tb = rec.tb; p.depth = (int32_t)rec.depth; p.origin = rec.origin;
                    p.tag = rec.tag; p.own_ratio = rec.own_ratio; p.inside = rec.inside != 0u;
                    p.rox = p.roy = p.roz = 0.0f;   // (a record is written by the lean kernel, none of whose hits comes from the exact code)
                    ar = ag = ab = 0.0;
                    in_path = true; walking = false;
                    if constexpr (TRI == kTriBvh) {
                        // the query that was handed back comes with its first pass done (the lean kernel walked it): it is
                        // closed - judged, settled - in this lane's next trip without another walk
                        walking = true; walk_node = -1;
                        first.init(S);
                        first.lo0 = u2f(rec.pad[0]); first.lo1 = u2f(rec.pad[1]); first.id0 = (int32_t)rec.pad[2];
                    }
                    have_job = !JOB(dirty)[rec.slot];   // (its job was given up later on: the job list covers it)
                } else {
                // Jobs (and their output slots) are numbered pixel by pixel: the chunks of one pixel go to neighbouring
                // lanes, which share the pixel's camera-ray table entry, start their first segment alike and write
                // neighbouring slots (measured against chunk-by-chunk numbering: lean kernel -2.7 % on the room,
                // -5.5 % at 1.18 M triangles).
                if (JOB(job_list) != nullptr) job = JOB(job_list)[job];
                const FastDiv div_ch = {JOB(div_chunks.d), JOB(div_chunks.m), JOB(div_chunks.sh1), JOB(div_chunks.sh2)};
                const FastDiv div_h = {JOB(div_rect_h.d), JOB(div_rect_h.m), JOB(div_rect_h.sh1), JOB(div_rect_h.sh2)};
                const uint32_t pl = fast_div(job, div_ch), chunk = job - pl * (uint32_t)JOB(n_chunks);
                const uint32_t col = fast_div(pl, div_h);
                const int px = JOB(x0) + (int)col, py = JOB(y0) + (int)(pl - col * (uint32_t)JOB(rect_h));
                p.pixel = (uint32_t)(px * JOB(height) + py);
                slot = job;
                const int chunk_samples = JOB(chunk_samples), sample_begin = JOB(sample_begin);
                s_cur = sample_begin + (int)chunk * chunk_samples;
                int lim = sample_begin + JOB(sample_count);
                s_end = s_cur + chunk_samples < lim ? s_cur + chunk_samples : lim;
                ar = ag = ab = 0.0;
                job_rays0 = l_rays; job_sky0 = l_sky; job_last_rec = -1;
                have_job = true; in_path = false;
                if (kJobRayLds) {
                    if (win_staged) job_ray[threadIdx.x] = win_ray[(threadIdx.x & ~63u) + ((slot - win_base) & 63u)];
                    else job_ray[threadIdx.x] = *(const f4 *)(A.ray_dir + 4 * (size_t)p.pixel);
                }
                }
            }
            uint32_t want = (uint32_t)__builtin_popcountll(idle);
            win_next += want < avail ? want : avail;
            idle = __ballot(!have_job);
        }
        if (__ballot(have_job) == 0) break;
#ifdef RT_WAVE_END
        if (drained) ++trips_dry;
#endif
        PROF(kProfHandout);
        PROF_LANES(kProfLanesJob, have_job);

        // Statistics: every lane counts its own closest-hit queries and environment lookups (one add
        // each, where the code runs anyway; summed over the wave at the end of the kernel); the two
        // rare counters (double-precision work) share one test per trip; paths are not counted at all
        // (a launch that runs to its end traces rectangle x samples of them, which the host knows).
        int rare = 0;
        if (have_job) {
            // ---- 2. start the next sample of this lane's pixel --------------------------------
            if (!in_path) {
                // (path mode never comes here: a record is a path under way)
                if (kJobRayLds) path_begin(S, job_ray[threadIdx.x], A.ray_res, p, p.pixel, (uint32_t)s_cur, A.seed_lo, A.seed_hi);
                else path_begin(S, A.ray_dir, A.ray_res, p, p.pixel, (uint32_t)s_cur, A.seed_lo, A.seed_hi);
                in_path = true;
            }
            PROF(kProfBegin);
            // ---- 3. one path segment ----------------------------------------------------------
            // (the path's Philox key is worked out again every trip - two exclusive-ors - from the pixel and the sample
            // index, which live across trips anyway: two vector registers less over the loop's back edge)
            p.key0 = p.pixel ^ A.seed_lo; p.key1 = (uint32_t)s_cur ^ A.seed_hi;
            float er = 0.0f, eg = 0.0f, eb = 0.0f;
            Hit h;
            bool alive = true;
            if (TRI == kTriBvh) {
                // resumable query: begin it if none is open, walk one slice, finish it when the walk is through
                const Ray ray = make_ray(S, p.ox, p.oy, p.oz, p.dx, p.dy, p.dz, p.origin, p.depth == 0, p.tag, p.own_ratio, path_origin_rounded(p));
                const RayResidue rr = path_residue(S, A.ray_res, p);
                PROF(kProfRaySetup);
                PROF_LANES(kProfLanesStart, !walking);
                if (!walking) {
                    walking = true;
                    walk_node = -1;
                    first.init(S);
                    if (ray_route<TRI>(ray) == kRouteFilter) { hit_begin<TRI, HAS_SPH>(S, ray, rr, first); walk_node = 0; }
                }
#ifdef RT_TUNE
                const int walk_slice = (!LEAN && JOB(path_mode)) ? 4 * A.tune[0] : A.tune[0], more_below = A.tune[1], more_times = A.tune[2], sub_slice = A.tune[3];
#else
                // (path mode: every lane holds a path that went deep or grazed something - longer slices, fewer trips)
                const int walk_slice = (!LEAN && JOB(path_mode)) ? 4 * kWalkSlice : kWalkSlice;
                constexpr int more_below = 32, more_times = 2, sub_slice = kSubSlice;
#endif
                if (walk_node >= 0) search_bvh(S, ray, first, bc, walk_node, walk_slice, sub_slice);
                // When only a few lanes are through, closing their queries now would run ~1,000
                // instructions for them alone: the wave walks another slice first, up to twice, while fewer than half
                // its lanes are through.  (Measured at 1.18 M triangles, kernel time for 32 samples per pixel: no
                // extra slice 50.7 ms; one while fewer than 16 lanes are through 49.0; than 32: 48.5; two while fewer
                // than 32: 47.4; than 48: 50.0; walking until every lane is through: 80.8.)
                if (!LEAN && JOB(path_mode)) {
                    // (path mode: all lanes close their queries in the same trip, see the hand-out above.  Measured at
                    // 1.18 M triangles, 780 k paths set aside: 6.4 ms; going on once three lanes in four are through:
                    // 6.5; slices and lane-by-lane refill as in the lean kernel: 7.4)
                    while (__ballot(walk_node >= 0) != 0) {
                        if (walk_node >= 0) search_bvh(S, ray, first, bc, walk_node, walk_slice, sub_slice);
                    }
                } else
                for (int k = 0; k < more_times && __builtin_popcountll(__ballot(walk_node < 0)) < more_below; ++k) {
                    if (walk_node >= 0) search_bvh(S, ray, first, bc, walk_node, walk_slice, sub_slice);
                }
                PROF_LANES(kProfLanesEnd, walk_node < 0);
                if (walk_node < 0) {
                    walking = false;
                    BvhCounters second_pass = bc; second_pass.nodes = 0; second_pass.tris = 0;
                    h = hit_finish<TRI, HAS_SPH, STRICT, LEAN>(S, ray, rr, first, second_pass, nullptr);
                    bc.nodes += h.bvh_nodes; bc.tris += h.bvh_tris;
                    if (LEAN && h.how == kHowDeferred) alive = false;
                    else alive = path_shade(S, p, A.max_bounces, h, er, eg, eb, l_sky);
                    PROF(kProfBounce);
                    l_rays += 1u;
                    rare = (h.how == 2 ? 1 : 0) | (h.refined ? 2 : 0) | (LEAN && h.how == kHowDeferred ? 4 : 0);
#ifdef RT_ASIDE_REASONS
                    if (LEAN && h.how == kHowDeferred) rare |= (h.prim & 7) << 4;
#endif
                }
            } else {
                l_rays += 1u;
                alive = path_step<TRI, HAS_SPH, STRICT, LEAN>(S, A.ray_res, p, A.max_bounces, er, eg, eb, h, l_sky);
                rare = (h.how == 2 ? 1 : 0) | (h.refined ? 2 : 0) | (LEAN && h.how == kHowDeferred ? 4 : 0);
#ifdef RT_ASIDE_REASONS
                if (LEAN && h.how == kHowDeferred) rare |= (h.prim & 7) << 4;
#endif
            }
            if (!alive) {
                // (products in float32 as ever; the SUM in double precision: see the note on LEAN above)
                ar += (double)(p.tr * er); ag += (double)(p.tg * eg); ab += (double)(p.tb * eb);
                in_path = false;
                s_cur++;
                if (s_cur >= s_end) {
                    if (!LEAN && JOB(path_mode)) {
                        AsideRec *rec = JOB(aside) + slot;
                        rec->r = (float)ar; rec->g = (float)ag; rec->b = (float)ab;   // (one product each: exact)
                        l_rays -= 1u;                                                // (the query handed back was counted by the lean kernel)
                    } else {
                        double *dst = JOB(partial) + 3 * (size_t)slot;
                        dst[0] = ar; dst[1] = ag; dst[2] = ab;
                    }
                    have_job = false;
                }
            }
        }
        PROF(kProfAccum);
        if (__ballot(rare != 0) != 0) {
            w_exact += (uint32_t)__builtin_popcountll(__ballot((rare & 1) != 0));
            w_refine += (uint32_t)__builtin_popcountll(__ballot((rare & 2) != 0));
            if (LEAN) {
                // paths set aside this trip: one counter bump for the wave, one record per path.  (s_cur has moved
                // on: the path is sample s_cur - 1 of job `slot`; p.depth is still the segment that was handed back.)
                // The records of one job form a list in the order the job's lane set them aside - descending sample
                // index from RenderArgs::head - so the fold adds them in an order that does not depend on where in the
                // list array they landed.
                const bool want = (rare & 4) != 0;
#ifdef RT_ASIDE_REASONS   // diagnostic builds: a histogram of the reasons in the section-timer slots
                if (want && A.prof != nullptr) atomicAdd(A.prof + ((rare >> 4) & 7), 1ull);
#endif
                const unsigned long long dm = __ballot(want);
                if (dm != 0) {
                    uint32_t base = 0;
                    if (lane_id() == (uint32_t)__builtin_ctzll(dm)) base = atomicAdd(JOB(aside_count), (uint32_t)__builtin_popcountll(dm));
                    base = (uint32_t)__shfl((int)base, __builtin_ctzll(dm), 64);
                    if (want) {
                        const uint32_t idx = base + rank_in(dm);
                        if (idx < JOB(aside_cap)) {
                            AsideRec rec;
                            rec.pixel = p.pixel; rec.sample = (uint32_t)(s_cur - 1); rec.slot = slot; rec.next = job_last_rec;
                            rec.r = rec.g = rec.b = 0.0f; rec.depth = (uint32_t)p.depth;
                            rec.ox = p.ox; rec.oy = p.oy; rec.oz = p.oz; rec.dx = p.dx; rec.dy = p.dy; rec.dz = p.dz;
                            rec.tr = p.tr; rec.tg = p.tg; rec.tb = p.tb; rec.origin = p.origin; rec.tag = p.tag;
                            rec.own_ratio = p.own_ratio; rec.inside = p.inside ? 1u : 0u; rec.pad[0] = rec.pad[1] = rec.pad[2] = 0u;
                            if constexpr (TRI == kTriBvh) { rec.pad[0] = f2u(first.lo0); rec.pad[1] = f2u(first.lo1); rec.pad[2] = (uint32_t)first.id0; }
                            JOB(aside)[idx] = rec;
                            JOB(head)[slot] = (int32_t)idx;
                            job_last_rec = (int32_t)idx;
                        } else {
                            // no room in the list: this lane's whole job is rendered again by the full kernel; nothing of it counts here
                            JOB(dirty)[slot] = 1;
                            JOB(dirty_list)[atomicAdd(JOB(dirty_count), 1u)] = slot;
                            l_rays = job_rays0; l_sky = job_sky0;
                            have_job = false;
                        }
                    }
                }
            }
        }
        PROF(kProfStats);
    }
#ifdef RT_SECTION_TIMERS
    if (lane_id() == 0 && A.prof != nullptr)
        for (int k = 0; k < kProfCount; ++k) atomicAdd(A.prof + k, (unsigned long long)g_prof_acc[threadIdx.x >> 6][k]);
#endif
#ifdef RT_WAVE_END
    if (LEAN && lane_id() == 0) {
        unsigned long long *w = g_wave_log + 6 * (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
        w[0] = wave_t0; w[1] = wave_dry; w[2] = wall_clock64(); w[3] = trips_dry; w[4] = lanes_at_dry; w[5] = 1;
    }
#endif
    // ---- statistics: one atomic per counter per wave ------------------------------------------
    unsigned long long nodes = bc.nodes, tris = bc.tris, w_rays = l_rays, w_sky = l_sky;
    for (int off = 32; off > 0; off >>= 1) {
        w_rays += __shfl_down(w_rays, off, 64); w_sky += __shfl_down(w_sky, off, 64);
        if (TRI == kTriBvh) { nodes += __shfl_down(nodes, off, 64); tris += __shfl_down(tris, off, 64); }
    }
    if (lane_id() == 0) {
        unsigned long long v[kStatCount] = {0};
        v[kStatRays] = w_rays; v[kStatExact] = w_exact; v[kStatRefine] = w_refine;
        v[kStatSky] = w_sky; v[kStatShaded] = w_rays - w_sky;
        v[kStatSphTests] = w_rays * (uint32_t)(S.n_sph_rows + S.n_big_sph);
        // linear-table tests (dead padding rows are not counted) plus, with a hierarchy, the leaf tests
        v[kStatTriTests] = w_rays * (uint32_t)S.n_tri_rows + (TRI == kTriBvh ? tris : 0ull);
        v[kStatBvhNodes] = nodes;
#pragma unroll
        for (int i = 0; i < kStatCount; ++i)
            if (v[i]) atomicAdd(JOB(stats) + i, v[i]);
    }
}
#undef JOB

// Folds a render call into the accumulation buffer: per pixel of the rectangle the job sums of its chunks, in
// chunk order, each followed by the paths of that job that were set aside (its AsideRec list, latest sample first) - all in
// double precision, rounded to float32 once.  Records of a job that was given up and rendered again
// (RenderArgs::dirty) are skipped: that job's slot already holds all of its samples.
__global__ __launch_bounds__(256) void k_reduce_chunks(const RenderArgs A) {
    // (Staging the slots, and the list heads, through LDS with coalesced loads was tried in three forms: all slower -
    // 0.63 to 0.88 ms against 0.50 for this loop on the room's 27 M jobs.  What bounds the kernel is the chain of
    // dependent loads on the few pixels whose every path was set aside: up to 260 records, one after the other.)
    const uint32_t rect_px = (uint32_t)(A.rect_w * A.rect_h);
    uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x;
    if (pl >= rect_px) return;
    int px = A.x0 + (int)(pl / (uint32_t)A.rect_h), py = A.y0 + (int)(pl % (uint32_t)A.rect_h);
    double r = 0, g = 0, b = 0;
    for (int c = 0; c < A.n_chunks; ++c) {
        const double *src = A.partial + 3 * ((size_t)pl * A.n_chunks + c);
        r += src[0]; g += src[1]; b += src[2];
    }
    for (int c = 0; A.head != nullptr && c < A.n_chunks; ++c) {
        const size_t slot = (size_t)pl * A.n_chunks + c;
        const int first = A.head[slot];
        if (first < 0 || A.dirty[slot]) continue;
        for (int i = first; i >= 0; i = A.aside[i].next) { r += (double)A.aside[i].r; g += (double)A.aside[i].g; b += (double)A.aside[i].b; }
    }
    float *dst = A.accum + 3 * ((size_t)px * A.height + py);
    dst[0] += (float)r; dst[1] += (float)g; dst[2] += (float)b;
}

// The same fold with one thread per JOB: consecutive threads read consecutive slots, and the chains of set-aside
// records - up to a few hundred dependent loads on a pixel whose every path was set aside - are walked per job, a
// block's worth side by side, not one after the other by the pixel's one thread.  A block takes as many whole pixels
// as fit 256 threads; the pixel's first thread then adds the jobs' sums up in chunk order.  (For n_chunks <= 256.)
__global__ __launch_bounds__(256) void k_reduce_jobs(const RenderArgs A, const int px_per_block) {
    __shared__ double sums[256][6];
    const uint32_t rect_px = (uint32_t)(A.rect_w * A.rect_h), n = (uint32_t)A.n_chunks;
    const uint32_t q = threadIdx.x / n, c = threadIdx.x - q * n;           // pixel of the block, chunk
    const uint32_t pl = blockIdx.x * (uint32_t)px_per_block + q;
    const bool live = q < (uint32_t)px_per_block && pl < rect_px;
    double r = 0, g = 0, b = 0, rr = 0, rg = 0, rb = 0;
    if (live) {
        const size_t slot = (size_t)pl * n + c;
        const double *src = A.partial + 3 * slot;
        r = src[0]; g = src[1]; b = src[2];
        if (A.head != nullptr) {
            const int first = A.dirty[slot] ? -1 : A.head[slot];
            for (int i = first; i >= 0; i = A.aside[i].next) { rr += (double)A.aside[i].r; rg += (double)A.aside[i].g; rb += (double)A.aside[i].b; }
        }
    }
    double *mine = sums[threadIdx.x];
    mine[0] = r; mine[1] = g; mine[2] = b; mine[3] = rr; mine[4] = rg; mine[5] = rb;
    __syncthreads();
    if (live && c == 0) {
        double tr = 0, tg = 0, tb = 0;
        for (uint32_t k = 0; k < n; ++k) { const double *t = sums[threadIdx.x + k]; tr += t[0]; tg += t[1]; tb += t[2]; }
        for (uint32_t k = 0; k < n; ++k) { const double *t = sums[threadIdx.x + k]; tr += t[3]; tg += t[4]; tb += t[5]; }
        const int px = A.x0 + (int)(pl / (uint32_t)A.rect_h), py = A.y0 + (int)(pl % (uint32_t)A.rect_h);
        float *dst = A.accum + 3 * ((size_t)px * A.height + py);
        dst[0] += (float)tr; dst[1] += (float)tg; dst[2] += (float)tb;
    }
}

// Diagnostic / parity kernel: one thread traces one whole path and records every segment.
struct TraceArgs {
    int32_t n_paths, width, height, max_bounces, seg_cap;
    const float *ray_dir, *ray_res;
    uint32_t seed_lo, seed_hi;
    const int32_t *xys;          // [n_paths][3] x, y, sample
    float *rgb;                  // [n_paths][3] path radiance
    int32_t *n_seg;              // [n_paths]
    int32_t *seg_obj, *seg_prim, *seg_how;   // [n_paths][seg_cap]
    float *seg_t;                // [n_paths][seg_cap]
    float *seg_org, *seg_dir;    // [n_paths][seg_cap][3]
    float *seg_org_res;          // [n_paths][seg_cap][3] or null: residue of the origin (RayResidue)
};

template <int TRI, bool HAS_SPH, bool STRICT>
__global__ __launch_bounds__(256) void k_trace_paths(const DevScene S_arg, const TraceArgs A) {
    SceneRef S = *kernarg_scene();   // the scene argument, read in place (scalar loads)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_paths) return;
    const int x = A.xys[3 * i], y = A.xys[3 * i + 1], smp = A.xys[3 * i + 2];
    Path p;
    path_begin(S, A.ray_dir, A.ray_res, p, (uint32_t)(x * A.height + y), (uint32_t)smp, A.seed_lo, A.seed_hi);
    float er = 0, eg = 0, eb = 0;
    int n = 0;
    for (;;) {
        const size_t k = (size_t)i * A.seg_cap + n;
        if (n < A.seg_cap) {
            A.seg_org[3 * k] = p.ox; A.seg_org[3 * k + 1] = p.oy; A.seg_org[3 * k + 2] = p.oz;
            A.seg_dir[3 * k] = p.dx; A.seg_dir[3 * k + 1] = p.dy; A.seg_dir[3 * k + 2] = p.dz;
            if (A.seg_org_res) { A.seg_org_res[3 * k] = p.rox; A.seg_org_res[3 * k + 1] = p.roy; A.seg_org_res[3 * k + 2] = p.roz; }
        }
        Hit h;
        uint32_t sky_unused = 0;
        bool alive = path_step<TRI, HAS_SPH, STRICT>(S, A.ray_res, p, A.max_bounces, er, eg, eb, h, sky_unused);
        if (n < A.seg_cap) { A.seg_obj[k] = h.obj; A.seg_prim[k] = h.prim; A.seg_how[k] = h.how; A.seg_t[k] = h.t; }
        n++;
        if (!alive) break;
    }
    A.n_seg[i] = n;
    A.rgb[3 * i] = p.tr * er; A.rgb[3 * i + 1] = p.tg * eg; A.rgb[3 * i + 2] = p.tb * eb;
}

// Renderer.raytrace (renderer.py:21-79) for caller-supplied rays: one thread follows one path from
// its given first ray (origin, direction, Ray.inside, upstream's `depth` argument) to its end and
// writes the radiance.  The float32 rounding of the given ray is the ray that is traced.
struct RaytraceArgs {
    long long n;
    const double *org, *dir;     // [n][3]
    const uint8_t *inside;       // [n] or null
    int32_t first_depth, max_bounces;
    uint32_t seed_lo, seed_hi, sample;
    uint32_t stream_begin;       // ray i draws from the random stream of "pixel" stream_begin + i
    float *rgb;                  // [n][3]
    unsigned long long *stats;
};

template <int TRI, bool HAS_SPH>
__global__ __launch_bounds__(256) void k_raytrace(const DevScene S_arg, const RaytraceArgs A) {
    SceneRef S = *kernarg_scene();   // the scene argument, read in place (scalar loads)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long rays = 0, sky = 0, shaded = 0, paths = 0;
    if (i < A.n) {
        Path p;
        p.pixel = A.stream_begin + (uint32_t)i;
        p.key0 = p.pixel ^ A.seed_lo; p.key1 = A.sample ^ A.seed_hi;
        p.ox = (float)A.org[3 * i]; p.oy = (float)A.org[3 * i + 1]; p.oz = (float)A.org[3 * i + 2];
        p.dx = (float)A.dir[3 * i]; p.dy = (float)A.dir[3 * i + 1]; p.dz = (float)A.dir[3 * i + 2];
        p.tr = p.tg = p.tb = 1.0f; p.depth = A.first_depth; p.origin = -1; p.tag = 0u; p.own_ratio = kInf; p.rox = p.roy = p.roz = 0.0f;
        p.inside = A.inside ? A.inside[i] != 0 : false;
        float er = 0, eg = 0, eb = 0;
        paths = 1;
        if (p.depth <= A.max_bounces) {   // renderer.py:29-30
            for (;;) {
                const Ray ray = make_ray(S, p.ox, p.oy, p.oz, p.dx, p.dy, p.dz, p.origin, false, p.tag, p.own_ratio, path_origin_rounded(p));
                RayResidue rr = no_residue();
                rr.ox = p.rox; rr.oy = p.roy; rr.oz = p.roz; rr.o_on = path_origin_rounded(p);
                const Hit h = closest_hit<TRI, HAS_SPH, true>(S, ray, nullptr, rr);
                rays++;
                if (h.obj < 0) sky++; else shaded++;
                uint32_t sky_unused = 0;
                if (!path_shade(S, p, A.max_bounces, h, er, eg, eb, sky_unused)) break;
            }
        }
        A.rgb[3 * i] = p.tr * er; A.rgb[3 * i + 1] = p.tg * eg; A.rgb[3 * i + 2] = p.tb * eb;
    }
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_down(rays, off, 64); sky += __shfl_down(sky, off, 64);
        shaded += __shfl_down(shaded, off, 64); paths += __shfl_down(paths, off, 64);
    }
    if (lane_id() == 0) {
        atomicAdd(A.stats + kStatRays, rays); atomicAdd(A.stats + kStatPaths, paths);
        atomicAdd(A.stats + kStatSky, sky); atomicAdd(A.stats + kStatShaded, shaded);
    }
}

}  // namespace rt
