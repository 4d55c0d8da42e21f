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
RT_DEV void environment(const DevScene &S, float dx, float dy, float dz, float &r, float &g, float &b) {
    if (S.sky == nullptr) { r = g = b = 0.0f; return; }
    const float kInv2Pi = 0.15915494309189535f, kInvPi = 0.3183098861837907f;
    float y = min_(max_(dy, -1.0f), 1.0f);
    float u = 0.5f + atan2f(dz, dx) * kInv2Pi;
    float v = 1.0f - (0.5f + asinf(y) * kInvPi);
    int px = (int)(u * (float)S.sky_w), py = (int)(v * (float)S.sky_h);
    px = px < 0 ? 0 : (px >= S.sky_w ? S.sky_w - 1 : px);
    py = py < 0 ? 0 : (py >= S.sky_h ? S.sky_h - 1 : py);
    uint32_t c = S.sky[(size_t)py * S.sky_w + px];
    const float k = S.sky_scale * (1.0f / 255.0f);
    r = (float)(c & 0xFFu) * k; g = (float)((c >> 8) & 0xFFu) * k; b = (float)((c >> 16) & 0xFFu) * k;
}

// Camera.get_ray (camera.py:39-50) for pixel (x, y) of a W x H image (renderer.py:96-103).
RT_DEV void camera_ray(const DevScene &S, int x, int y, int W, int H, uint32_t key0, uint32_t key1,
                       float &ox, float &oy, float &oz, float &dx, float &dy, float &dz) {
    const float aspect = (float)H / (float)W;
    float u = ((float)x + 0.5f) / (float)W * 2.0f - 1.0f;
    float v = ((float)y + 0.5f) / (float)H * 2.0f * aspect - aspect;
    float tx = fma_(S.cam_d, S.cam_fwd[0], fma_(v, S.cam_up[0], u * S.cam_right[0]));
    float ty = fma_(S.cam_d, S.cam_fwd[1], fma_(v, S.cam_up[1], u * S.cam_right[1]));
    float tz = fma_(S.cam_d, S.cam_fwd[2], fma_(v, S.cam_up[2], u * S.cam_right[2]));
    float s = rsq_(fma_(tz, tz, fma_(ty, ty, tx * tx)));
    dx = tx * s; dy = ty * s; dz = tz * s;
    ox = S.cam_pos[0]; oy = S.cam_pos[1]; oz = S.cam_pos[2];
    if (S.dof_on) {
        Draws dr = segment_draws(key0, key1, 0);
        float r1 = dr.f0, r2 = dr.f1;
        float j1 = (r1 * 2.0f - 1.0f) * S.dof_strength, j2 = (r2 * 2.0f - 1.0f) * S.dof_strength;
        float gx = fma_(dx, S.dof_dist, ox), gy = fma_(dy, S.dof_dist, oy), gz = fma_(dz, S.dof_dist, oz);
        ox = fma_(S.cam_up[0], j2, fma_(S.cam_right[0], j1, ox));
        oy = fma_(S.cam_up[1], j2, fma_(S.cam_right[1], j1, oy));
        oz = fma_(S.cam_up[2], j2, fma_(S.cam_right[2], j1, oz));
        tx = gx - ox; ty = gy - oy; tz = gz - oz;
        s = rsq_(fma_(tz, tz, fma_(ty, ty, tx * tx)));
        dx = tx * s; dy = ty * s; dz = tz * s;
    }
}

// One bounce of Renderer.raytrace after a hit (renderer.py:39-70).  Returns false when the path
// ends at this surface (emissive).  Draw order: U1 unless total internal reflection, then U2, then
// the three cube-sample coordinates (vec3.py:14-16); `u` holds this segment's five uniforms and
// they are consumed positionally, so skipping U1 shifts the rest down by one.
RT_DEV bool shade_bounce(const MatRow &m, const Hit &h, const Draws &u, float &dx, float &dy, float &dz,
                         bool &inside, float &tr, float &tg, float &tb, float &er, float &eg, float &eb) {
    float nx = h.nx, ny = h.ny, nz = h.nz;
    float eta = rcp_(m.ior);
    if (inside) { nx = -nx; ny = -ny; nz = -nz; eta = m.ior; }
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
            if (fma_(rz, nz, fma_(ry, ny, rx * nx)) < 0.0f) { rx = -rx; ry = -ry; rz = -rz; }
            dx = rx; dy = ry; dz = rz;
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

RT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
RT_DEV uint32_t rank_in(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

constexpr int kJobBlock = 256;   // jobs a wave takes from the global counter at a time

template <bool HAS_TRI, bool HAS_SPH, bool DETCUT>
__global__ __launch_bounds__(256) void k_render(const DevScene S, const RenderArgs A) {
    // wave-uniform job window
    uint32_t win_next = 0, win_end = 0;
    bool drained = false;
    // per-lane job state
    bool have_job = false, in_path = false;
    uint32_t pixel = 0, slot = 0;
    int px = 0, py = 0, s_cur = 0, s_end = 0;
    float ar = 0, ag = 0, ab = 0;
    // per-lane path state
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1;
    float tr = 1, tg = 1, tb = 1;
    int depth = 0; bool inside = false; int32_t origin = -1;
    uint32_t key0 = 0, key1 = 0;
    uint32_t n_rays = 0, n_paths = 0, n_sky = 0, n_shaded = 0;
    HitCounters cnt; cnt.exact = 0; cnt.refine = 0;
    const uint32_t rect_px = (uint32_t)(A.rect_w * A.rect_h);

    for (;;) {
        // ---- 1. hand jobs to idle lanes ------------------------------------------------------
        unsigned long long idle = __ballot(!have_job);
        while (idle != 0 && !(drained && win_next == win_end)) {
            if (win_next == win_end) {
                uint32_t base = 0;
                if (lane_id() == 0) base = atomicAdd(A.job_counter, (uint32_t)kJobBlock);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= A.total_jobs) { drained = true; break; }
                win_next = base;
                win_end = base + kJobBlock < A.total_jobs ? base + kJobBlock : A.total_jobs;
            }
            uint32_t avail = win_end - win_next;
            uint32_t r = rank_in(idle);
            if (!have_job && r < avail) {
                uint32_t job = win_next + r;
                uint32_t chunk = job / rect_px, pl = job - chunk * rect_px;
                px = A.x0 + (int)(pl / (uint32_t)A.rect_h);
                py = A.y0 + (int)(pl % (uint32_t)A.rect_h);
                pixel = (uint32_t)(px * A.height + py);
                slot = job;
                s_cur = A.sample_begin + (int)chunk * A.chunk_samples;
                int lim = A.sample_begin + A.sample_count;
                s_end = s_cur + A.chunk_samples < lim ? s_cur + A.chunk_samples : lim;
                ar = ag = ab = 0.0f;
                have_job = true; in_path = false;
            }
            uint32_t want = (uint32_t)__builtin_popcountll(idle);
            win_next += want < avail ? want : avail;
            idle = __ballot(!have_job);
        }
        if (__ballot(have_job) == 0) break;

        if (have_job) {
            // ---- 2. start the next sample of this lane's pixel --------------------------------
            if (!in_path) {
                key0 = pixel ^ A.seed_lo; key1 = (uint32_t)s_cur ^ A.seed_hi;
                camera_ray(S, px, py, A.width, A.height, key0, key1, ox, oy, oz, dx, dy, dz);
                tr = tg = tb = 1.0f; depth = 0; inside = false; origin = -1; in_path = true;
                n_paths++;
            }
            // ---- 3. one path segment: closest hit, then shade ---------------------------------
            Ray ray = make_ray(S, ox, oy, oz, dx, dy, dz, origin);
            n_rays++;
            Hit h = closest_hit<HAS_TRI, HAS_SPH, DETCUT>(S, ray, cnt, nullptr);
            float er = 0, eg = 0, eb = 0;
            bool alive;
            if (h.obj < 0) {
                environment(S, dx, dy, dz, er, eg, eb);
                n_sky++;
                alive = false;
            } else {
                n_shaded++;
                const MatRow m = S.materials[h.obj];
                const Draws u = segment_draws(key0, key1, (uint32_t)depth + 1u);
                alive = shade_bounce(m, h, u, dx, dy, dz, inside, tr, tg, tb, er, eg, eb);
                ox = h.px; oy = h.py; oz = h.pz; origin = h.gprim;
                depth++;
                if (alive && depth > A.max_bounces) { alive = false; er = eg = eb = 0.0f; }   // renderer.py:29-30
            }
            if (!alive) {
                ar = fma_(tr, er, ar); ag = fma_(tg, eg, ag); ab = fma_(tb, eb, ab);
                in_path = false;
                s_cur++;
                if (s_cur >= s_end) {
                    if (A.n_chunks > 1) {
                        float *dst = A.partial + 3 * (size_t)slot;
                        dst[0] = ar; dst[1] = ag; dst[2] = ab;
                    } else {
                        float *dst = A.accum + 3 * (size_t)pixel;
                        dst[0] += ar; dst[1] += ag; dst[2] += ab;
                    }
                    have_job = false;
                }
            }
        }
    }
    // ---- statistics: one atomic per counter per wave ------------------------------------------
    unsigned long long v[kStatCount] = {0};
    v[kStatRays] = n_rays; v[kStatPaths] = n_paths; v[kStatExact] = cnt.exact; v[kStatRefine] = cnt.refine;
    v[kStatSky] = n_sky; v[kStatShaded] = n_shaded;
    v[kStatSphTests] = (unsigned long long)n_rays * (uint32_t)(S.n_sph_rows + S.n_big_sph);
    v[kStatTriTests] = (unsigned long long)n_rays * (uint32_t)S.n_tri_rows;
#pragma unroll
    for (int i = 0; i < kStatCount; ++i) {
        unsigned long long x = v[i];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane_id() == 0 && x) atomicAdd(A.stats + i, x);
    }
}

// Folds the chunk slots of a render call into the accumulation buffer, in chunk order.
__global__ __launch_bounds__(256) void k_reduce_chunks(const RenderArgs A) {
    const uint32_t rect_px = (uint32_t)(A.rect_w * A.rect_h);
    uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x;
    if (pl >= rect_px) return;
    int px = A.x0 + (int)(pl / (uint32_t)A.rect_h), py = A.y0 + (int)(pl % (uint32_t)A.rect_h);
    float r = 0, g = 0, b = 0;
    for (int c = 0; c < A.n_chunks; ++c) {
        const float *src = A.partial + 3 * ((size_t)c * rect_px + pl);
        r += src[0]; g += src[1]; b += src[2];
    }
    float *dst = A.accum + 3 * ((size_t)px * A.height + py);
    dst[0] += r; dst[1] += g; dst[2] += b;
}

}  // namespace rt
