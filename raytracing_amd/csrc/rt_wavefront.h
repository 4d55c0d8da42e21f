// Wavefront form of the render path for scenes searched through the hierarchy (optional: RT_RENDER_WAVEFRONT;
// the single-kernel pipeline of rt_render.h is the default because it measures faster - see the end of this comment).
//
// The single render kernel keeps a whole path in registers.  Here the path is cut at the closest-hit query and the
// rays of one path segment are worked on queue by queue:
//
//   k_wf_generate   camera rays of a batch of paths (Camera.get_ray, camera.py:39-50)
//   k_wf_extend     closest hit of every queued ray (Scene.intersect, scene.py:23-27) - a LEAN, persistent kernel
//                   (no double-precision settling code: five waves per SIMD): lanes take rays from the queue one by
//                   one and walk the hierarchy in slices, so a lane whose ray is done takes the next ray at once; a
//                   query the float32 code cannot answer alone goes, with its first pass's result, on a second list
//   k_wf_settle     those queries: judged and settled in double precision (hit_finish, the full form)
//   k_wf_shade      environment or one bounce per answered ray (renderer.py:39-79), MATERIAL-SORTED: the 256 rays of
//                   a workgroup are regrouped by what they hit (environment, then by object = by material) with a
//                   counting sort in LDS before they are shaded, so the lanes of a wave run the same material's
//                   branch of renderer.py:45-70 and read the same MatRow; surviving paths append their next ray to
//                   the other queue, finished ones store their radiance
//   k_wf_resolve    per pixel, adds the batch's path radiances to the accumulation buffer in sample order
//                   (deterministic: no floating-point atomics anywhere)
//
// and the host runs extend + settle + shade once per path segment.  Path state lives in HBM between the kernels
// (32-byte ray, 32-byte path, 48-byte hit record: ~150 bytes moved per segment, far from the HBM bound).  The device
// functions - camera_ray, hit_begin / search_bvh / hit_finish, path_shade - are the ones the single kernel uses, so
// both forms trace identical paths.  Same-address atomics cost ~8 ns each on this GPU: queue appends and statistics
// are aggregated per wave or per block.
//
// Measured (room + 1.18 M triangles, 1920x1080, 32 spp, one MI355X): 73.6 ms with 64 M paths per batch, 103 ms with 16 M,
// against 28.6 ms for the single persistent kernel - nine segments x three launches per batch each end with waves
// waiting for their slowest lanes, the late segments' queues are too short to fill the chip, and every ray's state is
// cold when its lane picks it up.  The single kernel is the default; this form is an option (RT_RENDER_WAVEFRONT).
#pragma once
#include <hip/hip_runtime.h>

#include "rt_render.h"

namespace rt {

struct WfRay { float ox, oy, oz; int32_t origin; float dx, dy, dz; uint32_t path; };   // 32 B
struct WfPath { float tr, tg, tb; uint32_t inside; uint32_t pixel, sample; int32_t depth; uint32_t pad; float rox, roy, roz; uint32_t pad1; };   // 48 B; ro*: Path::rox ...
struct WfHit {
    float nx, ny, nz; int32_t obj;
    float px, py, pz; int32_t gprim;
    uint32_t flags;   // bit 0: contenders settled in double precision, bit 1: winner re-evaluated in double precision
    uint32_t first[3];   // a query on the settle list: its first pass's result (PairBest::lo0, lo1, id0 as bits)
    float rpx, rpy, rpz; uint32_t pad;   // Hit::rpx ...
};   // 64 B

struct WfArgs {
    int32_t width, height, x0, y0, rect_w, rect_h;
    const float *ray_dir, *ray_res;
    int32_t sample_begin;        // global index of the batch's first sample
    int32_t batch_samples;       // samples per pixel in this batch
    uint32_t n_paths;            // rect_w * rect_h * batch_samples
    FastDiv div_samples, div_rect_h;
    int32_t max_bounces;
    uint32_t seed_lo, seed_hi;
    WfRay *rays_in, *rays_out;
    WfPath *paths;
    WfHit *hits;
    float *radiance;             // [n_paths][3]
    uint32_t *count_in, *count_out, *cursor;   // device words: rays queued for this / the next segment, extend's read cursor
    uint32_t *settle_list, *settle_count;      // queue positions of the rays whose query k_wf_settle has to close, and their number
    float *accum;
    unsigned long long *stats;
    unsigned long long *partials;   // [blocks][kStatCount] per-block statistics, folded by k_wf_fold_stats (no hot atomics)
};

// path index -> (pixel of the rectangle, sample of the batch): consecutive paths are the samples of one pixel
RT_DEV void wf_path_coords(const WfArgs &A, uint32_t p, uint32_t &pixel, uint32_t &sample) {
    const uint32_t pl = fast_div(p, A.div_samples), j = p - pl * (uint32_t)A.batch_samples;
    const uint32_t col = fast_div(pl, A.div_rect_h);
    const int px = A.x0 + (int)col, py = A.y0 + (int)(pl - col * (uint32_t)A.rect_h);
    pixel = (uint32_t)(px * A.height + py);
    sample = (uint32_t)A.sample_begin + j;
}

__global__ __launch_bounds__(256) void k_wf_generate(const DevScene S_arg, const WfArgs A) {
    SceneRef S = *kernarg_scene();   // the scene argument, read in place (scalar loads)
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.n_paths) return;
    uint32_t pixel, sample;
    wf_path_coords(A, p, pixel, sample);
    Path q;
    path_begin(S, A.ray_dir, A.ray_res, q, pixel, sample, A.seed_lo, A.seed_hi);
    WfRay r; r.ox = q.ox; r.oy = q.oy; r.oz = q.oz; r.origin = -1; r.dx = q.dx; r.dy = q.dy; r.dz = q.dz; r.path = p;
    WfPath s; s.tr = s.tg = s.tb = 1.0f; s.inside = 0; s.pixel = pixel; s.sample = sample; s.depth = 0; s.pad = 0; s.rox = s.roy = s.roz = 0.0f; s.pad1 = 0;
    A.rays_in[p] = r;
    A.paths[p] = s;
    if (p == 0) *A.count_in = A.n_paths;
}

constexpr int kWfSlice = 3;      // wide nodes per trip of the extend loop
constexpr int kWfGrab = 256;    // rays a wave reserves from the queue at a time (one atomic each: keep them rare,
                                 // same-address atomics complete at only ~0.1 G/s)

template <bool HAS_SPH>
__global__ __launch_bounds__(256, 5) void k_wf_extend(const DevScene S_arg, const WfArgs A) {
    SceneRef S = *kernarg_scene();   // the scene argument, read in place (scalar loads)
    constexpr int TRI = kTriBvh;
    const uint32_t total = *A.count_in;
    uint32_t win_next = 0, win_end = 0;
    bool drained = false;
    bool have = false, walking = false;
    WfRay ray; ray.ox = ray.oy = ray.oz = 0; ray.dx = ray.dy = 0; ray.dz = 1; ray.origin = -1; ray.path = 0;
    uint32_t pixel = 0, slot = 0; int32_t depth = 0;
    float rox = 0.0f, roy = 0.0f, roz = 0.0f;   // residue of the ray's origin (RayResidue)
    typename FirstPass<TRI>::type first;
    first.init(S);
    int walk_node = -1;
    BvhCounters bc = bvh_counters();   // per-lane test counts, reduced at the end
    for (;;) {
        // ---- idle lanes take the next rays of the queue -------------------------------------------
        unsigned long long idle = __ballot(!have);
        while (idle != 0 && !(drained && win_next == win_end)) {
            if (win_next == win_end) {
                uint32_t base = 0;
                if (lane_id() == 0) base = atomicAdd(A.cursor, (uint32_t)kWfGrab);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= total) { drained = true; break; }
                win_next = base;
                win_end = base + kWfGrab < total ? base + kWfGrab : total;
            }
            const uint32_t avail = win_end - win_next;
            const uint32_t rk = rank_in(idle);
            if (!have && rk < avail) {
                slot = win_next + rk;
                ray = A.rays_in[slot];
                const WfPath s = A.paths[ray.path];
                pixel = s.pixel; depth = s.depth; rox = s.rox; roy = s.roy; roz = s.roz;
                have = true; walking = false;
            }
            const uint32_t want = (uint32_t)__builtin_popcountll(idle);
            win_next += want < avail ? want : avail;
            idle = __ballot(!have);
        }
        if (__ballot(have) == 0) break;
        bool to_settle = false;
        if (have) {
            const bool o_on = rox != 0.0f || roy != 0.0f || roz != 0.0f;
            const Ray r = make_ray(S, ray.ox, ray.oy, ray.oz, ray.dx, ray.dy, ray.dz, ray.origin, depth == 0, origin_tag(S, ray.origin), kInf, o_on);
            RayResidue rr;
            rr.table = A.ray_res; rr.pixel = pixel; rr.on = (depth == 0) && !S.dof_on;
            rr.ox = rox; rr.oy = roy; rr.oz = roz; rr.o_on = o_on;
            if (!walking) {
                walking = true;
                walk_node = -1;
                first.init(S);
                if (ray_route<TRI>(r) == kRouteFilter) { hit_begin<TRI, HAS_SPH>(S, r, rr, first); walk_node = 0; }
            }
            if (walk_node >= 0) search_bvh(S, r, first, bc, walk_node, kWfSlice);
            if (walk_node < 0) {
                const Hit h = hit_finish<TRI, HAS_SPH, true, true>(S, r, rr, first, bvh_counters(), nullptr);
                WfHit w;
                w.nx = h.nx; w.ny = h.ny; w.nz = h.nz; w.obj = h.obj;
                w.px = h.px; w.py = h.py; w.pz = h.pz; w.gprim = h.gprim;
                w.flags = h.refined ? 2u : 0u;
                w.first[0] = f2u(first.lo0); w.first[1] = f2u(first.lo1); w.first[2] = (uint32_t)first.id0;
                w.rpx = h.rpx; w.rpy = h.rpy; w.rpz = h.rpz; w.pad = 0;
                A.hits[ray.path] = w;
                to_settle = h.how == kHowDeferred;
                have = false; walking = false;
            }
        }
        // queries for k_wf_settle: one counter bump per wave
        const unsigned long long sm = __ballot(to_settle);
        if (sm != 0) {
            uint32_t base = 0;
            if (lane_id() == (uint32_t)__builtin_ctzll(sm)) base = atomicAdd(A.settle_count, (uint32_t)__builtin_popcountll(sm));
            base = (uint32_t)__shfl((int)base, __builtin_ctzll(sm), 64);
            if (to_settle) A.settle_list[base + rank_in(sm)] = slot;
        }
    }
    // statistics: per-block sums into the block's own row of `partials` (launches on one stream
    // run one after another, so a plain read-modify-write is safe)
    __shared__ unsigned long long sh_nodes[4], sh_tris[4];
    unsigned long long nodes = bc.nodes, tris = bc.tris;
    for (int off = 32; off > 0; off >>= 1) { nodes += __shfl_down(nodes, off, 64); tris += __shfl_down(tris, off, 64); }
    const int wave = threadIdx.x >> 6;
    if (lane_id() == 0) { sh_nodes[wave] = nodes; sh_tris[wave] = tris; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *row = A.partials + (size_t)blockIdx.x * kStatCount;
        row[kStatBvhNodes] += sh_nodes[0] + sh_nodes[1] + sh_nodes[2] + sh_nodes[3];
        row[kStatTriTests] += sh_tris[0] + sh_tris[1] + sh_tris[2] + sh_tris[3];
    }
}

// The queries the lean extend kernel could not close: each goes on from its first pass's result (as the single-kernel
// pipeline's set-aside paths do) through the full hit_finish - judge, double-precision settling, second walk.
template <bool HAS_SPH>
__global__ __launch_bounds__(256) void k_wf_settle(const DevScene S_arg, const WfArgs A) {
    SceneRef S = *kernarg_scene();   // the scene argument, read in place (scalar loads)
    constexpr int TRI = kTriBvh;
    const uint32_t n = *A.settle_count;
    BvhCounters bc = bvh_counters();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const WfRay ray = A.rays_in[A.settle_list[i]];
        const WfPath s = A.paths[ray.path];
        WfHit w = A.hits[ray.path];
        const bool o_on = s.rox != 0.0f || s.roy != 0.0f || s.roz != 0.0f;
        const Ray r = make_ray(S, ray.ox, ray.oy, ray.oz, ray.dx, ray.dy, ray.dz, ray.origin, s.depth == 0, origin_tag(S, ray.origin), kInf, o_on);
        RayResidue rr;
        rr.table = A.ray_res; rr.pixel = s.pixel; rr.on = (s.depth == 0) && !S.dof_on;
        rr.ox = s.rox; rr.oy = s.roy; rr.oz = s.roz; rr.o_on = o_on;
        typename FirstPass<TRI>::type first;
        first.init(S);
        first.lo0 = u2f(w.first[0]); first.lo1 = u2f(w.first[1]); first.id0 = (int32_t)w.first[2];
        const Hit h = hit_finish<TRI, HAS_SPH, true, false>(S, r, rr, first, bvh_counters(), nullptr);
        bc.nodes += h.bvh_nodes; bc.tris += h.bvh_tris;
        w.nx = h.nx; w.ny = h.ny; w.nz = h.nz; w.obj = h.obj;
        w.px = h.px; w.py = h.py; w.pz = h.pz; w.gprim = h.gprim;
        w.flags = (h.how == 2 ? 1u : 0u) | (h.refined ? 2u : 0u);
        w.rpx = h.rpx; w.rpy = h.rpy; w.rpz = h.rpz;
        A.hits[ray.path] = w;
    }
    // (second walks are counted like the first: per-block sums, one plain update of the block's own row)
    __shared__ unsigned long long sh_nodes[4], sh_tris[4];
    unsigned long long nodes = bc.nodes, tris = bc.tris;
    for (int off = 32; off > 0; off >>= 1) { nodes += __shfl_down(nodes, off, 64); tris += __shfl_down(tris, off, 64); }
    const int wave = threadIdx.x >> 6;
    if (lane_id() == 0) { sh_nodes[wave] = nodes; sh_tris[wave] = tris; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *row = A.partials + (size_t)blockIdx.x * kStatCount;
        row[kStatBvhNodes] += sh_nodes[0] + sh_nodes[1] + sh_nodes[2] + sh_nodes[3];
        row[kStatTriTests] += sh_tris[0] + sh_tris[1] + sh_tris[2] + sh_tris[3];
    }
}

__global__ __launch_bounds__(256) void k_wf_shade(const DevScene S_arg, const WfArgs A) {
    SceneRef S = *kernarg_scene();   // the scene argument, read in place (scalar loads)
    const uint32_t total = *A.count_in;
    // Material sort: the workgroup's 256 rays are regrouped by what they hit - the environment first, then by
    // object (every object has its own MatRow) - with a counting sort in LDS, and thread t shades the ray at sorted
    // position t.  Which ray a thread shades changes nothing in the ray's result; it changes which lanes share a
    // wave: lanes that read the same material and take the same branch of renderer.py:45-70.
    constexpr int kBins = 18;   // environment, objects 0 .. 15 (higher ones share the last bin), no ray
    __shared__ uint32_t sort_hist[kBins], sort_base[kBins];
    __shared__ uint16_t sort_perm[256];
    const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x;
    int key = kBins - 1;
    if (i0 < total) {
        const int32_t obj = A.hits[A.rays_in[i0].path].obj;
        key = obj < 0 ? 0 : (obj < 16 ? obj + 1 : 16);
    }
    if (threadIdx.x < kBins) sort_hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t in_bin = atomicAdd(&sort_hist[key], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int k = 0; k < kBins; ++k) { sort_base[k] = run; run += sort_hist[k]; } }
    __syncthreads();
    sort_perm[sort_base[key] + in_bin] = (uint16_t)threadIdx.x;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + sort_perm[threadIdx.x];
    const bool live = i < total;
    bool alive = false, f_exact = false, f_refine = false, f_sky = false, f_shaded = false, f_done = false;
    WfRay next; next.ox = next.oy = next.oz = 0; next.dx = next.dy = next.dz = 0; next.origin = -1; next.path = 0;
    if (live) {
        const WfRay ray = A.rays_in[i];
        const WfPath s = A.paths[ray.path];
        const WfHit w = A.hits[ray.path];
        Path p;
        p.ox = ray.ox; p.oy = ray.oy; p.oz = ray.oz; p.dx = ray.dx; p.dy = ray.dy; p.dz = ray.dz;
        p.tr = s.tr; p.tg = s.tg; p.tb = s.tb;
        p.key0 = s.pixel ^ A.seed_lo; p.key1 = s.sample ^ A.seed_hi;
        p.depth = s.depth; p.origin = ray.origin; p.inside = s.inside != 0; p.pixel = s.pixel;
        Hit h;
        h.obj = w.obj; h.prim = 0; h.gprim = w.gprim; h.t = 0; h.nx = w.nx; h.ny = w.ny; h.nz = w.nz;
        h.px = w.px; h.py = w.py; h.pz = w.pz; h.how = 0; h.refined = false; h.bvh_nodes = 0; h.bvh_tris = 0; h.tag = 0u; h.own_ratio = kInf;
        h.rpx = w.rpx; h.rpy = w.rpy; h.rpz = w.rpz;
        float er, eg, eb;
        uint32_t sky_unused = 0;   // (this kernel counts from its own flags)
        alive = path_shade(S, p, A.max_bounces, h, er, eg, eb, sky_unused);
        f_exact = (w.flags & 1u) != 0; f_refine = (w.flags & 2u) != 0; f_sky = w.obj < 0; f_shaded = w.obj >= 0;
        if (alive) {
            WfPath t = s;
            t.tr = p.tr; t.tg = p.tg; t.tb = p.tb; t.depth = p.depth; t.inside = p.inside ? 1u : 0u;
            t.rox = p.rox; t.roy = p.roy; t.roz = p.roz;
            A.paths[ray.path] = t;
            next.ox = p.ox; next.oy = p.oy; next.oz = p.oz; next.origin = p.origin;
            next.dx = p.dx; next.dy = p.dy; next.dz = p.dz; next.path = ray.path;
        } else {
            float *dst = A.radiance + 3 * (size_t)ray.path;
            dst[0] = p.tr * er; dst[1] = p.tg * eg; dst[2] = p.tb * eb;
            f_done = true;
        }
    }
    // surviving paths append their next ray: one counter update per BLOCK
    __shared__ uint32_t sh_keep[4], sh_base;
    __shared__ uint32_t sh_stat[4][6];
    const int wave = threadIdx.x >> 6;
    const unsigned long long keep = __ballot(alive);
    const uint32_t n_ray = (uint32_t)__builtin_popcountll(__ballot(live));
    const uint32_t n_exact = (uint32_t)__builtin_popcountll(__ballot(f_exact)), n_refine = (uint32_t)__builtin_popcountll(__ballot(f_refine));
    const uint32_t n_sky = (uint32_t)__builtin_popcountll(__ballot(f_sky)), n_shaded = (uint32_t)__builtin_popcountll(__ballot(f_shaded));
    const uint32_t n_done = (uint32_t)__builtin_popcountll(__ballot(f_done));
    if (lane_id() == 0) {
        sh_keep[wave] = (uint32_t)__builtin_popcountll(keep);
        sh_stat[wave][0] = n_ray; sh_stat[wave][1] = n_exact; sh_stat[wave][2] = n_refine;
        sh_stat[wave][3] = n_sky; sh_stat[wave][4] = n_shaded; sh_stat[wave][5] = n_done;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n_keep = sh_keep[0] + sh_keep[1] + sh_keep[2] + sh_keep[3];
        sh_base = n_keep ? atomicAdd(A.count_out, n_keep) : 0u;
        uint32_t t[6];
        for (int k = 0; k < 6; ++k) t[k] = sh_stat[0][k] + sh_stat[1][k] + sh_stat[2][k] + sh_stat[3][k];
        if (t[0]) {
            unsigned long long *row = A.partials + (size_t)blockIdx.x * kStatCount;
            row[kStatRays] += t[0];
            row[kStatSphTests] += (unsigned long long)t[0] * (uint32_t)(S.n_sph_rows + S.n_big_sph);
            row[kStatTriTests] += (unsigned long long)t[0] * (uint32_t)S.n_tri_rows;
            row[kStatExact] += t[1]; row[kStatRefine] += t[2]; row[kStatSky] += t[3]; row[kStatShaded] += t[4]; row[kStatPaths] += t[5];
        }
    }
    __syncthreads();
    if (alive) {
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += sh_keep[w];
        A.rays_out[sh_base + before + rank_in(keep)] = next;
    }
}

// Folds the per-block statistics rows into the context's counters and clears them.
__global__ __launch_bounds__(256) void k_wf_fold_stats(unsigned long long *partials, int n_rows, unsigned long long *stats) {
    __shared__ unsigned long long sh[256];
    for (int k = 0; k < kStatCount; ++k) {
        unsigned long long v = 0;
        for (int r = threadIdx.x; r < n_rows; r += 256) { v += partials[(size_t)r * kStatCount + k]; partials[(size_t)r * kStatCount + k] = 0; }
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
        if (threadIdx.x == 0 && sh[0]) atomicAdd(stats + k, sh[0]);
        __syncthreads();
    }
}

// Adds the batch's path radiances to the accumulation buffer, per pixel, in sample order.
__global__ __launch_bounds__(256) void k_wf_resolve(const WfArgs A) {
    const uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x;
    if (pl >= (uint32_t)(A.rect_w * A.rect_h)) return;
    const uint32_t col = fast_div(pl, A.div_rect_h);
    const int px = A.x0 + (int)col, py = A.y0 + (int)(pl - col * (uint32_t)A.rect_h);
    float r = 0, g = 0, b = 0;
    const float *src = A.radiance + 3 * (size_t)pl * (size_t)A.batch_samples;
    for (int j = 0; j < A.batch_samples; ++j) { r += src[3 * j]; g += src[3 * j + 1]; b += src[3 * j + 2]; }
    float *dst = A.accum + 3 * ((size_t)px * A.height + py);
    dst[0] += r; dst[1] += g; dst[2] += b;
}

}  // namespace rt
