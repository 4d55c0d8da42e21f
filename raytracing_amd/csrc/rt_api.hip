// C ABI of the path-tracing backend (include/raytracing_amd.h): context, scene upload (double
// tables -> float32 filter rows with error-bound constants), the closest-hit entry and the render
// entry.  gfx950 only; built with -ffp-contract=off (the exact double-precision code must round
// like upstream's Python floats, and the float32 code spells out every fused multiply-add).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is opened when rt_comm_* is first called

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracing_amd.h"
#include "rt_bvh_build.h"
#include "rt_exact.h"
#include "rt_filter.h"
#include "rt_obj.h"
#include "rt_post.h"
#include "rt_render.h"
#include "rt_scene.h"
#include "rt_wavefront.h"

using namespace rt;

namespace {

thread_local std::string g_create_error;
std::atomic<int> g_build_threads{0};   // rt_build_bvh_threads

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, n ? n : 1);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <class T> hipError_t upload(const std::vector<T> &v) {
        hipError_t e = ensure(std::max<size_t>(v.size() * sizeof(T), 16));
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

struct EventPair { hipEvent_t a, b; };

// Fixed device-side scratch of rt_postprocess_noalloc: the per-block reduction slots of k_post_measure
// plus the folded pair, laid out as PostArgs::partials.  Every GPU the library touches holds its own
// copy of this array (32,784 bytes) at its own address until the process ends.  In host code the
// array's NAME is only a handle for hipGetSymbolAddress; the address kernels and copies use is the
// one looked up per device in scratch_for_device().
constexpr int kPostScratchBlocks = 2048;
static __device__ __align__(256) double scratch[2 * kPostScratchBlocks + 2];

constexpr int kMaxScratchDevices = 64;
std::mutex g_scratch_mu;
double *g_scratch_ptr[kMaxScratchDevices] = {nullptr};

// Device address of `scratch` on `device`, which must be the current device.  Looked up once per device.
hipError_t scratch_for_device(int device, double **out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxScratchDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        double *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

}  // namespace

struct rt_ctx {
    int device = 0;
    int n_cu = 256;
    std::mutex mu;               // guards err and the host-side stats bookkeeping
    char err[512] = {0};         // fixed storage: rt_last_error hands out this address, which never moves
    std::atomic<int> stop{0};
    bool has_scene = false;
    DevScene S{};
    int tri_mode = 0; bool has_sph = false;
    DevBuf tri_rows, quad_rows, slab_rows, box_rows, box_face_id, box_face_rows, leaf_rows, sph_rows, sph_loop, sph_loop_id, big_sph, tri_by_id, tri_aux, sph_by_id, sph_obj, materials;
    DevBuf tri_tight;
    DevBuf tri64, sph64, aabb64, mesh_tri_begin, obj_kind, obj_index, sky, bvh, bvh_wide, bvh_tri, cold;
    int32_t walk_stack_need = 0;   // stack entries the deepest walk of the current wide hierarchy can ask for
    DevBuf accum; int acc_w = 0, acc_h = 0;
    DevBuf partial, job_counter, stats;
    DevBuf aside, aside_head, dirty, dirty_list;   // rt_render: paths set aside by the lean kernel, jobs it gave up (grow-only)
    DevBuf post_partials;   // rt_postprocess: per-block reduction slots
    double *post_scratch = nullptr;   // rt_postprocess_noalloc: this GPU's copy of the static scratch array
    DevBuf wf_rays[2], wf_paths, wf_hits, wf_radiance, wf_counters, wf_partials, wf_settle;   // wavefront form (hierarchy scenes)
    DevBuf ray_dir, ray_res; int ray_w = 0, ray_h = 0;   // camera-ray table of the current scene at ray_w x ray_h
    // The hierarchy rt_set_scene last BUILT (host copy: threaded depth-first nodes, leaf order, which triangles stay in
    // the linear tables) and what it was built for: a later scene that differs only in vertex positions can be
    // refitted over it (RT_SCENE_REFIT)
    std::vector<rt_bvh_node> topo_nodes; std::vector<int32_t> topo_order, topo_mesh_begin; std::vector<char> topo_linear;
    int32_t topo_leaf = 0; uint32_t topo_flags = 0; double topo_area = 0;   // topo_area: summed box areas of that hierarchy when it was built
    bool last_refit = false;
    DevBuf isect_in, isect_out;          // rt_intersect_batch staging, grow-only (one call in flight per context)
    std::vector<EventPair> timed;        // recorded around render launches, not yet read
    std::vector<EventPair> spare_events; // read and ready for reuse (events are created once, not per launch)
    double kernel_ms = 0; uint64_t launches = 0;
    double kernel_ms_min = 0, kernel_ms_max = 0; uint64_t launches_timed = 0;
    ncclComm_t comm = nullptr; int comm_ranks = 1, comm_rank = 0;   // rt_comm_init
    uint64_t paths_launched = 0;         // paths of rt_render's single-kernel launches: rectangle x samples, known here, not counted on the device
};

namespace {

int fail(rt_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) { std::lock_guard<std::mutex> g(c->mu); std::memcpy(c->err, buf, sizeof c->err); c->err[sizeof c->err - 1] = 0; } else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail((c), RT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct V3 { double x, y, z; };
V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
double norm(V3 a) { return std::sqrt(dot(a, a)); }
double norm1(V3 a) { return std::fabs(a.x) + std::fabs(a.y) + std::fabs(a.z); }

// RCCL is opened on first use (dlopen), so that single-GPU users of this library neither link nor
// load it.  Only four entry points are needed.
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*error_string)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { const char *e = dlerror(); r.why = std::string("cannot open librccl.so: ") + (e ? e : "?"); return; }
        r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.handle, "ncclGetUniqueId");
        r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.handle, "ncclCommInitRank");
        r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.handle, "ncclCommDestroy");
        r.all_reduce = (decltype(r.all_reduce))dlsym(r.handle, "ncclAllReduce");
        r.error_string = (decltype(r.error_string))dlsym(r.handle, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) { r.why = "librccl.so lacks a required symbol"; r.handle = nullptr; }
    });
    return r;
}

const char *rccl_error(ncclResult_t e) { Rccl &r = rccl(); return r.error_string ? r.error_string(e) : "RCCL error"; }

FastDiv make_fast_div(uint32_t d) {
    FastDiv f{}; f.d = d;
    uint32_t L = 0;
    while ((1ull << L) < d) ++L;                       // ceil(log2 d)
    f.m = (uint32_t)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
    f.sh1 = L < 1 ? L : 1; f.sh2 = L > 0 ? L - 1 : 0;
    return f;
}

// Largest float not above / smallest float not below a positive double (bounds must stay bounds).
float up(double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, INFINITY) : f; }

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char *rt_last_error(const rt_ctx *ctx) {
    if (!ctx) return g_create_error.c_str();
    return ctx->err;
}

int rt_create(int device_id, rt_ctx **out) {
    if (!out) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device_id < 0 || device_id >= n) return fail(nullptr, RT_ERR_NO_DEVICE, "device %d out of range (%d visible)", device_id, n);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(nullptr, RT_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    rt_ctx *c = new rt_ctx();
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount;
    if (hipSetDevice(device_id) != hipSuccess || c->job_counter.ensure(256) != hipSuccess ||
        c->stats.ensure(sizeof(unsigned long long) * (kStatCount + 2 * kProfCount)) != hipSuccess ||
        hipMemset(c->stats.p, 0, sizeof(unsigned long long) * (kStatCount + 2 * kProfCount)) != hipSuccess) {
        delete c;
        return fail(nullptr, RT_ERR_HIP, "device %d: initial allocation failed", device_id);
    }
    {
        const hipError_t e = scratch_for_device(device_id, &c->post_scratch);   // device_id is current here
        if (e != hipSuccess) {
            delete c;
            return fail(nullptr, RT_ERR_HIP, "device %d: hipGetSymbolAddress(scratch) failed: %s", device_id, hipGetErrorString(e));
        }
    }
    *out = c;
    return RT_OK;
}

int rt_destroy(rt_ctx *ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->comm) { (void)rccl().comm_destroy(ctx->comm); ctx->comm = nullptr; }
    for (auto *v : {&ctx->timed, &ctx->spare_events})
        for (auto &e : *v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    delete ctx;
    return RT_OK;
}

struct rt_obj { ObjFile f; };

int rt_obj_open(const char *path, rt_obj **out, int32_t *n_objects, int64_t *n_vertices, int64_t *n_triangles) {
    if (!path || !out) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    rt_obj *o = new rt_obj();
    if (!parse_obj(path, o->f)) { g_create_error = o->f.error; delete o; return RT_ERR_INVALID_ARG; }
    if (n_objects) *n_objects = (int32_t)o->f.names.size();
    if (n_vertices) *n_vertices = (int64_t)o->f.vertices.size() / 3;
    if (n_triangles) *n_triangles = (int64_t)o->f.triangles.size() / 3;
    *out = o;
    return RT_OK;
}

int rt_obj_read(const rt_obj *o, char *names, int32_t name_stride, int64_t *vertex_begin, int64_t *tri_begin, double *vertices,
                int32_t *triangles) {
    if (!o) return RT_ERR_INVALID_ARG;
    const size_t n = o->f.names.size();
    if (names && name_stride > 0)
        for (size_t i = 0; i < n; ++i) {
            std::memset(names + i * (size_t)name_stride, 0, (size_t)name_stride);
            std::memcpy(names + i * (size_t)name_stride, o->f.names[i].data(), std::min<size_t>(o->f.names[i].size(), (size_t)name_stride - 1));
        }
    if (vertex_begin) std::memcpy(vertex_begin, o->f.vertex_begin.data(), sizeof(int64_t) * (n ? n + 1 : 1));
    if (tri_begin) std::memcpy(tri_begin, o->f.tri_begin.data(), sizeof(int64_t) * (n ? n + 1 : 1));
    if (vertices && !o->f.vertices.empty()) std::memcpy(vertices, o->f.vertices.data(), sizeof(double) * o->f.vertices.size());
    if (triangles && !o->f.triangles.empty()) std::memcpy(triangles, o->f.triangles.data(), sizeof(int32_t) * o->f.triangles.size());
    return RT_OK;
}

int rt_obj_close(rt_obj *o) { delete o; return RT_OK; }

int rt_build_bvh(int32_t n_tris, const double *tri_verts, int32_t leaf_size, rt_bvh_node *nodes_out, int32_t nodes_capacity,
                 int32_t *tri_index_out, int32_t *n_nodes_out) {
    if (n_tris <= 0 || n_tris >= (1 << 27) || !tri_verts || !tri_index_out || !n_nodes_out || nodes_capacity < 0) return RT_ERR_INVALID_ARG;
    bvh::Builder b;
    b.run(n_tris, tri_verts, leaf_size > 0 ? leaf_size : 2, nullptr, g_build_threads.load());
    *n_nodes_out = (int32_t)b.nodes.size();
    std::memcpy(tri_index_out, b.order.data(), sizeof(int32_t) * (size_t)n_tris);
    if ((int64_t)b.nodes.size() > nodes_capacity || !nodes_out) return nodes_out ? RT_ERR_INVALID_ARG : RT_OK;
    std::memcpy(nodes_out, b.nodes.data(), sizeof(rt_bvh_node) * b.nodes.size());
    return RT_OK;
}

int rt_collapse_bvh(int32_t n_nodes, const rt_bvh_node *nodes, void *wide_out, int32_t wide_capacity, int32_t *n_wide_out,
                    int32_t *stack_need_out) {
    if (n_nodes <= 0 || !nodes || !n_wide_out || wide_capacity < 0) return RT_ERR_INVALID_ARG;
    // (structure only: the links must point forward and nest, as rt_set_scene requires of a caller's hierarchy)
    for (int32_t i = 0; i < n_nodes; ++i) {
        const int32_t end = nodes[i].skip < 0 ? n_nodes : nodes[i].skip;
        if (end <= i || end > n_nodes || (nodes[i].tri_begin_count < 0 && end < i + 3)) return RT_ERR_INVALID_ARG;
    }
    std::vector<rt_bvh_node> src(nodes, nodes + n_nodes);
    std::vector<WideNode> wide;
    int32_t need = 0;
    if (!bvh::collapse_wide(src, wide, &need)) return RT_ERR_UNSUPPORTED;
    *n_wide_out = (int32_t)wide.size();
    if (stack_need_out) *stack_need_out = need;
    if (!wide_out) return RT_OK;
    if ((int64_t)wide.size() > wide_capacity) return RT_ERR_INVALID_ARG;
    std::memcpy(wide_out, wide.data(), sizeof(WideNode) * wide.size());
    return RT_OK;
}

int rt_build_bvh_threads(int32_t n_threads) {
    if (n_threads < 0) return RT_ERR_INVALID_ARG;
    g_build_threads.store(n_threads);
    return RT_OK;
}

int rt_set_scene(rt_ctx *c, const rt_scene_desc *d) {
    if (!c || !d) return fail(c, RT_ERR_INVALID_ARG, "rt_set_scene: NULL argument");
    if (d->struct_size != sizeof(rt_scene_desc))
        return fail(c, RT_ERR_INVALID_ARG, "rt_scene_desc.struct_size %u != %zu", d->struct_size, sizeof(rt_scene_desc));
    if (d->n_objects < 0 || d->n_spheres < 0 || d->n_meshes < 0 || d->n_tris < 0)
        return fail(c, RT_ERR_INVALID_ARG, "negative count in scene description");
    if ((d->n_objects && (!d->obj_kind || !d->obj_index || !d->materials)) || (d->n_spheres && (!d->spheres || !d->sphere_obj)) ||
        (d->n_meshes && (!d->mesh_aabb || !d->mesh_obj || !d->mesh_tri_begin)) || (d->n_tris && !d->tri_verts))
        return fail(c, RT_ERR_INVALID_ARG, "NULL table with a non-zero count");
    if (d->n_objects != d->n_spheres + d->n_meshes)
        return fail(c, RT_ERR_INVALID_ARG, "n_objects %d != n_spheres %d + n_meshes %d", d->n_objects, d->n_spheres, d->n_meshes);
    for (int i = 0; i < d->n_objects; ++i) {
        int k = d->obj_kind[i], ix = d->obj_index[i];
        if ((k != RT_OBJ_SPHERE && k != RT_OBJ_MESH) || ix < 0 || ix >= (k == RT_OBJ_SPHERE ? d->n_spheres : d->n_meshes))
            return fail(c, RT_ERR_INVALID_ARG, "object %d: bad kind/index (%d, %d)", i, k, ix);
    }
    if (d->n_meshes && (d->mesh_tri_begin[0] != 0 || d->mesh_tri_begin[d->n_meshes] != d->n_tris))
        return fail(c, RT_ERR_INVALID_ARG, "mesh_tri_begin does not span [0, n_tris]");
    for (int m = 0; m < d->n_meshes; ++m)
        if (d->mesh_tri_begin[m + 1] < d->mesh_tri_begin[m]) return fail(c, RT_ERR_INVALID_ARG, "mesh_tri_begin not monotone");
    const long long n_gprim = (long long)d->n_spheres + d->n_tris;
    if (d->n_tris >= (1 << 27)) return fail(c, RT_ERR_UNSUPPORTED, "%d triangles: at most 2^27 - 1", d->n_tris);
    // Triangle search: the linear table walk for a few dozen triangles (and never more than 4096
    // primitives), the hierarchy beyond that (built here when the caller brings none) or whenever
    // the caller asks for it (n_bvh_nodes != 0).
    // Measured on room + k triangles of a mesh (G rays/s, linear / hierarchy): 38 triangles 24.7 / 23.5,
    // 54: 18.8 / 21.8, 86: 12.5 / 19.4, 150: 7.6 / 16.9, 310: 3.7 / 13.8 - so the linear tables are left at 48.
    if (d->linear_max_tris < 0 || d->bvh_leaf_size < 0 || d->build_threads < 0)
        return fail(c, RT_ERR_INVALID_ARG, "negative tuning value in scene description");
    const int linear_max_tris = d->linear_max_tris > 0 ? std::min(d->linear_max_tris, 4096) : 48;
    const bool use_bvh = d->n_tris > 0 && (d->n_bvh_nodes != 0 || n_gprim > 4096 || d->n_tris > linear_max_tris);
    if (!use_bvh && n_gprim > 4096)
        return fail(c, RT_ERR_UNSUPPORTED, "%lld primitives without triangles: the linear sphere walk takes at most 4096", n_gprim);
    std::vector<rt_bvh_node> bvh_nodes; std::vector<int32_t> bvh_order;
    bool refitted = false;
    std::vector<char> linear_tri;   // with a hierarchy: triangles that stay in the linear table instead (empty = none)
    if (use_bvh) {
        if (d->n_bvh_nodes > 0) {
            if (!d->bvh_nodes || !d->bvh_tri_index) return fail(c, RT_ERR_INVALID_ARG, "n_bvh_nodes > 0 with a NULL table");
            const rt_bvh_node *nn = static_cast<const rt_bvh_node *>(d->bvh_nodes);
            const std::string why = bvh::validate(d->n_bvh_nodes, nn, d->n_tris, d->bvh_tri_index, d->tri_verts);
            if (!why.empty()) return fail(c, RT_ERR_INVALID_ARG, "bad hierarchy: %s", why.c_str());
            bvh_nodes.assign(nn, nn + d->n_bvh_nodes);
            bvh_order.assign(d->bvh_tri_index, d->bvh_tri_index + d->n_tris);
        } else if ((d->flags & RT_SCENE_REFIT) && !c->topo_nodes.empty() && (int64_t)c->topo_linear.size() == d->n_tris &&
                   c->topo_mesh_begin == std::vector<int32_t>(d->mesh_tri_begin, d->mesh_tri_begin + d->n_meshes + 1) &&
                   c->topo_leaf == d->bvh_leaf_size && c->topo_flags == (d->flags & RT_SCENE_NO_LINEAR_SPLIT)) {
            // same meshes, moved vertices: the topology and leaf order of the hierarchy built last time, boxes refitted
            bvh_nodes = c->topo_nodes; bvh_order = c->topo_order; linear_tri = c->topo_linear;
            bvh::refit(bvh_nodes, bvh_order, d->tri_verts);
            refitted = true;
            // A topology built for other positions gets worse to walk the further the vertices move (boxes of
            // neighbours overlap more and more): its quality is the sum of the boxes' areas, which is what the
            // builder minimises.  Beyond 1.5 times the sum the hierarchy had when it was built, it is built anew.
            if (bvh::area_sum(bvh_nodes) > 1.5 * c->topo_area) { refitted = false; bvh_nodes.clear(); bvh_order.clear(); linear_tri.clear(); }
        }
        if (use_bvh && bvh_nodes.empty() && d->n_bvh_nodes <= 0) {
            // Triangles that span a sizeable part of the scene (walls, floors) would sit in leaves near
            // the root whose boxes nearly every ray enters, and each test there costs four divergent
            // 16-byte loads per lane; the linear table tests them from scalar registers instead.  Up
            // to 128 triangles whose box area is at least 1 % of the scene's stay out of the hierarchy.
            std::vector<double> area(d->n_tris, 0.0);
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int t = 0; t < d->n_tris; ++t) {
                const double *v = d->tri_verts + 9 * (size_t)t;
                double e[3];
                bool finite = true;
                for (int k = 0; k < 9; ++k) finite = finite && std::isfinite(v[k]);
                if (!finite) continue;
                for (int k = 0; k < 3; ++k) {
                    const double l = std::min({v[k], v[3 + k], v[6 + k]}), h = std::max({v[k], v[3 + k], v[6 + k]});
                    e[k] = h - l; lo[k] = std::min(lo[k], l); hi[k] = std::max(hi[k], h);
                }
                area[t] = e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
            }
            const double scene_area = (hi[0] - lo[0]) * (hi[1] - lo[1]) + (hi[1] - lo[1]) * (hi[2] - lo[2]) + (hi[2] - lo[2]) * (hi[0] - lo[0]);
            std::vector<int32_t> large;
            if (!(d->flags & RT_SCENE_NO_LINEAR_SPLIT))
                for (int t = 0; t < d->n_tris; ++t) if (area[t] >= 0.01 * scene_area) large.push_back(t);
            if (large.size() > 128) {
                std::sort(large.begin(), large.end(), [&](int32_t a, int32_t b) { return area[a] > area[b]; });
                large.resize(128);
            }
            linear_tri.assign(d->n_tris, 0);
            for (int32_t t : large) linear_tri[t] = 1;
            std::vector<int32_t> rest;
            rest.reserve(d->n_tris - large.size());
            for (int t = 0; t < d->n_tris; ++t) if (!linear_tri[t]) rest.push_back(t);
            if (rest.empty()) { rest.push_back(large.back()); linear_tri[large.back()] = 0; }   // keep the hierarchy non-empty
            bvh::Builder b;
            // leaf target: default 2 (measured with the wide walk at 1.18 M triangles, 32 spp: 1 / 2 / 3 / 4 / 6 -> 30.1 / 30.3 / 31.1 / 34.7 / 39.1 ms)
            b.run(d->n_tris, d->tri_verts, d->bvh_leaf_size > 0 ? d->bvh_leaf_size : 2, &rest, d->build_threads);
            bvh_nodes.swap(b.nodes); bvh_order.swap(b.order);
            c->topo_nodes = bvh_nodes; c->topo_order = bvh_order; c->topo_linear = linear_tri;
            c->topo_mesh_begin.assign(d->mesh_tri_begin, d->mesh_tri_begin + d->n_meshes + 1);
            c->topo_leaf = d->bvh_leaf_size; c->topo_flags = d->flags & RT_SCENE_NO_LINEAR_SPLIT;
            c->topo_area = bvh::area_sum(bvh_nodes);
        }
    }
    c->last_refit = refitted;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    // From here on the tables of the previous scene are being replaced (uploads may free and
    // reallocate them): until everything has arrived the context has NO scene, so that a failed
    // upload leaves later calls with RT_ERR_NO_SCENE instead of pointers into freed memory.
    c->has_scene = false;
    c->ray_w = c->ray_h = 0;

    const V3 cam = {d->cam_pos[0], d->cam_pos[1], d->cam_pos[2]};
    double scene_radius = norm(cam);

    // ---- triangles ---------------------------------------------------------------------------
    std::vector<int> tri_mesh(d->n_tris, 0);
    for (int m = 0; m < d->n_meshes; ++m)
        for (int t = d->mesh_tri_begin[m]; t < d->mesh_tri_begin[m + 1]; ++t) tri_mesh[t] = m;
    std::vector<TriRow> rows_all(d->n_tris);
    std::vector<TriAux> aux(d->n_tris);
    bool any_det = false;
    for (int t = 0; t < d->n_tris; ++t) {
        const double *v = d->tri_verts + 9 * (size_t)t;
        V3 a = {v[0], v[1], v[2]}, b = {v[3], v[4], v[5]}, cc = {v[6], v[7], v[8]};
        V3 ab = sub(b, a), ac = sub(cc, a), N = cross(ab, ac);
        double nl = norm(N);
        TriRow r{}; TriAux x{};
        const int m = tri_mesh[t];
        x.obj = d->mesh_obj[m]; x.face = t - d->mesh_tri_begin[m]; x.tag = 0u; x.pair_row = -1;
        r.gprim = d->n_spheres + t;
        // |det| = |d.N| <= |d||N|: with |d| <= 2 (longer rays use the exact code) a triangle with
        // |N| < 4e-7 can never pass upstream's |det| >= 1e-6 cut, nor can a non-finite one.
        const bool live = std::isfinite(nl) && nl >= 4e-7;
        if (live) {
            V3 e1 = cross(ac, N), e2 = cross(N, ab);
            double n2 = nl * nl;
            e1 = {e1.x / n2, e1.y / n2, e1.z / n2}; e2 = {e2.x / n2, e2.y / n2, e2.z / n2};
            r.nx = (float)N.x; r.ny = (float)N.y; r.nz = (float)N.z; r.kn = (float)(-dot(N, a));
            r.e1x = (float)e1.x; r.e1y = (float)e1.y; r.e1z = (float)e1.z; r.k1 = (float)(-dot(e1, a));
            r.e2x = (float)e2.x; r.e2y = (float)e2.y; r.e2z = (float)e2.z; r.k2 = (float)(-dot(e2, a));
            r.cb = (float)(norm(e1) + norm(e2));   // finished below, once the scene radius is known
            r.ct = up(1.01 * nl);
            const double noise = 16.0 * (double)kU * nl;
            r.det_lo = (float)std::max(0.0, 1e-6 - noise) * (1.0f - 4.0f * kU);
            x.det_hi = up(1e-6 + noise);
            if (r.det_lo > 0) any_det = true;
            x.nx = (float)(N.x / nl); x.ny = (float)(N.y / nl); x.nz = (float)(N.z / nl);
            scene_radius = std::max({scene_radius, norm(a), norm(b), norm(cc)});
        } else {
            // never a candidate; judged "undecided" if ever asked, so the exact code answers
            r.nx = r.ny = r.nz = 0; r.kn = 1; r.k1 = r.k2 = -1; r.cb = 0; r.ct = 0; r.det_lo = INFINITY;
            x.det_hi = INFINITY;
        }
        rows_all[t] = r; aux[t] = x;
    }

    // ---- spheres -----------------------------------------------------------------------------
    // (the scene radius also takes the small spheres; triangle rows are finished after this block)
    std::vector<double> radii(d->n_spheres);
    for (int s = 0; s < d->n_spheres; ++s) radii[s] = std::fabs(d->spheres[4 * (size_t)s + 3]);
    double median_r = 0;
    if (d->n_spheres >= 3) {
        std::vector<double> tmp = radii;
        std::nth_element(tmp.begin(), tmp.begin() + tmp.size() / 2, tmp.end());
        median_r = tmp[tmp.size() / 2];
    }
    std::vector<SphRow> sph_small, sph_all(d->n_spheres);
    std::vector<int32_t> big;
    for (int s = 0; s < d->n_spheres; ++s) {
        const double *p = d->spheres + 4 * (size_t)s;
        V3 ctr = {p[0], p[1], p[2]};
        double r = radii[s];
        // Huge spheres (a ground "plane" of radius 1e5 under a camera a few units away) lose all
        // float32 precision in |o-c|^2 - r^2; they are few, and are evaluated in double instead.
        double gap = std::fabs(norm(sub(cam, ctr)) - r);
        bool is_big = !(std::isfinite(r) && std::isfinite(norm1(ctr))) || (median_r > 0 && r > 64.0 * median_r) ||
                      r > 64.0 * std::max(gap, 1e-3 * r);
        SphRow row{};
        row.cx = (float)ctr.x; row.cy = (float)ctr.y; row.cz = (float)ctr.z;
        row.r2 = (float)(r * r); row.r = (float)r; row.cn = up(norm(ctr)); row.gprim = s;
        // how far the float32 centre lies from the true one (0 when it is representable); negative marks the huge
        // spheres in the by-id table (the search rows never hold them)
        // (a huge sphere's marker carries what hit_finish needs of it: -2^-49 (|c| + r), the scale of the root's own noise)
        row.qs = is_big ? -up(std::max(1.8e-15 * (norm(ctr) + r), 1e-30)) : up(norm(sub(ctr, V3{(double)row.cx, (double)row.cy, (double)row.cz})));
        sph_all[s] = row;
        if (is_big) big.push_back(s);
        else { sph_small.push_back(row); scene_radius = std::max(scene_radius, norm(ctr) + r); }
    }

    // barycentric bound constant: cb = 2.25 (|e1|+|e2|) |N| + 2 |N| / S_min, S_min = scene radius
    {
        const double smin = std::max(scene_radius, 1e-30);
        auto finish = [&](TriRow &r) {
            if (!(r.det_lo < INFINITY)) return;
            const double nl = (double)r.ct / 1.01;
            r.cb = up(1.01 * (2.25 * (double)r.cb * nl + 2.0 * nl / smin));
        };
        for (auto &r : rows_all) finish(r);
    }
    // Linear table: live rows in triangle order.  Two triangles of one mesh that share an edge and
    // form a parallelogram (d = a + c - b to 1e-11 of the triangle's smallest height - rectangles
    // typed in decimal are not exact in binary) share one row: the first triangle's functionals,
    // rebuilt for the vertex order that puts the shared edge at a-c, with both bound constants
    // widened by 1e-3, far more than the tolerance (see row_candidates).
    std::vector<TriRow> rows_live, rows_quad;
    // per parallelogram row: its first triangle, the axis its plane is perpendicular to (-1: none),
    // the plane's coordinate on that axis and |N| of the row's plane functional
    struct PairInfo { int tri; int axis; double c0, nl; V3 corner[4]; int box, code; };
    std::vector<PairInfo> pair_info;
    int n_live_tris = 0;
    {
        const bool merge_pairs = !(d->flags & RT_SCENE_NO_PAIR_ROWS);   // switch for A/B runs and tests
        auto in_linear = [&](int t) { return !use_bvh || (!linear_tri.empty() && linear_tri[t]); };
        std::vector<char> taken(d->n_tris, 0);
        auto vtx = [&](int t, int k) { const double *v = d->tri_verts + 9 * (size_t)t + 3 * k; return V3{v[0], v[1], v[2]}; };
        auto same = [](V3 p, V3 q) { return p.x == q.x && p.y == q.y && p.z == q.z; };
        for (int t = 0; t < d->n_tris; ++t) {
            if (taken[t] || !in_linear(t) || !(rows_all[t].det_lo < INFINITY)) continue;
            TriRow r = rows_all[t];
            ++n_live_tris;
            const int mesh_end = d->mesh_tri_begin[tri_mesh[t] + 1];
            for (int y = t + 1; merge_pairs && y < std::min(mesh_end, t + 65) && d->n_spheres + y < (1 << kQuadIdBits); ++y) {
                if (taken[y] || !in_linear(y) || !(rows_all[y].det_lo < INFINITY)) continue;
                // shared vertices (bitwise equal), and each triangle's remaining one
                int xs = -1, ys = -1, n_shared = 0;
                bool x_shared[3] = {false, false, false}, y_shared[3] = {false, false, false};
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
                    if (!x_shared[i] && !y_shared[j] && same(vtx(t, i), vtx(y, j))) { x_shared[i] = y_shared[j] = true; ++n_shared; }
                if (n_shared != 2) continue;
                for (int i = 0; i < 3; ++i) { if (!x_shared[i]) xs = i; if (!y_shared[i]) ys = i; }
                // rotate X so that its own vertex is b: (a, b, c) = (X[xs-1], X[xs], X[xs+1])
                const V3 a = vtx(t, (xs + 2) % 3), b = vtx(t, xs), cc = vtx(t, (xs + 1) % 3), dd = vtx(y, ys);
                const V3 ab = sub(b, a), ac = sub(cc, a), cb_ = sub(b, cc), N = cross(ab, ac);
                const double nl = norm(N), longest = std::max({norm(ab), norm(ac), norm(cb_)});
                const V3 gap = {a.x + cc.x - b.x - dd.x, a.y + cc.y - b.y - dd.y, a.z + cc.z - b.z - dd.z};
                if (!(norm(gap) <= 1e-11 * nl / longest)) continue;
                // functionals of the rotated triangle, as in the per-triangle rows above
                V3 e1 = cross(ac, N), e2 = cross(N, ab);
                const double n2 = nl * nl;
                e1 = {e1.x / n2, e1.y / n2, e1.z / n2}; e2 = {e2.x / n2, e2.y / n2, e2.z / n2};
                r.nx = (float)N.x; r.ny = (float)N.y; r.nz = (float)N.z; r.kn = (float)(-dot(N, a));
                r.e1x = (float)e1.x; r.e1y = (float)e1.y; r.e1z = (float)e1.z; r.k1 = (float)(-dot(e1, a));
                r.e2x = (float)e2.x; r.e2y = (float)e2.y; r.e2z = (float)e2.z; r.k2 = (float)(-dot(e2, a));
                const double smin = std::max(scene_radius, 1e-30);
                r.cb = up(1.001 * 1.01 * (2.25 * (norm(e1) + norm(e2)) * nl + 2.0 * nl / smin));
                r.ct = up(1.001 * 1.01 * nl);
                r.det_lo = std::min(rows_all[t].det_lo, rows_all[y].det_lo);
                r.gprim = kQuadFlag | ((d->n_spheres + y) << kQuadIdBits) | (d->n_spheres + t);
                taken[y] = 1;
                ++n_live_tris;
                aux[t].tag = (uint32_t)(d->n_spheres + t + 1); aux[y].tag = (uint32_t)(d->n_spheres + t + 1) | kTagSecond;   // ids < 8192
                aux[t].det_hi = aux[y].det_hi = std::max(aux[t].det_hi, aux[y].det_hi);
                int axis = -1;
                if (N.y == 0 && N.z == 0) axis = 0; else if (N.x == 0 && N.z == 0) axis = 1; else if (N.x == 0 && N.y == 0) axis = 2;
                const double c0 = axis == 0 ? a.x : (axis == 1 ? a.y : a.z);
                // the compact form reads e1/e2 without their component along the axis: it must be exactly zero
                if (axis >= 0 && ((axis == 0 && (r.e1x != 0 || r.e2x != 0)) || (axis == 1 && (r.e1y != 0 || r.e2y != 0)) ||
                                  (axis == 2 && (r.e1z != 0 || r.e2z != 0)))) axis = -1;
                if (d->flags & RT_SCENE_NO_COMPACT_ROWS) axis = -1;   // switch for A/B runs and tests
                pair_info.push_back({t, axis, c0, nl, {a, b, cc, dd}, -1, -1});
                break;
            }
            ((r.gprim & kQuadFlag) ? rows_quad : rows_live).push_back(r);
        }
    }
    // ---- boxes ---------------------------------------------------------------------------------
    // Parallelogram rows that are axis-aligned rectangles (to a tolerance: walls typed with a wobble of
    // 1e-6 count) are grouped into axis-aligned boxes; a box with at least four of its six faces in the
    // scene replaces those faces' rows in the first pass by one box test (see BoxRow).  Greedy: of all
    // boxes spanned by two rectangles, the one matching most rectangles first.
    std::vector<BoxRow> box_rows;
    std::vector<int32_t> box_face_id;
    std::vector<int> box_face_pair;   // [8 * box + code] -> index into pair_info, -1 = none
    double box_slack = 0, box_tilt = 0;
    if (!(d->flags & RT_SCENE_NO_BOX_ROWS) && !pair_info.empty()) {
        const double tol = 1e-5 * std::max(scene_radius, 1e-30);
        struct Rect { int k, axis; double lo[3], hi[3]; double area; };
        std::vector<Rect> rects;
        for (int k = 0; k < (int)pair_info.size(); ++k) {
            const PairInfo &pi = pair_info[k];
            Rect q; q.k = k; q.axis = -1;
            for (int ax = 0; ax < 3; ++ax) { q.lo[ax] = INFINITY; q.hi[ax] = -INFINITY; }
            for (const V3 &p : pi.corner) {
                const double c[3] = {p.x, p.y, p.z};
                for (int ax = 0; ax < 3; ++ax) { q.lo[ax] = std::min(q.lo[ax], c[ax]); q.hi[ax] = std::max(q.hi[ax], c[ax]); }
            }
            int thin = -1, n_thin = 0;
            for (int ax = 0; ax < 3; ++ax) {
                if (q.hi[ax] - q.lo[ax] <= tol) { thin = ax; ++n_thin; }
                else if (!(q.hi[ax] - q.lo[ax] > 100.0 * tol)) n_thin = 5;   // neither flat nor clearly extended (or not finite)
            }
            if (n_thin != 1) continue;
            // every corner of the bounding rectangle is taken by one corner of the parallelogram
            const int b = thin == 0 ? 1 : 0, cc = thin == 2 ? 1 : 2;
            int seen = 0;
            for (const V3 &p : pi.corner) {
                const double c[3] = {p.x, p.y, p.z};
                const bool bl = std::fabs(c[b] - q.lo[b]) <= tol, bh = std::fabs(c[b] - q.hi[b]) <= tol;
                const bool cl = std::fabs(c[cc] - q.lo[cc]) <= tol, ch = std::fabs(c[cc] - q.hi[cc]) <= tol;
                if ((bl || bh) && (cl || ch)) seen |= 1 << ((bh ? 1 : 0) | (ch ? 2 : 0));
            }
            if (seen != 15) continue;
            q.axis = thin;
            q.area = (q.hi[b] - q.lo[b]) * (q.hi[cc] - q.lo[cc]);
            rects.push_back(q);
        }
        if (rects.size() > 128) {   // the search below is cubic: the largest rectangles only
            std::sort(rects.begin(), rects.end(), [](const Rect &x, const Rect &y) { return x.area > y.area; });
            rects.resize(128);
        }
        struct Box { double lo[3], hi[3]; };
        auto near_ = [&](double x, double y) { return std::fabs(x - y) <= tol; };
        auto plane_of = [](const Rect &q) { return 0.5 * (q.lo[q.axis] + q.hi[q.axis]); };
        // which face of `bx` rectangle q is (-1: none)
        auto face_of = [&](const Box &bx, const Rect &q) {
            for (int ax = 0; ax < 3; ++ax)
                if (ax != q.axis && !(near_(q.lo[ax], bx.lo[ax]) && near_(q.hi[ax], bx.hi[ax]))) return -1;
            if (near_(plane_of(q), bx.lo[q.axis])) return 2 * q.axis;
            if (near_(plane_of(q), bx.hi[q.axis])) return 2 * q.axis + 1;
            return -1;
        };
        std::vector<char> used(rects.size(), 0);
        while ((int)box_rows.size() < kMaxBoxes) {
            int best_count = 0; Box best{};
            for (size_t i = 0; i < rects.size(); ++i) {
                if (used[i]) continue;
                for (size_t j = i + 1; j < rects.size(); ++j) {
                    if (used[j]) continue;
                    const Rect &p = rects[i], &q = rects[j];
                    Box bx;
                    if (p.axis == q.axis) {   // opposite faces
                        const int a = p.axis;
                        bool same = true;
                        for (int ax = 0; ax < 3; ++ax) if (ax != a) same = same && near_(p.lo[ax], q.lo[ax]) && near_(p.hi[ax], q.hi[ax]);
                        if (!same) continue;
                        for (int ax = 0; ax < 3; ++ax) { bx.lo[ax] = p.lo[ax]; bx.hi[ax] = p.hi[ax]; }
                        bx.lo[a] = std::min(plane_of(p), plane_of(q)); bx.hi[a] = std::max(plane_of(p), plane_of(q));
                    } else {                  // faces that share an edge
                        const int a1 = p.axis, a2 = q.axis, a3 = 3 - a1 - a2;
                        if (!(near_(p.lo[a3], q.lo[a3]) && near_(p.hi[a3], q.hi[a3]))) continue;
                        if (!(near_(plane_of(p), q.lo[a1]) || near_(plane_of(p), q.hi[a1]))) continue;
                        if (!(near_(plane_of(q), p.lo[a2]) || near_(plane_of(q), p.hi[a2]))) continue;
                        bx.lo[a1] = q.lo[a1]; bx.hi[a1] = q.hi[a1];
                        bx.lo[a2] = p.lo[a2]; bx.hi[a2] = p.hi[a2];
                        bx.lo[a3] = p.lo[a3]; bx.hi[a3] = p.hi[a3];
                    }
                    bool thick = true;
                    for (int ax = 0; ax < 3; ++ax) thick = thick && (bx.hi[ax] - bx.lo[ax] > 100.0 * tol);
                    if (!thick) continue;
                    int faces = 0, count = 0;
                    for (size_t m = 0; m < rects.size(); ++m) {
                        if (used[m]) continue;
                        const int f = face_of(bx, rects[m]);
                        if (f >= 0 && !(faces & (1 << f))) { faces |= 1 << f; ++count; }
                    }
                    if (count > best_count) { best_count = count; best = bx; }
                }
            }
            if (best_count < 4) break;
            const int bi = (int)box_rows.size();
            int face_rect[6] = {-1, -1, -1, -1, -1, -1};
            for (size_t m = 0; m < rects.size(); ++m) {
                if (used[m]) continue;
                const int f = face_of(best, rects[m]);
                if (f >= 0 && face_rect[f] < 0) { face_rect[f] = (int)m; used[m] = 1; }
            }
            // the ideal planes: a present face's own plane, else the extent of the others
            float lo[3], hi[3];
            for (int ax = 0; ax < 3; ++ax) {
                lo[ax] = (float)(face_rect[2 * ax] >= 0 ? plane_of(rects[face_rect[2 * ax]]) : best.lo[ax]);
                hi[ax] = (float)(face_rect[2 * ax + 1] >= 0 ? plane_of(rects[face_rect[2 * ax + 1]]) : best.hi[ax]);
            }
            BoxRow br{};
            br.lox = lo[0]; br.loy = lo[1]; br.loz = lo[2]; br.hix = hi[0]; br.hiy = hi[1]; br.hiz = hi[2];
            br.base = (uint32_t)(n_gprim + 1 + 8 * bi);
            for (int f = 0; f < 8; ++f) { box_face_id.push_back(-1); box_face_pair.push_back(-1); }
            for (int f = 0; f < 6; ++f) {
                if (face_rect[f] < 0) continue;
                const Rect &q = rects[face_rect[f]];
                PairInfo &pi = pair_info[q.k];
                // how far the true face is from the ideal one, per coordinate
                for (const V3 &p : pi.corner) {
                    const double c[3] = {p.x, p.y, p.z};
                    for (int ax = 0; ax < 3; ++ax) {
                        const double dl = std::fabs(c[ax] - (double)lo[ax]), dh = std::fabs(c[ax] - (double)hi[ax]);
                        const double dev = ax == q.axis ? ((f & 1) ? dh : dl) : std::min(dl, dh);
                        box_slack = std::max(box_slack, dev);
                    }
                }
                {   // how far the true face's normal leans away from its axis (sine of the angle)
                    const V3 N = cross(sub(pi.corner[1], pi.corner[0]), sub(pi.corner[2], pi.corner[0]));
                    const double n[3] = {N.x, N.y, N.z};
                    const int o1 = (q.axis + 1) % 3, o2 = (q.axis + 2) % 3;
                    box_tilt = std::max(box_tilt, std::sqrt(n[o1] * n[o1] + n[o2] * n[o2]) / norm(N));
                }
                br.present |= 1u << f;
                pi.box = bi; pi.code = f;
                const int id1 = d->n_spheres + pi.tri;
                box_face_id[8 * bi + f] = id1;
                box_face_pair[8 * bi + f] = q.k;
            }
            box_rows.push_back(br);
        }
        // the origin tags of the faces' triangles (both triangles of a row carry the row's box face)
        for (int t = 0; t < d->n_tris; ++t) {
            const uint32_t id1p = aux[t].tag & kTagIdMask;
            if (id1p == 0) continue;
            for (size_t f = 0; f < box_face_id.size(); ++f)
                if (box_face_id[f] >= 0 && (uint32_t)(box_face_id[f] + 1) == id1p) aux[t].tag |= (uint32_t)(f + 1) << kTagBoxShift;
        }
    }
    // Parallelogram rows in general position first, then those in axis-perpendicular planes (by axis);
    // the latter also get their compact form.  At most 1022 live ones per axis (the counts share one word, and an axis is padded to an even count).
    std::vector<SlabRow> rows_slab;
    int n_quad_general = 0;
    uint32_t slab_counts = 0;
    {
        int per_axis[3] = {0, 0, 0};
        // order: general position, box faces (tested through their boxes), compact rows by axis
        for (auto &pi : pair_info) { if (pi.box >= 0) pi.axis = -2; if (pi.axis >= 0 && ++per_axis[pi.axis] > 1022) pi.axis = -1; }
        std::vector<int> order;
        for (int pass : {-1, -2, 0, 1, 2})
            for (int k = 0; k < (int)pair_info.size(); ++k) if (pair_info[k].axis == pass) order.push_back(k);
        std::vector<TriRow> sorted(rows_quad.size());
        int counts[3] = {0, 0, 0};
        // the dead compact row: vq = -inf, so it is never inside; no origin tag equals its idf
        SlabRow dead_slab{}; dead_slab.k2h = -INFINITY; dead_slab.idf = -5.0f; dead_slab.gprim = kQuadFlag;
        int last_axis = -1;
        auto close_axis = [&](int axis) {   // the compact-row loop takes two rows per trip: every axis holds an even number
            if (axis >= 0 && counts[axis] % 2) { rows_slab.push_back(dead_slab); ++counts[axis]; }
        };
        for (int k = 0; k < (int)order.size(); ++k) {
            const PairInfo &pi = pair_info[order[k]];
            if (pi.axis != last_axis) { close_axis(last_axis); last_axis = pi.axis; }
            const TriRow &q = rows_quad[order[k]];
            sorted[k] = q;
            aux[pi.tri].pair_row = k;
            if (pi.axis == -1) { ++n_quad_general; continue; }
            if (pi.axis == -2) continue;
            ++counts[pi.axis];
            SlabRow sr{};
            sr.c0 = (float)pi.c0;
            const float e1[3] = {q.e1x, q.e1y, q.e1z}, e2[3] = {q.e2x, q.e2y, q.e2z};
            const int b = pi.axis == 0 ? 1 : 0, cc = pi.axis == 2 ? 1 : 2;
            // centred coordinates: vq = v - 1/2, sq = u + v - 1/2 (see SlabRow)
            sr.e2b = e2[b]; sr.e2c = e2[cc]; sr.k2h = (float)((double)q.k2 - 0.5);
            sr.esb = (float)((double)e1[b] + (double)e2[b]); sr.esc = (float)((double)e1[cc] + (double)e2[cc]);
            sr.ksh = (float)((double)q.k1 + (double)q.k2 - 0.5);
            // bound constants for a unit normal: the general row's divided by |N|; the barycentric one
            // also takes the rounding of the two shifted constants and of 1/2 - max(|vq|, |sq|), about
            // 1.5 u in all: the depth-error scale is at least 2 u S, S >= scene radius, so 2 / S_min covers it
            sr.cb = up(1.000001 * ((double)q.cb / pi.nl + 2.0 / std::max(scene_radius, 1e-30)));
            sr.ct = up(1.000001 * (double)q.ct / pi.nl);
            sr.id1 = q.gprim & ((1 << kQuadIdBits) - 1);
            sr.idf = (float)(sr.id1 + 1);
            sr.gprim = q.gprim;
            rows_slab.push_back(sr);
        }
        close_axis(last_axis);
        rows_quad.swap(sorted);
        // (the sort above moved the rows: pair_info[k] now sits at rows_quad[aux[pair_info[k].tri].pair_row])
        slab_counts = (uint32_t)counts[0] | ((uint32_t)counts[1] << 10) | ((uint32_t)counts[2] << 20);
        if (!rows_slab.empty()) rows_slab.push_back(dead_slab);   // the row the loops may read ahead into
    }
    // ---- sharper plane constants for the strict depth acceptance (see judge_tri) -------------------
    // Per triangle (by id) and per parallelogram row (by row index): what the rounding of the row's plane to
    // float32 and the order of its multiply-adds can cost, as coefficients of |o_x|, |o_y|, |o_z| and a constant.
    std::vector<float> tri_tight(4 * ((size_t)d->n_tris + pair_info.size()), 0.0f);
    {
        const double h = (double)kU * 1.01;   // one rounding: at most u = 2^-24 of the result
        auto put = [&](size_t slot, V3 N, double kN, const TriRow &r, double extra) {
            const double nl = norm(N);
            if (!std::isfinite(nl) || !(r.det_lo < INFINITY)) return;
            const double r1 = r.nx != 0 ? 1 : 0, r2 = r.ny != 0 ? 1 : 0, tilt = extra * nl;
            float *K = tri_tight.data() + 4 * slot;
            K[0] = up(std::fabs(N.x - (double)r.nx) + tilt);
            K[1] = up(h * std::fabs((double)r.ny) * r1 + std::fabs(N.y - (double)r.ny) + tilt);
            K[2] = up(h * std::fabs((double)r.nz) * (r1 + r2) + std::fabs(N.z - (double)r.nz) + tilt);
            K[3] = up(std::fabs(kN - (double)r.kn) + tilt * scene_radius);
        };
        for (int t = 0; t < d->n_tris; ++t) {
            const double *v = d->tri_verts + 9 * (size_t)t;
            const V3 a = {v[0], v[1], v[2]}, b = {v[3], v[4], v[5]}, cc = {v[6], v[7], v[8]};
            const V3 N = cross(sub(b, a), sub(cc, a));
            put((size_t)t, N, -dot(N, a), rows_all[t], 0.0);
        }
        // a parallelogram row stands for two triangles whose planes may differ by the 1e-11 the pairing tolerates
        for (const PairInfo &pi : pair_info) {
            const V3 N = cross(sub(pi.corner[1], pi.corner[0]), sub(pi.corner[2], pi.corner[0]));
            const int row = aux[pi.tri].pair_row;
            put((size_t)d->n_tris + (size_t)row, N, -dot(N, pi.corner[0]), rows_quad[row], 1e-10);
        }
    }
    // ---- materials, environment ---------------------------------------------------------------
    std::vector<MatRow> mats(d->n_objects);
    for (int i = 0; i < d->n_objects; ++i) {
        const double *m = d->materials + 7 * (size_t)i;
        double r0 = (1 - m[6]) / (1 + m[6]);
        mats[i] = {(float)m[0], (float)m[1], (float)m[2], (float)m[3], (float)m[4], (float)m[5], (float)m[6], (float)(r0 * r0)};
    }
    std::vector<uint32_t> sky;
    const bool has_sky = d->sky_rgb && d->sky_w > 0 && d->sky_h > 0;
    if (has_sky) {
        sky.resize((size_t)d->sky_w * d->sky_h);
        for (size_t i = 0; i < sky.size(); ++i)
            sky[i] = (uint32_t)d->sky_rgb[3 * i] | ((uint32_t)d->sky_rgb[3 * i + 1] << 8) | ((uint32_t)d->sky_rgb[3 * i + 2] << 16);
    }

    // ---- upload --------------------------------------------------------------------------------
    std::vector<double> tri64(d->tri_verts, d->tri_verts + 9 * (size_t)d->n_tris);
    std::vector<double> sph64(d->spheres, d->spheres + 4 * (size_t)d->n_spheres);
    {
        // radius**2 the way upstream's interpreter computes it: pow() of this machine's C library, called through a
        // pointer the compiler cannot see through (it would turn pow(r, 2.0) into r * r, which is not always the same double)
        double (*volatile lib_pow)(double, double) = static_cast<double (*)(double, double)>(&std::pow);
        for (int s = 0; s < d->n_spheres; ++s) {
            const double r = d->spheres[4 * (size_t)s + 3];
#ifdef RT_LIBM_TABLES_GENERATED
            sph64.push_back(lib_pow(r, 2.0));
#else
            sph64.push_back(r * r);
            (void)lib_pow;
#endif
        }
    }
    std::vector<double> aabb64(d->mesh_aabb, d->mesh_aabb + 6 * (size_t)d->n_meshes);
    std::vector<int32_t> mtb(d->n_meshes + 1, 0);
    for (int m = 0; m <= d->n_meshes && d->n_meshes; ++m) mtb[m] = d->mesh_tri_begin[m];
    std::vector<int32_t> okind(d->obj_kind, d->obj_kind + d->n_objects), oindex(d->obj_index, d->obj_index + d->n_objects);
    std::vector<int32_t> sobj(d->sphere_obj, d->sphere_obj + d->n_spheres);
    // pad the live rows with dead ones (det_lo = +inf) to an even count, plus two to read ahead
    int n_live = n_live_tris;
    std::vector<TriRow> rows_leaf;
    if (use_bvh) {   // the walk reads rows in leaf order (dead triangles included: they kill themselves)
        rows_leaf.resize(bvh_order.size());
        for (size_t k = 0; k < bvh_order.size(); ++k) rows_leaf[k] = rows_all[bvh_order[k]];
    }
    const bool any_tri_rows = !rows_live.empty(), any_quad_rows = !rows_quad.empty();
    for (auto *tab : {&rows_live, &rows_quad}) {
        if (tab->empty()) continue;
        TriRow dead{}; dead.kn = 1; dead.k1 = dead.k2 = -1; dead.det_lo = INFINITY; dead.gprim = -2;
        while (tab->size() % 2) tab->push_back(dead);
        for (int k = 0; k < 2; ++k) tab->push_back(dead);
    }
    // box tables: the faces' parallelogram rows by 8 * box + code (dead where there is no face), and one spare box row to read ahead into
    std::vector<TriRow> box_face_rows(box_face_id.size());
    for (size_t f = 0; f < box_face_rows.size(); ++f) {
        TriRow dead{}; dead.kn = 1; dead.k1 = dead.k2 = -1; dead.det_lo = INFINITY; dead.gprim = -2;
        box_face_rows[f] = box_face_pair[f] >= 0 ? rows_quad[aux[pair_info[box_face_pair[f]].tri].pair_row] : dead;
    }
    const int n_boxes = (int)box_rows.size();
    if (n_boxes > 0) {
        // the box loop takes two per trip: an odd count is padded with a box far outside every scene the
        // float32 code handles (it has no faces; its ids lie above the real boxes'), plus one row to read ahead into
        BoxRow far_box{};
        far_box.lox = far_box.loy = far_box.loz = 3.0e18f; far_box.hix = far_box.hiy = far_box.hiz = 3.1e18f;
        far_box.present = 0; far_box.base = (uint32_t)(n_gprim + 1 + 8 * n_boxes);
        if (n_boxes % 2) box_rows.push_back(far_box);
        box_rows.push_back(far_box);
    }
    HIP_TRY(c, c->box_rows.upload(box_rows)); HIP_TRY(c, c->box_face_id.upload(box_face_id)); HIP_TRY(c, c->box_face_rows.upload(box_face_rows));
    HIP_TRY(c, c->tri_rows.upload(rows_live)); HIP_TRY(c, c->quad_rows.upload(rows_quad)); HIP_TRY(c, c->slab_rows.upload(rows_slab)); HIP_TRY(c, c->leaf_rows.upload(rows_leaf)); HIP_TRY(c, c->tri_by_id.upload(rows_all)); HIP_TRY(c, c->tri_aux.upload(aux));
    HIP_TRY(c, c->tri_tight.upload(tri_tight));
    const int n_sph_live = (int)sph_small.size();
    // the search loop's own table: eight rows per trip and four rows read ahead, so dead rows (never candidates) fill it
    // up to a multiple of eight and four more follow
    std::vector<SphLoopRow> sph_loop(n_sph_live);
    std::vector<int32_t> sph_loop_id(n_sph_live);
    for (int i = 0; i < n_sph_live; ++i) {
        const SphRow &q = sph_small[i];
        const double *p = d->spheres + 4 * (size_t)q.gprim;
        SphLoopRow L{};
        L.cx = q.cx; L.cy = q.cy; L.cz = q.cz;
        L.ks = (float)(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] - p[3] * p[3]);
        sph_loop[i] = L; sph_loop_id[i] = q.gprim;
    }
    {
        SphLoopRow dead{}; dead.ks = INFINITY;
        while (sph_loop.size() % 8) { sph_loop.push_back(dead); sph_loop_id.push_back(-2); }
        for (int k = 0; k < 4; ++k) { sph_loop.push_back(dead); sph_loop_id.push_back(-2); }
    }
    HIP_TRY(c, c->sph_loop.upload(sph_loop)); HIP_TRY(c, c->sph_loop_id.upload(sph_loop_id));
    HIP_TRY(c, c->sph_rows.upload(sph_small)); HIP_TRY(c, c->sph_by_id.upload(sph_all)); HIP_TRY(c, c->big_sph.upload(big));
    HIP_TRY(c, c->sph_obj.upload(sobj)); HIP_TRY(c, c->materials.upload(mats));
    HIP_TRY(c, c->tri64.upload(tri64)); HIP_TRY(c, c->sph64.upload(sph64)); HIP_TRY(c, c->aabb64.upload(aabb64));
    HIP_TRY(c, c->mesh_tri_begin.upload(mtb)); HIP_TRY(c, c->obj_kind.upload(okind)); HIP_TRY(c, c->obj_index.upload(oindex));
    if (has_sky) HIP_TRY(c, c->sky.upload(sky));
    if (use_bvh) {
        // Device form: the wide hierarchy the kernels walk (four children per node, one copy for every ray
        // direction), and the threaded two-child hierarchy as the builder left it, for the walk of last resort.
        std::vector<WideNode> wide;
        int32_t need = 0;
        if (!bvh::collapse_wide(bvh_nodes, wide, &need))
            return fail(c, RT_ERR_UNSUPPORTED, "hierarchy of %zu nodes: the wide form holds at most 2^24 nodes (use larger leaves)", bvh_nodes.size());
        c->walk_stack_need = need;
        HIP_TRY(c, c->bvh.upload(bvh_nodes));
        HIP_TRY(c, c->bvh_wide.upload(wide));
    }

    DevScene &S = c->S;
    S = DevScene{};
    S.tri_rows = c->tri_rows.as<TriRow>(); S.n_tri_rows = n_live; S.n_tri_rows_padded = any_tri_rows ? (int)rows_live.size() - 2 : 0;
    S.quad_rows = c->quad_rows.as<TriRow>(); S.n_quad_rows_padded = any_quad_rows ? (int)rows_quad.size() - 2 : 0;
    S.slab_rows = c->slab_rows.as<SlabRow>(); S.n_quad_general = n_quad_general; S.slab_counts = slab_counts;
    S.box_rows = c->box_rows.as<BoxRow>(); S.box_face_id = c->box_face_id.as<int32_t>(); S.box_face_rows = c->box_face_rows.as<TriRow>();
    S.n_boxes = n_boxes; S.box_slack = up(1.01 * box_slack); S.box_tilt = up(1.01 * box_tilt + 1e-7);
    S.sph_rows = c->sph_rows.as<SphRow>(); S.n_sph_rows = n_sph_live; S.sph_loop = c->sph_loop.as<SphLoopRow>(); S.sph_loop_id = c->sph_loop_id.as<int32_t>();
    S.big_sph = c->big_sph.as<int32_t>(); S.n_big_sph = (int)big.size();
    S.tri_by_id = c->tri_by_id.as<TriRow>(); S.tri_aux = c->tri_aux.as<TriAux>();
    S.sph_by_id = c->sph_by_id.as<SphRow>(); S.sph_obj = c->sph_obj.as<int32_t>();
    S.materials = c->materials.as<MatRow>();
    S.tri_tight = c->tri_tight.as<float>();
    S.leaf_rows = c->leaf_rows.as<TriRow>();
    S.bvh_nodes = use_bvh ? c->bvh.as<BvhNode>() : nullptr; S.n_bvh_nodes = use_bvh ? (int32_t)bvh_nodes.size() : 0;
    S.wide_nodes = use_bvh ? c->bvh_wide.as<WideNode>() : nullptr;
    S.walk_stack = (d->flags & RT_SCENE_TINY_WALK_STACK) ? 6 : kWalkStack;   // (the flag: tests of the walk of last resort)
    ColdScene cold{};
    cold.tri64 = c->tri64.as<double>(); cold.sph64 = c->sph64.as<double>(); cold.aabb64 = c->aabb64.as<double>();
    cold.mesh_tri_begin = c->mesh_tri_begin.as<int32_t>(); cold.obj_kind = c->obj_kind.as<int32_t>(); cold.obj_index = c->obj_index.as<int32_t>();
    cold.n_objects = d->n_objects; cold.n_tris = d->n_tris; cold.n_meshes = d->n_meshes;
    for (int k = 0; k < 3; ++k) {
        cold.cam64[k] = d->cam_pos[k]; cold.cam64[3 + k] = d->cam_right[k]; cold.cam64[6 + k] = d->cam_up[k]; cold.cam64[9 + k] = d->cam_forward[k];
        S.cam_pos[k] = (float)d->cam_pos[k];
        S.cam_res[k] = (float)(d->cam_pos[k] - (double)(float)d->cam_pos[k]);
    }
    S.cam_rko = (S.cam_res[0] != 0.0f || S.cam_res[1] != 0.0f || S.cam_res[2] != 0.0f) ? 1.01f * kU : 0.0f;
    cold.cam64[12] = d->cam_d; cold.cam64[13] = d->cam_dof_strength; cold.cam64[14] = d->cam_dof_dist;
    HIP_TRY(c, c->cold.upload(std::vector<ColdScene>{cold}));
    S.cold = c->cold.as<ColdScene>();
    S.n_spheres = d->n_spheres;
    S.n_prims = (int32_t)n_gprim;
    S.scene_radius = up(scene_radius * 1.000001);
    int bits = 5;
    while ((1ll << bits) < n_gprim + 1 + 8 * (n_boxes + 1) && bits < 23) ++bits;   // only the linear walk's 32-bit keys use it (box faces have ids too)
    S.key_mask = (1u << bits) - 1u;
    S.dof_on = !(d->cam_dof_dist == 0 || d->cam_dof_strength == 0);   // camera.py:41
    S.sky = has_sky ? c->sky.as<uint32_t>() : nullptr; S.sky_w = has_sky ? d->sky_w : 0; S.sky_h = has_sky ? d->sky_h : 0;
    S.sky_scale = (float)d->sky_scale;
    c->tri_mode = use_bvh ? kTriBvh : (n_live > 0 ? kTriBrute : kTriNone); c->has_sph = d->n_spheres > 0;
    c->ray_w = c->ray_h = 0;   // camera changed: the ray table is stale
    c->has_scene = true;
    return RT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// closest-hit entry
// ---------------------------------------------------------------------------------------------
namespace {

// (Re)build the camera-ray table for width x height on `stream` if the cached one does not match.
int ensure_ray_table(rt_ctx *c, int width, int height, hipStream_t stream) {
    if (c->ray_w == width && c->ray_h == height) return RT_OK;
    const size_t bytes = (size_t)width * height * 4 * sizeof(float);   // directions: four floats; residues: four doubles per pixel
    if (bytes > c->ray_dir.bytes || 2 * bytes > c->ray_res.bytes) {
        HIP_TRY(c, hipDeviceSynchronize());   // earlier launches may still read the old tables
        HIP_TRY(c, c->ray_dir.ensure(bytes)); HIP_TRY(c, c->ray_res.ensure(2 * bytes));
    }
    hipLaunchKernelGGL(k_raygen, dim3((unsigned)(((size_t)width * height + 255) / 256)), dim3(256), 0, stream, c->S, width, height,
                       c->ray_dir.as<float>(), c->ray_res.as<float>());
    HIP_TRY(c, hipGetLastError());
    c->ray_w = width; c->ray_h = height;
    return RT_OK;
}

struct IsectArgs {
    long long n;
    const double *org, *dir; const int32_t *origin;
    int32_t *obj, *prim, *gprim, *how; double *t, *normal, *pos;
    unsigned long long *stats;
};

template <int TRI, bool HAS_SPH>
__global__ __launch_bounds__(256) void k_intersect(const DevScene S_arg, const IsectArgs A) {
    SceneRef S = *kernarg_scene();   // the scene argument, read in place (scalar loads)
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const double ox = A.org[3 * i], oy = A.org[3 * i + 1], oz = A.org[3 * i + 2];
    const double dx = A.dir[3 * i], dy = A.dir[3 * i + 1], dz = A.dir[3 * i + 2];
    const int32_t origin = A.origin ? A.origin[i] : -1;
    const float fox = (float)ox, foy = (float)oy, foz = (float)oz, fdx = (float)dx, fdy = (float)dy, fdz = (float)dz;
    // The float32 filter is only valid for the ray it is given: rays that are not exactly
    // representable in float32 are answered by the exact code on the doubles themselves.
    const bool f32_ray = (double)fox == ox && (double)foy == oy && (double)foz == oz && (double)fdx == dx &&
                         (double)fdy == dy && (double)fdz == dz;
    ExactHit e; e.hit = false; e.prim = 0; e.t = 0; e.n = d3(0, 0, 0); e.pos = e.n;
    Hit h; h.obj = -1; h.prim = -1; h.gprim = -1; h.how = 2; h.refined = false; h.bvh_nodes = 0; h.bvh_tris = 0;
    bool did_exact = false;
    if (f32_ray) {
        Ray r = make_ray(S, fox, foy, foz, fdx, fdy, fdz, origin, false, origin_tag(S, origin));
        h = closest_hit<TRI, HAS_SPH, true>(S, r, &e, no_residue());
        did_exact = h.how == 2;
    } else {
        const D3 o64 = d3(ox, oy, oz), d64 = d3(dx, dy, dz);
        ScanHit sh = exact_scan(S, o64, d64, origin);
        did_exact = true;
        if (sh.gprim >= 0) e = exact_single(S, sh.gprim, o64, d64, origin);
        if (e.hit) {
            h.gprim = sh.gprim;
            if (sh.gprim < S.n_spheres) { h.obj = S.sph_obj[sh.gprim]; h.prim = e.prim; }
            else { const TriAux X = S.tri_aux[sh.gprim - S.n_spheres]; h.obj = X.obj; h.prim = X.face; }
        }
    }
    const bool have64 = e.hit;
    A.obj[i] = h.obj; A.prim[i] = h.prim;
    if (A.gprim) A.gprim[i] = h.gprim;
    if (A.how) A.how[i] = h.how;
    const bool hit = h.obj >= 0;
    A.t[i] = !hit ? (double)INFINITY : (have64 ? e.t : (double)h.t);
    if (A.normal) {
        A.normal[3 * i] = !hit ? 0.0 : (have64 ? e.n.x : (double)h.nx);
        A.normal[3 * i + 1] = !hit ? 0.0 : (have64 ? e.n.y : (double)h.ny);
        A.normal[3 * i + 2] = !hit ? 0.0 : (have64 ? e.n.z : (double)h.nz);
    }
    if (A.pos) {
        A.pos[3 * i] = !hit ? 0.0 : (have64 ? e.pos.x : (double)h.px);
        A.pos[3 * i + 1] = !hit ? 0.0 : (have64 ? e.pos.y : (double)h.py);
        A.pos[3 * i + 2] = !hit ? 0.0 : (have64 ? e.pos.z : (double)h.pz);
    }
    if (did_exact) atomicAdd(A.stats + kStatExact, 1ull);
    if (h.refined) atomicAdd(A.stats + kStatRefine, 1ull);
}

template <class F> void dispatch_variant(int tri, bool sph, F &&f) {
    using std::integral_constant;
    if (tri == kTriBvh) { if (sph) f(integral_constant<int, kTriBvh>{}, std::true_type{}); else f(integral_constant<int, kTriBvh>{}, std::false_type{}); }
    else if (tri == kTriBrute) { if (sph) f(integral_constant<int, kTriBrute>{}, std::true_type{}); else f(integral_constant<int, kTriBrute>{}, std::false_type{}); }
    else f(integral_constant<int, kTriNone>{}, std::true_type{});
}

}  // namespace

extern "C" {

int rt_intersect_batch(rt_ctx *c, int64_t n, const double *org, const double *dir, const int32_t *origin_prim,
                       int32_t *obj, int32_t *prim, int32_t *gprim, int32_t *how, double *t, double *normal, double *pos) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && (!org || !dir || !obj || !prim || !t))) return fail(c, RT_ERR_INVALID_ARG, "rt_intersect_batch: bad arguments");
    if (!c->has_scene) return fail(c, RT_ERR_NO_SCENE, "rt_intersect_batch: no scene set");
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // Staging lives in two context-owned buffers that only ever grow (a closest-hit query for one ray -
    // Scene.intersect(ray) - used to pay ten allocations and frees): inputs org | dir | origin, outputs
    // t | normal | pos | obj | prim | gprim | how, each array aligned to 16 bytes.
    const size_t N = (size_t)n, n3 = 3 * N * sizeof(double);
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t in_org = 0, in_dir = al(n3), in_origin = in_dir + al(n3), in_bytes = in_origin + al(N * 4);
    const size_t o_t = 0, o_n = al(N * 8), o_p = o_n + al(n3), o_obj = o_p + al(n3), o_prim = o_obj + al(N * 4),
                 o_gprim = o_prim + al(N * 4), o_how = o_gprim + al(N * 4), out_bytes = o_how + al(N * 4);
    if (in_bytes > c->isect_in.bytes || out_bytes > c->isect_out.bytes) {
        HIP_TRY(c, hipDeviceSynchronize());
        HIP_TRY(c, c->isect_in.ensure(in_bytes + in_bytes / 2)); HIP_TRY(c, c->isect_out.ensure(out_bytes + out_bytes / 2));
    }
    char *bi = c->isect_in.as<char>(), *bo = c->isect_out.as<char>();
    HIP_TRY(c, hipMemcpy(bi + in_org, org, n3, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(bi + in_dir, dir, n3, hipMemcpyHostToDevice));
    if (origin_prim) HIP_TRY(c, hipMemcpy(bi + in_origin, origin_prim, N * 4, hipMemcpyHostToDevice));
    IsectArgs A{};
    A.n = n; A.org = (const double *)(bi + in_org); A.dir = (const double *)(bi + in_dir);
    A.origin = origin_prim ? (const int32_t *)(bi + in_origin) : nullptr;
    A.obj = (int32_t *)(bo + o_obj); A.prim = (int32_t *)(bo + o_prim); A.gprim = (int32_t *)(bo + o_gprim); A.how = (int32_t *)(bo + o_how);
    A.t = (double *)(bo + o_t); A.normal = (double *)(bo + o_n); A.pos = (double *)(bo + o_p); A.stats = c->stats.as<unsigned long long>();
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const DevScene S = c->S;
    dispatch_variant(c->tri_mode, c->has_sph, [&](auto T, auto P) {
        hipLaunchKernelGGL((k_intersect<decltype(T)::value, decltype(P)::value>), dim3(blocks), dim3(256), 0, 0, S, A);
    });
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(obj, bo + o_obj, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(prim, bo + o_prim, N * 4, hipMemcpyDeviceToHost));
    if (gprim) HIP_TRY(c, hipMemcpy(gprim, bo + o_gprim, N * 4, hipMemcpyDeviceToHost));
    if (how) HIP_TRY(c, hipMemcpy(how, bo + o_how, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(t, bo + o_t, N * 8, hipMemcpyDeviceToHost));
    if (normal) HIP_TRY(c, hipMemcpy(normal, bo + o_n, n3, hipMemcpyDeviceToHost));
    if (pos) HIP_TRY(c, hipMemcpy(pos, bo + o_p, n3, hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// accumulation buffer
// ---------------------------------------------------------------------------------------------
int rt_accum_resize(rt_ctx *c, int32_t w, int32_t h) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (w <= 0 || h <= 0 || (long long)w * h > (1ll << 28)) return fail(c, RT_ERR_INVALID_ARG, "rt_accum_resize: bad size %d x %d", w, h);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, c->accum.ensure((size_t)w * h * 3 * sizeof(float)));
    c->acc_w = w; c->acc_h = h;
    HIP_TRY(c, hipMemset(c->accum.p, 0, (size_t)w * h * 3 * sizeof(float)));
    return RT_OK;
}

int rt_accum_clear(rt_ctx *c) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!c->accum.p) return fail(c, RT_ERR_INVALID_ARG, "rt_accum_clear: no buffer");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemset(c->accum.p, 0, (size_t)c->acc_w * c->acc_h * 3 * sizeof(float)));
    return RT_OK;
}

void *rt_accum_device_ptr(rt_ctx *c) { return (c && c->acc_w) ? c->accum.p : nullptr; }

int rt_accum_read(rt_ctx *c, float *host) {
    if (!c || !host) return fail(c, RT_ERR_INVALID_ARG, "rt_accum_read: NULL argument");
    if (!c->acc_w) return fail(c, RT_ERR_INVALID_ARG, "rt_accum_read: no buffer");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(host, c->accum.p, (size_t)c->acc_w * c->acc_h * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// render, wavefront form (see rt_wavefront.h): scenes searched through the hierarchy
// ---------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

int render_wavefront(rt_ctx *c, const rt_render_params *p, float *accum, hipStream_t stream) {
    const long long rect_px = (long long)(p->x1 - p->x0) * (p->y1 - p->y0);
    // paths in flight per batch: enough that even the last, sparsely populated segments fill the GPU
    const long long target = p->wavefront_paths > 0 ? std::max<long long>(1ll << 16, p->wavefront_paths) : 16ll << 20;
    const int batch_cap = (int)std::max<long long>(1, std::min<long long>(p->sample_count, target / rect_px));
    const size_t cap_paths = (size_t)rect_px * batch_cap;
    if (cap_paths >= (1ull << 31)) return fail(c, RT_ERR_INVALID_ARG, "rt_render: rectangle too large for one wavefront batch");
    const int n_seg = p->max_bounces + 1;
    bool grow = false;
    for (int k = 0; k < 2; ++k) grow = grow || cap_paths * sizeof(WfRay) > c->wf_rays[k].bytes;
    const size_t n_part_rows = (cap_paths + 255) / 256;
    grow = grow || cap_paths * sizeof(WfPath) > c->wf_paths.bytes || (size_t)(3 * (64 + 1)) * 4 > c->wf_counters.bytes || cap_paths * 4 > c->wf_settle.bytes ||
           n_part_rows * kStatCount * 8 > c->wf_partials.bytes;
    if (grow) {
        HIP_TRY(c, hipDeviceSynchronize());   // earlier launches may still use the old buffers
        for (int k = 0; k < 2; ++k) HIP_TRY(c, c->wf_rays[k].ensure(cap_paths * sizeof(WfRay)));
        HIP_TRY(c, c->wf_paths.ensure(cap_paths * sizeof(WfPath)));
        HIP_TRY(c, c->wf_hits.ensure(cap_paths * sizeof(WfHit)));
        HIP_TRY(c, c->wf_radiance.ensure(cap_paths * 3 * sizeof(float)));
        HIP_TRY(c, c->wf_counters.ensure((size_t)(3 * (64 + 1)) * 4));
        HIP_TRY(c, c->wf_settle.ensure(cap_paths * 4));
        HIP_TRY(c, c->wf_partials.ensure(n_part_rows * kStatCount * 8));
        HIP_TRY(c, hipMemset(c->wf_partials.p, 0, c->wf_partials.bytes));
    }
    if (int st = ensure_ray_table(c, p->width, p->height, stream)) return st;

    EventPair ev;
    bool have_ev = false;
    { std::lock_guard<std::mutex> g(c->mu); if (!c->spare_events.empty()) { ev = c->spare_events.back(); c->spare_events.pop_back(); have_ev = true; } }
    if (!have_ev) { HIP_TRY(c, hipEventCreate(&ev.a)); HIP_TRY(c, hipEventCreate(&ev.b)); }
    HIP_TRY(c, hipEventRecord(ev.a, stream));
    const DevScene S = c->S;
    uint32_t *counts = c->wf_counters.as<uint32_t>();          // [n_seg + 1] rays queued per segment
    uint32_t *cursors = counts + (64 + 1);                      // [n_seg] extend's read position per segment
    uint32_t *settles = cursors + (64 + 1);                     // [n_seg] queries per segment that k_wf_settle has to close
    for (int s0 = 0; s0 < p->sample_count; s0 += batch_cap) {
        const int bs = std::min(batch_cap, p->sample_count - s0);
        WfArgs A{};
        A.width = p->width; A.height = p->height; A.x0 = p->x0; A.y0 = p->y0; A.rect_w = p->x1 - p->x0; A.rect_h = p->y1 - p->y0;
        A.ray_dir = c->ray_dir.as<float>(); A.ray_res = c->ray_res.as<float>();
        A.sample_begin = p->sample_begin + s0; A.batch_samples = bs; A.n_paths = (uint32_t)(rect_px * bs);
        A.div_samples = make_fast_div((uint32_t)bs); A.div_rect_h = make_fast_div((uint32_t)A.rect_h);
        A.max_bounces = p->max_bounces; A.seed_lo = (uint32_t)p->seed; A.seed_hi = (uint32_t)(p->seed >> 32);
        A.paths = c->wf_paths.as<WfPath>(); A.hits = c->wf_hits.as<WfHit>(); A.radiance = c->wf_radiance.as<float>();
        A.accum = accum; A.stats = c->stats.as<unsigned long long>(); A.partials = c->wf_partials.as<unsigned long long>();
        HIP_TRY(c, hipMemsetAsync(counts, 0, (size_t)(3 * (64 + 1)) * 4, stream));
        const unsigned path_blocks = (unsigned)((A.n_paths + 255) / 256);
        A.rays_in = c->wf_rays[0].as<WfRay>(); A.rays_out = c->wf_rays[1].as<WfRay>();
        A.count_in = counts; A.count_out = counts + 1; A.cursor = cursors;
        A.settle_list = c->wf_settle.as<uint32_t>(); A.settle_count = settles;
        hipLaunchKernelGGL(k_wf_generate, dim3(path_blocks), dim3(256), 0, stream, S, A);
        const unsigned extend_blocks = std::min<unsigned>(path_blocks, (unsigned)c->n_cu * 5u);
        for (int seg = 0; seg < n_seg; ++seg) {
            A.rays_in = c->wf_rays[seg & 1].as<WfRay>(); A.rays_out = c->wf_rays[(seg + 1) & 1].as<WfRay>();
            A.count_in = counts + seg; A.count_out = counts + seg + 1; A.cursor = cursors + seg; A.settle_count = settles + seg;
            const unsigned settle_blocks = std::min<unsigned>(path_blocks, (unsigned)c->n_cu * 4u);
            if (c->has_sph) {
                hipLaunchKernelGGL((k_wf_extend<true>), dim3(extend_blocks), dim3(256), 0, stream, S, A);
                hipLaunchKernelGGL((k_wf_settle<true>), dim3(settle_blocks), dim3(256), 0, stream, S, A);
            } else {
                hipLaunchKernelGGL((k_wf_extend<false>), dim3(extend_blocks), dim3(256), 0, stream, S, A);
                hipLaunchKernelGGL((k_wf_settle<false>), dim3(settle_blocks), dim3(256), 0, stream, S, A);
            }
            hipLaunchKernelGGL(k_wf_shade, dim3(path_blocks), dim3(256), 0, stream, S, A);
        }
        hipLaunchKernelGGL(k_wf_resolve, dim3((unsigned)((rect_px + 255) / 256)), dim3(256), 0, stream, A);
        HIP_TRY(c, hipGetLastError());
    }
    hipLaunchKernelGGL(k_wf_fold_stats, dim3(1), dim3(256), 0, stream, c->wf_partials.as<unsigned long long>(), (int)n_part_rows,
                       c->stats.as<unsigned long long>());
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(ev.b, stream));
    { std::lock_guard<std::mutex> g(c->mu); c->timed.push_back(ev); c->launches++; }
    return RT_OK;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
// render
// ---------------------------------------------------------------------------------------------
int rt_render(rt_ctx *c, const rt_render_params *p, float *accum_dev, void *stream_) {
    if (!c || !p) return fail(c, RT_ERR_INVALID_ARG, "rt_render: NULL argument");
    if (p->struct_size != sizeof(rt_render_params)) return fail(c, RT_ERR_INVALID_ARG, "rt_render_params.struct_size mismatch");
    if (!c->has_scene) return fail(c, RT_ERR_NO_SCENE, "rt_render: no scene set");
    if (c->stop.load()) return fail(c, RT_ERR_STOPPED, "rt_render: stop requested");
    if (p->width <= 0 || p->height <= 0 || p->x0 < 0 || p->y0 < 0 || p->x1 > p->width || p->y1 > p->height || p->x1 < p->x0 || p->y1 < p->y0)
        return fail(c, RT_ERR_INVALID_ARG, "rt_render: rectangle [%d,%d)x[%d,%d) outside %dx%d", p->x0, p->x1, p->y0, p->y1, p->width, p->height);
    if (p->sample_count < 0 || p->sample_begin < 0 || p->max_bounces < 0 || p->max_bounces > 62)
        return fail(c, RT_ERR_INVALID_ARG, "rt_render: bad sample range or max_bounces");
    if ((p->flags & ~(uint32_t)(RT_RENDER_WAVEFRONT | RT_RENDER_STRICT_DEPTH | RT_RENDER_NO_DEFER | RT_RENDER_LOOSE_DEPTH)) != 0 || p->wavefront_paths < 0)
        return fail(c, RT_ERR_INVALID_ARG, "rt_render: unknown flag bits 0x%x or negative wavefront_paths", p->flags);
    if ((p->flags & RT_RENDER_LOOSE_DEPTH) && (p->flags & (RT_RENDER_STRICT_DEPTH | RT_RENDER_WAVEFRONT)))
        return fail(c, RT_ERR_INVALID_ARG, "rt_render: RT_RENDER_LOOSE_DEPTH goes with neither RT_RENDER_STRICT_DEPTH nor RT_RENDER_WAVEFRONT");
    float *accum = accum_dev;
    if (!accum) {
        if (c->acc_w != p->width || c->acc_h != p->height)
            return fail(c, RT_ERR_INVALID_ARG, "rt_render: context buffer is %dx%d, call rt_accum_resize(%d,%d) first", c->acc_w, c->acc_h, p->width, p->height);
        accum = c->accum.as<float>();
    }
    const long long rect_px = (long long)(p->x1 - p->x0) * (p->y1 - p->y0);
    if (rect_px == 0 || p->sample_count == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = (hipStream_t)stream_;
    // The wavefront form exists for hierarchy scenes and is selectable, but the single kernel below
    // is faster on every configuration measured (room + 1.18 M triangles, 1080p: 4.1 against 3.3
    // G rays/s with 64 M paths in flight, 1.4 with 4 M), so it stays the default.
    if (c->tri_mode == kTriBvh && (p->flags & RT_RENDER_WAVEFRONT))
        return render_wavefront(c, p, accum, stream);

    // Job granularity: enough (pixel, chunk) jobs that every resident lane runs many of them, so the
    // end-of-launch tail (lanes waiting for the last jobs) stays a few percent of the launch.
    const long long lanes = (long long)c->n_cu * 2048;
    // (measured, jobs numbered pixel by pixel, windows of 64 that halve near the end of the range, factors 16 / 24 / 32 /
    // 48 / 64 / 80: room 46.40 / 46.16 / 46.18 / 46.47 / 46.73 / 47.2 ms per 256 spp and 23.70 / 23.81 / 23.99 / 24.28 per
    // 128 spp - jobs of 25 to 40 samples; 1.18 M triangles, 128 spp: 105.1 / 104.8 / 104.8 / 104.8.  With windows of 256,
    // factors 24 / 48 / 96 / 192 had measured 45.96 / 45.61 / 46.86 / 48.46 in the loose depth mode.)  Where a path segment is
    // dear - many table rows, or a hierarchy - longer jobs only lengthen the launch's tail: spheres scene, 64 spp, factor
    // 24 / 48: 16.05 / 15.65 ms; there the factor stays 48.
    const bool cheap_segments = c->tri_mode != kTriBvh && c->S.n_tri_rows + c->S.n_sph_rows + c->S.n_big_sph <= 64;   // a few dozen table rows, no hierarchy
    long long want_chunks = ((cheap_segments ? 24 : 48) * lanes + rect_px - 1) / rect_px;
    // ... but no job shorter than eight samples: a call of few samples per pixel (one GPU's share of a frame that
    // many GPUs render together) would otherwise spend its time handing jobs out and folding their slots (measured,
    // 1080p room, 32 / 64 spp per call: 7.73 -> 7.15 ms and 13.58 -> 13.09 ms; at least 4 or 16 samples: 7.39 / 7.47 ms)
    // Sixteen where a path segment is cheap (see above): room, 32 / 64 spp per call 6.74 -> 6.62
    // and 12.73 -> 12.45 ms (24 or 32: 7.05 / 12.6); where segments are dear, longer jobs only lengthen the launch's tail:
    // spheres scene 32 spp 7.5 -> 7.9 ms, 1.18 M triangles 28.0 -> 29.4.
    want_chunks = std::min<long long>(want_chunks, std::max<long long>(1, p->sample_count / (cheap_segments ? 16 : 8)));
    int n_chunks = (int)std::min<long long>(std::max<long long>(want_chunks, 1), p->sample_count);
    int chunk_samples = (p->sample_count + n_chunks - 1) / n_chunks;
    n_chunks = (p->sample_count + chunk_samples - 1) / chunk_samples;
    const long long total_jobs = rect_px * n_chunks;
    if (total_jobs >= (1ll << 32) - 2 * kJobBlock) return fail(c, RT_ERR_INVALID_ARG, "rt_render: too many jobs in one call; split the sample range");

    RenderArgs A{};
    A.width = p->width; A.height = p->height; A.x0 = p->x0; A.y0 = p->y0; A.rect_w = p->x1 - p->x0; A.rect_h = p->y1 - p->y0;
    A.sample_begin = p->sample_begin; A.sample_count = p->sample_count; A.chunk_samples = chunk_samples; A.n_chunks = n_chunks;
    if (int st = ensure_ray_table(c, p->width, p->height, stream)) return st;
    A.ray_dir = c->ray_dir.as<float>(); A.ray_res = c->ray_res.as<float>();
    A.max_bounces = p->max_bounces; A.seed_lo = (uint32_t)p->seed; A.seed_hi = (uint32_t)(p->seed >> 32);
    A.total_jobs = (uint32_t)total_jobs;
    A.div_chunks = make_fast_div((uint32_t)n_chunks); A.div_rect_h = make_fast_div((uint32_t)A.rect_h); A.job_counter = c->job_counter.as<uint32_t>(); A.accum = accum;
    A.stats = c->stats.as<unsigned long long>();
#if defined(RT_SECTION_TIMERS) || defined(RT_ASIDE_REASONS)
    A.prof = A.stats + kStatCount;
#endif
#ifdef RT_TUNE
    {   // tuning builds only (the shipped library reads no environment variable)
        auto num = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
        A.tune[0] = num("RT_TUNE_SLICE", 3); A.tune[1] = num("RT_TUNE_MORE_BELOW", 32); A.tune[2] = num("RT_TUNE_MORE_TIMES", 2); A.tune[3] = num("RT_TUNE_SUB_SLICE", 3);
    }
#endif
    // The lean kernel (no double-precision settling code, five waves per SIMD) renders; the few paths it sets
    // aside are traced again by k_aside, the jobs it gives up (no room left in the list) by the full kernel;
    // k_reduce_chunks folds everything into the buffer.  Strict depth (a tenth of the rays - every imprecise sphere
    // depth - would be set aside: measured 7.9 against 37.6 G rays/s) and RT_RENDER_NO_DEFER go straight to the full
    // kernel.  All on `stream`, no host synchronisation.
    const bool strict = !(p->flags & RT_RENDER_LOOSE_DEPTH);
    const bool lean = !(p->flags & RT_RENDER_NO_DEFER);
    const unsigned long long total_paths = (unsigned long long)rect_px * (unsigned long long)p->sample_count;
    const uint32_t aside_cap = lean ? (uint32_t)std::min<unsigned long long>(std::max<unsigned long long>(total_paths / 64, 4096), 1ull << 25) : 0;
    {
        // scratch grows only; a larger request first drains work that may still read the old buffers
        const size_t need_partial = (size_t)total_jobs * 3 * sizeof(double), need_aside = (size_t)aside_cap * sizeof(AsideRec);
        const size_t need_head = lean ? (size_t)total_jobs * sizeof(int32_t) : 0, need_dirty = lean ? (size_t)total_jobs : 0, need_list = lean ? (size_t)total_jobs * 4 : 0;
        if (need_partial > c->partial.bytes || need_aside > c->aside.bytes || need_head > c->aside_head.bytes ||
            need_dirty > c->dirty.bytes || need_list > c->dirty_list.bytes) {
            HIP_TRY(c, hipDeviceSynchronize());
            HIP_TRY(c, c->partial.ensure(need_partial)); HIP_TRY(c, c->aside.ensure(need_aside)); HIP_TRY(c, c->aside_head.ensure(need_head));
            HIP_TRY(c, c->dirty.ensure(need_dirty)); HIP_TRY(c, c->dirty_list.ensure(need_list));
        }
        A.partial = c->partial.as<double>();
        uint32_t *counters = c->job_counter.as<uint32_t>();   // [0] jobs handed out, [1] records in the list, [2] jobs given up, [3] jobs handed out again, [4] records handed out
        if (lean) {
            A.aside = c->aside.as<AsideRec>(); A.aside_count = counters + 1; A.aside_cap = aside_cap;
            A.dirty_list = c->dirty_list.as<uint32_t>(); A.dirty_count = counters + 2; A.dirty = c->dirty.as<uint8_t>();
            A.head = c->aside_head.as<int32_t>();
            HIP_TRY(c, hipMemsetAsync(c->aside_head.p, 0xFF, need_head, stream));
            HIP_TRY(c, hipMemsetAsync(c->dirty.p, 0, need_dirty, stream));
        }
    }
    HIP_TRY(c, hipMemsetAsync(c->job_counter.p, 0, 256, stream));
    const long long waves_needed = (total_jobs + 63) / 64;
    const unsigned blocks = (unsigned)std::min<long long>((waves_needed + 3) / 4, (long long)c->n_cu * 8);
    EventPair ev;
    bool have_ev = false;
    { std::lock_guard<std::mutex> g(c->mu); if (!c->spare_events.empty()) { ev = c->spare_events.back(); c->spare_events.pop_back(); have_ev = true; } }
    if (!have_ev) { HIP_TRY(c, hipEventCreate(&ev.a)); HIP_TRY(c, hipEventCreate(&ev.b)); }
    HIP_TRY(c, hipEventRecord(ev.a, stream));
    const DevScene S = c->S;
    dispatch_variant(c->tri_mode, c->has_sph, [&](auto T, auto P) {
        constexpr int TRI = decltype(T)::value; constexpr bool SPH = decltype(P)::value;
        auto full = [&](unsigned grid, const RenderArgs &args) {
            if (strict) hipLaunchKernelGGL((k_render<TRI, SPH, true, false>), dim3(grid), dim3(256), 0, stream, S, args);
            else hipLaunchKernelGGL((k_render<TRI, SPH, false, false>), dim3(grid), dim3(256), 0, stream, S, args);
        };
        if (!lean) { full(blocks, A); return; }
        if (strict) hipLaunchKernelGGL((k_render<TRI, SPH, true, true>), dim3(blocks), dim3(256), 0, stream, S, A);
        else hipLaunchKernelGGL((k_render<TRI, SPH, false, true>), dim3(blocks), dim3(256), 0, stream, S, A);
        // the paths set aside: the full kernel's job loop over the records, one path each
        RenderArgs Q = A;
        Q.path_mode = 1; Q.job_list_count = A.aside_count; Q.job_counter = A.job_counter + 4;
#ifdef RT_SECTION_TIMERS
        Q.prof = A.prof + kProfCount;   // (a bank of its own for the path-list launch)
#endif
        full(std::min(blocks, (unsigned)c->n_cu * 4), Q);
        // the jobs given up: the same job loop over the list (nearly always an empty one)
        RenderArgs R = A;
        R.job_list = A.dirty_list; R.job_list_count = A.dirty_count; R.job_counter = A.job_counter + 3;
        full(std::min(blocks, (unsigned)c->n_cu * 4), R);
    });
    HIP_TRY(c, hipGetLastError());
    if (n_chunks <= 256) {
        const int px_per_block = 256 / n_chunks;
        hipLaunchKernelGGL(k_reduce_jobs, dim3((unsigned)((rect_px + px_per_block - 1) / px_per_block)), dim3(256), 0, stream, A, px_per_block);
    } else {
        hipLaunchKernelGGL(k_reduce_chunks, dim3((unsigned)((rect_px + 255) / 256)), dim3(256), 0, stream, A);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(ev.b, stream));   // (the timed span is everything the call launched: render, set-aside paths, jobs again, fold)
    { std::lock_guard<std::mutex> g(c->mu); c->timed.push_back(ev); c->launches++; c->paths_launched += (uint64_t)rect_px * (uint64_t)p->sample_count; }
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// merge across GPUs: one RCCL all-reduce of the accumulation buffers
// ---------------------------------------------------------------------------------------------
}  // extern "C"

namespace {


}  // namespace

extern "C" {

int rt_comm_available(void) {
    Rccl &r = rccl();
    if (!r.handle) return fail(nullptr, RT_ERR_UNSUPPORTED, "%s", r.why.c_str());
    return RT_OK;
}

int rt_comm_unique_id(void *out_128_bytes) {
    if (!out_128_bytes) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_comm_unique_id: NULL argument");
    Rccl &r = rccl();
    if (!r.handle) return fail(nullptr, RT_ERR_UNSUPPORTED, "%s", r.why.c_str());
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "rt_comm_unique_id hands out an ncclUniqueId");
    ncclUniqueId id;
    const ncclResult_t e = r.get_unique_id(&id);
    if (e != ncclSuccess) return fail(nullptr, RT_ERR_HIP, "ncclGetUniqueId failed: %s", rccl_error(e));
    std::memcpy(out_128_bytes, &id, sizeof id);
    return RT_OK;
}

int rt_comm_init(rt_ctx *c, int32_t n_ranks, int32_t rank, const void *unique_id_128_bytes) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, RT_ERR_INVALID_ARG, "rt_comm_init: rank %d of %d", rank, n_ranks);
    if (c->comm) return fail(c, RT_ERR_INVALID_ARG, "rt_comm_init: this context already has a communicator");
    if (n_ranks > 1 && !unique_id_128_bytes) return fail(c, RT_ERR_INVALID_ARG, "rt_comm_init: more than one rank needs rank 0's id");
    c->comm_ranks = n_ranks; c->comm_rank = rank;
    if (!unique_id_128_bytes) return RT_OK;   // one rank, no id given: rt_merge has nothing to do and RCCL stays unopened
    Rccl &r = rccl();
    if (!r.handle) return fail(c, RT_ERR_UNSUPPORTED, "%s", r.why.c_str());
    HIP_TRY(c, hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id_128_bytes, sizeof id);
    const ncclResult_t e = r.comm_init_rank(&c->comm, n_ranks, id, rank);
    if (e != ncclSuccess) { c->comm = nullptr; return fail(c, RT_ERR_HIP, "ncclCommInitRank failed: %s", rccl_error(e)); }
    return RT_OK;
}

int rt_merge(rt_ctx *c, float *accum_dev, int32_t width, int32_t height, void *stream) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 28)) return fail(c, RT_ERR_INVALID_ARG, "rt_merge: bad size %d x %d", width, height);
    float *buf = accum_dev;
    if (!buf) {
        if (c->acc_w != width || c->acc_h != height) return fail(c, RT_ERR_INVALID_ARG, "rt_merge: context buffer is %dx%d, not %dx%d", c->acc_w, c->acc_h, width, height);
        buf = c->accum.as<float>();
    }
    if (c->comm_ranks == 1 && !c->comm) return RT_OK;   // nothing to merge with
    if (!c->comm) return fail(c, RT_ERR_INVALID_ARG, "rt_merge: %d ranks but no communicator (rt_comm_init)", c->comm_ranks);
    HIP_TRY(c, hipSetDevice(c->device));
    const ncclResult_t e = rccl().all_reduce(buf, buf, (size_t)width * height * 3, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
    if (e != ncclSuccess) return fail(c, RT_ERR_HIP, "ncclAllReduce failed: %s", rccl_error(e));
    return RT_OK;
}

// Hands the context's grow-only scratch back to the device (see the header): the render's job slots and set-aside
// lists, the wavefront form's queues, the closest-hit entry's staging.  Tables of the scene, the accumulation buffer
// and the camera-ray table stay.  Synchronises the device first; the next call that needs scratch allocates it again.
int rt_release_scratch(rt_ctx *c) {
    if (!c) return RT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    for (DevBuf *b : {&c->partial, &c->aside, &c->aside_head, &c->dirty, &c->dirty_list, &c->wf_rays[0], &c->wf_rays[1], &c->wf_paths,
                      &c->wf_hits, &c->wf_radiance, &c->wf_partials, &c->wf_settle, &c->isect_in, &c->isect_out, &c->post_partials})
        b->release();
    return RT_OK;
}

int rt_synchronize(rt_ctx *c, void *stream) {
    if (!c) return RT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize((hipStream_t)stream));
    return RT_OK;
}

int rt_request_stop(rt_ctx *c) { if (!c) return RT_ERR_INVALID_ARG; c->stop.store(1); return RT_OK; }
int rt_clear_stop(rt_ctx *c) { if (!c) return RT_ERR_INVALID_ARG; c->stop.store(0); return RT_OK; }

static int drain_events(rt_ctx *c) {
    std::vector<EventPair> evs;
    { std::lock_guard<std::mutex> g(c->mu); evs.swap(c->timed); }
    double ms_sum = 0, ms_min = INFINITY, ms_max = 0; uint64_t n = 0;
    for (auto &e : evs) {
        float ms = 0;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            ms_sum += ms; ms_min = std::min(ms_min, (double)ms); ms_max = std::max(ms_max, (double)ms); ++n;
        }
    }
    {
        std::lock_guard<std::mutex> g(c->mu);
        c->kernel_ms += ms_sum;
        if (n) {
            c->kernel_ms_min = c->launches_timed ? std::min(c->kernel_ms_min, ms_min) : ms_min;
            c->kernel_ms_max = std::max(c->kernel_ms_max, ms_max);
            c->launches_timed += n;
        }
        c->spare_events.insert(c->spare_events.end(), evs.begin(), evs.end());
    }
    return RT_OK;
}

int rt_get_stats(rt_ctx *c, rt_stats *out) {
    if (!c || !out) return fail(c, RT_ERR_INVALID_ARG, "rt_get_stats: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    drain_events(c);
    unsigned long long v[kStatCount];
    HIP_TRY(c, hipMemcpy(v, c->stats.p, sizeof v, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    out->rays = v[kStatRays]; out->paths = v[kStatPaths]; out->sphere_tests = v[kStatSphTests]; out->tri_tests = v[kStatTriTests];
    out->aabb_tests = 0; out->bvh_node_tests = v[kStatBvhNodes]; out->exact_rays = v[kStatExact]; out->refined_hits = v[kStatRefine];
    out->sky_lookups = v[kStatSky]; out->shaded_hits = v[kStatRays] - v[kStatSky];   // every query ends in a shaded hit or the environment
    { std::lock_guard<std::mutex> g(c->mu); out->kernel_ms = c->kernel_ms; out->launches = c->launches; out->paths += c->paths_launched;
      out->kernel_ms_min = c->kernel_ms_min; out->kernel_ms_max = c->kernel_ms_max; }
    return RT_OK;
}

#if defined(RT_SECTION_TIMERS) || defined(RT_ASIDE_REASONS)
// Diagnostic builds only: wave cycles charged to each section of the render loop since rt_reset_stats
// (RT_ASIDE_REASONS builds: slots 0-6 count the paths set aside by reason instead).
int rt_debug_section_cycles(rt_ctx *c, unsigned long long *out, int n) {
    if (!c || !out) return RT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    unsigned long long v[2 * kProfCount];
    HIP_TRY(c, hipMemcpy(v, c->stats.as<unsigned long long>() + kStatCount, sizeof v, hipMemcpyDeviceToHost));
    for (int i = 0; i < n && i < 2 * kProfCount; ++i) out[i] = v[i];
    return RT_OK;
}
#endif

#ifdef RT_WAVE_END
int rt_debug_wave_log(rt_ctx *c, unsigned long long *out, int n) {
    if (!c || !out) return RT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_log), sizeof(unsigned long long) * (size_t)std::min(n, 6 * 8192)));
    return RT_OK;
}
#endif

int rt_reset_stats(rt_ctx *c) {
    if (!c) return RT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    drain_events(c);
    HIP_TRY(c, hipMemset(c->stats.p, 0, sizeof(unsigned long long) * (kStatCount + 2 * kProfCount)));
    { std::lock_guard<std::mutex> g(c->mu); c->kernel_ms = 0; c->launches = 0; c->paths_launched = 0; c->kernel_ms_min = c->kernel_ms_max = 0; c->launches_timed = 0; }
    return RT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// RNG known-answer hook
// ---------------------------------------------------------------------------------------------
namespace {
__global__ void k_debug_uniforms(uint32_t key0, uint32_t key1, int n_segments, float *out) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segments) return;
    Draws d = segment_draws(key0, key1, (uint32_t)s);
    out[5 * s] = d.f0; out[5 * s + 1] = d.f1; out[5 * s + 2] = d.f2; out[5 * s + 3] = d.f3; out[5 * s + 4] = d.f4;
}
}  // namespace

extern "C" int rt_debug_uniforms(rt_ctx *c, uint32_t pixel, uint32_t sample, uint64_t seed, int32_t n_segments, float *out_host) {
    if (!c || !out_host || n_segments <= 0) return fail(c, RT_ERR_INVALID_ARG, "rt_debug_uniforms: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf b;
    HIP_TRY(c, b.ensure((size_t)n_segments * 5 * sizeof(float)));
    hipLaunchKernelGGL(k_debug_uniforms, dim3((n_segments + 63) / 64), dim3(64), 0, 0, pixel ^ (uint32_t)seed, sample ^ (uint32_t)(seed >> 32), n_segments, b.as<float>());
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(out_host, b.p, (size_t)n_segments * 5 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

// What the last rt_render call set aside (tests and tuning): out[0] jobs handed out, [1] paths set aside by the lean
// kernel, [2] jobs it gave up, [3] jobs the full kernel took from that list.  Synchronises.
extern "C" int rt_debug_render_counters(rt_ctx *c, uint32_t *out4) {
    if (!c || !out4) return fail(c, RT_ERR_INVALID_ARG, "rt_debug_render_counters: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(out4, c->job_counter.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

// How many walks gave their stack up and took the stackless walk of last resort since the library was loaded
// on this context's GPU (tests and tuning).  Synchronises.
extern "C" int rt_debug_walk_fallbacks(rt_ctx *c, uint64_t *out) {
    if (!c || !out) return fail(c, RT_ERR_INVALID_ARG, "rt_debug_walk_fallbacks: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    unsigned long long v = 0;
    HIP_TRY(c, hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_walk_fallbacks), sizeof v));
    *out = v;
    return RT_OK;
}

// How the current scene's triangles were laid out (tests and tuning).
extern "C" int rt_debug_scene_layout(rt_ctx *c, int32_t *out, int32_t n_out, float *box_slack) {
    if (!c || !out || n_out < 8) return fail(c, RT_ERR_INVALID_ARG, "rt_debug_scene_layout: bad arguments");
    if (!c->has_scene) return fail(c, RT_ERR_NO_SCENE, "rt_debug_scene_layout: no scene");
    const DevScene &S = c->S;
    out[0] = S.n_quad_rows_padded; out[1] = S.n_quad_general;
    out[2] = (int32_t)(S.slab_counts & 1023u); out[3] = (int32_t)((S.slab_counts >> 10) & 1023u); out[4] = (int32_t)(S.slab_counts >> 20);
    out[5] = S.n_tri_rows_padded; out[6] = S.n_boxes; out[7] = S.n_bvh_nodes;
    if (n_out >= 9) out[8] = c->last_refit ? 1 : 0;
    if (n_out >= 10) out[9] = c->walk_stack_need;
    if (box_slack) *box_slack = S.box_slack;
    return RT_OK;
}

// Whole paths, one per thread, with every segment recorded (parity tests; not a fast path).
extern "C" int rt_debug_trace_paths(rt_ctx *c, int32_t n_paths, const int32_t *xys, int32_t width, int32_t height,
                                    int32_t max_bounces, uint64_t seed, uint32_t render_flags, float *rgb, int32_t *n_seg, int32_t *seg_obj,
                                    int32_t *seg_prim, int32_t *seg_how, float *seg_t, float *seg_org, float *seg_dir, float *seg_org_residue) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (n_paths <= 0 || !xys || !rgb || !n_seg || !seg_obj || !seg_prim || !seg_how || !seg_t || !seg_org || !seg_dir ||
        width <= 0 || height <= 0 || max_bounces < 0 || max_bounces > 62)
        return fail(c, RT_ERR_INVALID_ARG, "rt_debug_trace_paths: bad arguments");
    if (!c->has_scene) return fail(c, RT_ERR_NO_SCENE, "rt_debug_trace_paths: no scene set");
    HIP_TRY(c, hipSetDevice(c->device));
    const int cap = max_bounces + 1;
    const size_t np = (size_t)n_paths, ns = np * cap;
    DevBuf b_xys, b_rgb, b_n, b_obj, b_prim, b_how, b_t, b_org, b_dir, b_res;
    HIP_TRY(c, b_xys.ensure(np * 12)); HIP_TRY(c, hipMemcpy(b_xys.p, xys, np * 12, hipMemcpyHostToDevice));
    HIP_TRY(c, b_rgb.ensure(np * 12)); HIP_TRY(c, b_n.ensure(np * 4));
    HIP_TRY(c, b_obj.ensure(ns * 4)); HIP_TRY(c, b_prim.ensure(ns * 4)); HIP_TRY(c, b_how.ensure(ns * 4)); HIP_TRY(c, b_t.ensure(ns * 4));
    HIP_TRY(c, b_org.ensure(ns * 12)); HIP_TRY(c, b_dir.ensure(ns * 12));
    if (seg_org_residue) { HIP_TRY(c, b_res.ensure(ns * 12)); HIP_TRY(c, hipMemset(b_res.p, 0, ns * 12)); }
    HIP_TRY(c, hipMemset(b_obj.p, 0xFF, ns * 4)); HIP_TRY(c, hipMemset(b_prim.p, 0xFF, ns * 4)); HIP_TRY(c, hipMemset(b_how.p, 0xFF, ns * 4));
    HIP_TRY(c, hipMemset(b_t.p, 0, ns * 4)); HIP_TRY(c, hipMemset(b_org.p, 0, ns * 12)); HIP_TRY(c, hipMemset(b_dir.p, 0, ns * 12));
    TraceArgs A{};
    A.n_paths = n_paths; A.width = width; A.height = height; A.max_bounces = max_bounces; A.seg_cap = cap;
    A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32);
    if (int st = ensure_ray_table(c, width, height, nullptr)) return st;
    A.ray_dir = c->ray_dir.as<float>(); A.ray_res = c->ray_res.as<float>();
    A.xys = b_xys.as<int32_t>(); A.rgb = b_rgb.as<float>(); A.n_seg = b_n.as<int32_t>(); A.seg_obj = b_obj.as<int32_t>();
    A.seg_prim = b_prim.as<int32_t>(); A.seg_how = b_how.as<int32_t>(); A.seg_t = b_t.as<float>();
    A.seg_org = b_org.as<float>(); A.seg_dir = b_dir.as<float>(); A.seg_org_res = seg_org_residue ? b_res.as<float>() : nullptr;
    const DevScene S = c->S;
    dispatch_variant(c->tri_mode, c->has_sph, [&](auto T, auto P) {
        if (!(render_flags & RT_RENDER_LOOSE_DEPTH))
            hipLaunchKernelGGL((k_trace_paths<decltype(T)::value, decltype(P)::value, true>),
                               dim3((unsigned)((n_paths + 255) / 256)), dim3(256), 0, 0, S, A);
        else
            hipLaunchKernelGGL((k_trace_paths<decltype(T)::value, decltype(P)::value, false>),
                               dim3((unsigned)((n_paths + 255) / 256)), dim3(256), 0, 0, S, A);
    });
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(rgb, b_rgb.p, np * 12, hipMemcpyDeviceToHost)); HIP_TRY(c, hipMemcpy(n_seg, b_n.p, np * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(seg_obj, b_obj.p, ns * 4, hipMemcpyDeviceToHost)); HIP_TRY(c, hipMemcpy(seg_prim, b_prim.p, ns * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(seg_how, b_how.p, ns * 4, hipMemcpyDeviceToHost)); HIP_TRY(c, hipMemcpy(seg_t, b_t.p, ns * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(seg_org, b_org.p, ns * 12, hipMemcpyDeviceToHost)); HIP_TRY(c, hipMemcpy(seg_dir, b_dir.p, ns * 12, hipMemcpyDeviceToHost));
    if (seg_org_residue) HIP_TRY(c, hipMemcpy(seg_org_residue, b_res.p, ns * 12, hipMemcpyDeviceToHost));
    return RT_OK;
}

// Renderer.raytrace for a batch of caller-supplied rays (see k_raytrace).
extern "C" int rt_raytrace_batch(rt_ctx *c, int64_t n, const double *org, const double *dir, const uint8_t *inside,
                                 int32_t first_depth, int32_t max_bounces, uint64_t seed, uint32_t sample,
                                 uint32_t stream_begin, float *rgb) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && (!org || !dir || !rgb)) || first_depth < 0 || max_bounces < 0 || max_bounces > 62)
        return fail(c, RT_ERR_INVALID_ARG, "rt_raytrace_batch: bad arguments");
    if (n > (1ll << 31)) return fail(c, RT_ERR_INVALID_ARG, "rt_raytrace_batch: at most 2^31 rays per call");
    if (!c->has_scene) return fail(c, RT_ERR_NO_SCENE, "rt_raytrace_batch: no scene set");
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf b_org, b_dir, b_inside, b_rgb;
    const size_t n3 = 3 * (size_t)n * sizeof(double);
    HIP_TRY(c, b_org.ensure(n3)); HIP_TRY(c, b_dir.ensure(n3)); HIP_TRY(c, b_rgb.ensure(3 * (size_t)n * sizeof(float)));
    HIP_TRY(c, hipMemcpy(b_org.p, org, n3, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(b_dir.p, dir, n3, hipMemcpyHostToDevice));
    if (inside) {
        HIP_TRY(c, b_inside.ensure((size_t)n));
        HIP_TRY(c, hipMemcpy(b_inside.p, inside, (size_t)n, hipMemcpyHostToDevice));
    }
    RaytraceArgs A{};
    A.n = n; A.org = b_org.as<double>(); A.dir = b_dir.as<double>(); A.inside = inside ? b_inside.as<uint8_t>() : nullptr;
    A.first_depth = first_depth; A.max_bounces = max_bounces;
    A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32); A.sample = sample; A.stream_begin = stream_begin;
    A.rgb = b_rgb.as<float>(); A.stats = c->stats.as<unsigned long long>();
    const DevScene S = c->S;
    dispatch_variant(c->tri_mode, c->has_sph, [&](auto T, auto P) {
        hipLaunchKernelGGL((k_raytrace<decltype(T)::value, decltype(P)::value>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, S, A);
    });
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(rgb, b_rgb.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// post-processing (see rt_post.h)
// ---------------------------------------------------------------------------------------------
namespace {

// Argument checks and the three launches shared by rt_postprocess and rt_postprocess_noalloc; the
// two differ only in where the reduction slots live.
int postprocess_check(rt_ctx *c, const char *who, const float *src_dev, int32_t width, int32_t height, int32_t flip_y,
                      const rt_postprocess_params *p, float *dst_dev, const float **src_out) {
    if (!c || !p || !dst_dev) return fail(c, RT_ERR_INVALID_ARG, "%s: NULL argument", who);
    if (p->struct_size != sizeof(rt_postprocess_params)) return fail(c, RT_ERR_INVALID_ARG, "rt_postprocess_params.struct_size mismatch");
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 28)) return fail(c, RT_ERR_INVALID_ARG, "%s: bad size %d x %d (at most 2^28 pixels)", who, width, height);
    if (!(p->tone_saturation > 0.0f) || !(p->key > 0.0f)) return fail(c, RT_ERR_INVALID_ARG, "%s: key and tone_saturation must be positive", who);
    const float *src = src_dev;
    if (!src) {
        if (c->acc_w != width || c->acc_h != height) return fail(c, RT_ERR_INVALID_ARG, "%s: context buffer is %dx%d, not %dx%d", who, c->acc_w, c->acc_h, width, height);
        src = c->accum.as<float>();
    }
    if (src == dst_dev && flip_y) return fail(c, RT_ERR_INVALID_ARG, "%s: in place only without the flip", who);
    *src_out = src;
    return RT_OK;
}

int postprocess_launch(rt_ctx *c, const float *src, int32_t width, int32_t height, float scale, int32_t flip_y,
                       const rt_postprocess_params *p, float *dst_dev, double *partials, int n_blocks, hipStream_t stream) {
    const long long n = (long long)width * height;
    PostArgs A{};
    A.src = src; A.dst = dst_dev; A.width = width; A.height = height; A.flip_y = flip_y ? 1 : 0; A.scale = scale;
    A.exposure = p->exposure; A.brightness = p->brightness; A.contrast = p->contrast; A.saturation = p->saturation; A.gamma = p->gamma;
    A.key = p->key; A.tone_saturation = p->tone_saturation;
    A.partials = partials; A.n_blocks = n_blocks;
    hipLaunchKernelGGL(k_post_measure, dim3((unsigned)n_blocks), dim3(256), 0, stream, A);
    hipLaunchKernelGGL(k_post_fold, dim3(1), dim3(64), 0, stream, A);
    hipLaunchKernelGGL(k_post_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, A);
    HIP_TRY(c, hipGetLastError());
    return RT_OK;
}

int post_blocks(const rt_ctx *c, int32_t width, int32_t height) {
    const long long n = (long long)width * height;
    return (int)std::min<long long>({(n + 255) / 256, (long long)c->n_cu * 8, (long long)kPostScratchBlocks});
}

}  // namespace

extern "C" int rt_postprocess(rt_ctx *c, const float *src_dev, int32_t width, int32_t height, float scale, int32_t flip_y,
                              const rt_postprocess_params *p, float *dst_dev, void *stream_) {
    const float *src = nullptr;
    if (int st = postprocess_check(c, "rt_postprocess", src_dev, width, height, flip_y, p, dst_dev, &src)) return st;
    HIP_TRY(c, hipSetDevice(c->device));
    const int n_blocks = post_blocks(c, width, height);
    const size_t need = (size_t)(2 * n_blocks + 2) * sizeof(double);
    if (need > c->post_partials.bytes) { HIP_TRY(c, hipDeviceSynchronize()); HIP_TRY(c, c->post_partials.ensure(need)); }
    return postprocess_launch(c, src, width, height, scale, flip_y, p, dst_dev, c->post_partials.as<double>(), n_blocks, (hipStream_t)stream_);
}

// rt_postprocess over the fixed scratch array: no hipMalloc / hipFree / synchronisation, only three
// kernel launches on `stream`.  The reduction slots are written in full before they are read, so
// nothing has to be cleared between calls; calls that overlap on one GPU (two streams, two contexts
// or two processes' worth of threads in this process) share the array and race.
extern "C" int rt_postprocess_noalloc(rt_ctx *c, const float *src_dev, int32_t width, int32_t height, float scale, int32_t flip_y,
                                      const rt_postprocess_params *p, float *dst_dev, void *stream_) {
    const float *src = nullptr;
    if (int st = postprocess_check(c, "rt_postprocess_noalloc", src_dev, width, height, flip_y, p, dst_dev, &src)) return st;
    if ((long long)width * height > RT_NOALLOC_MAX_PIXELS)
        return fail(c, RT_ERR_INVALID_ARG, "rt_postprocess_noalloc: %d x %d is more than the %lld pixels this entry point takes; use rt_postprocess",
                    width, height, (long long)RT_NOALLOC_MAX_PIXELS);
    if (!c->post_scratch) return fail(c, RT_ERR_HIP, "rt_postprocess_noalloc: no scratch address for device %d", c->device);
    HIP_TRY(c, hipSetDevice(c->device));
    const int n_blocks = post_blocks(c, width, height);   // <= kPostScratchBlocks, which the array holds
    return postprocess_launch(c, src, width, height, scale, flip_y, p, dst_dev, c->post_scratch, n_blocks, (hipStream_t)stream_);
}
