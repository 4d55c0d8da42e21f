/*
 * raytracing_amd.h — C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * The upstream renderer (m-J-K-b/raytracing) is pure Python and has no FFI of its own; the seam
 * this library sits behind is its Python object API.  Every entry point below therefore names the
 * upstream function whose work it takes over (file:line under the upstream tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only, no C++ or torch types.
 *   - Every function returns an rt_status (0 = ok).  Nothing throws across the boundary; the text
 *     of the last failure on a context is available from rt_last_error().
 *   - The caller owns every array it passes in; the library copies what it needs during the call.
 *     Device memory allocated by the library lives until rt_destroy().
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP device memory of the
 *     context's GPU.  `stream` arguments are a hipStream_t passed as void* (NULL = default stream).
 *   - Image layout follows the upstream accumulation array RenderResult.arr
 *     (src/raytracing/render_result.py:14): shape (W, H, 3), x-major, y up, un-normalised SUM of
 *     per-sample radiance; element (x, y, c) sits at ((x * H) + y) * 3 + c.
 *   - One render call in flight per context; rt_request_stop / rt_get_stats / rt_last_error may be
 *     called from other threads while it runs.
 */
#ifndef RAYTRACING_AMD_H
#define RAYTRACING_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = 1,   /* NULL pointer, negative count, inconsistent table sizes, ... */
    RT_ERR_NO_SCENE = 2,      /* render / intersect before rt_set_scene */
    RT_ERR_HIP = 3,           /* a HIP runtime call failed; see rt_last_error */
    RT_ERR_NO_DEVICE = 4,     /* no usable GPU / bad device ordinal */
    RT_ERR_STOPPED = 5,       /* rt_request_stop was raised before or during the call */
    RT_ERR_UNSUPPORTED = 6
} rt_status;

enum { RT_OBJ_SPHERE = 0, RT_OBJ_MESH = 1 };

/*
 * Flat description of one scene: what the host packer extracts from the upstream objects
 *   scene.objects[i].material.{color,emission_strength,smoothness,transmittance,ior}  (src/core/material.py:8-21)
 *   Sphere.origin / Sphere.radius                                                      (src/mesh/sphere.py:9-11)
 *   Mesh.transformed_vertices / Mesh.triangles / Mesh.bounding_box                     (src/mesh/mesh.py:25-59)
 *   scene.camera.{pos,right,up,forward,d,dof_strength,dof_dist}                        (src/raytracing/camera.py:9-37)
 *   the scene module's sky_map + its scale factor                                      (scenes/room.py:15-22)
 * All geometry is double precision exactly as the Python objects hold it; the library derives its
 * own float32 working copies.  Object order is scene.objects order and defines hit tie-breaks
 * (src/raytracing/scene.py:23-27: stable sort by depth over objects, then per-object hit order).
 */
typedef struct rt_scene_desc {
    uint32_t struct_size;          /* = sizeof(rt_scene_desc), checked */
    int32_t n_objects;
    const int32_t *obj_kind;       /* [n_objects] RT_OBJ_SPHERE | RT_OBJ_MESH */
    const int32_t *obj_index;      /* [n_objects] index into spheres[] or meshes */
    const double *materials;       /* [n_objects][7] r,g,b, emission_strength, smoothness, transmittance, ior */

    int32_t n_spheres;
    const double *spheres;         /* [n_spheres][4] cx,cy,cz,radius */
    const int32_t *sphere_obj;     /* [n_spheres] owning object index */

    int32_t n_meshes;
    const double *mesh_aabb;       /* [n_meshes][6] min xyz, max xyz  (Mesh.bounding_box) */
    const int32_t *mesh_obj;       /* [n_meshes] owning object index */
    const int32_t *mesh_tri_begin; /* [n_meshes+1] first triangle of each mesh in tri_verts */
    int32_t n_tris;
    const double *tri_verts;       /* [n_tris][9] a,b,c (transformed vertices of each face, face order) */

    double cam_pos[3], cam_right[3], cam_up[3], cam_forward[3];
    double cam_d, cam_dof_strength, cam_dof_dist;

    int32_t sky_w, sky_h;          /* 0,0 => black environment (Scene.get_environment default, scene.py:35-36) */
    const uint8_t *sky_rgb;        /* [sky_h][sky_w][3] row-major, as decoded from the image file */
    double sky_scale;              /* radiance = texel/255 * sky_scale */

    /* Optional flat bounding-volume hierarchy over ALL triangles (no upstream counterpart: it
     * replaces the linear loops of Mesh.intersect, src/mesh/mesh.py:94-107, and returns the same
     * hits).  n_bvh_nodes > 0: the caller's hierarchy (checked: links, coverage, containment);
     * n_bvh_nodes == 0: linear table walk up to 48 triangles (and 4096 primitives), a hierarchy built
     * by the library beyond that; n_bvh_nodes < 0: the library builds one whatever the size.
     * Layout: rt_bvh_node. */
    int32_t n_bvh_nodes;
    const void *bvh_nodes;         /* [n_bvh_nodes] rt_bvh_node */
    const int32_t *bvh_tri_index;  /* [n_tris] triangle ids in leaf order */

    /* Layout switches for A/B runs, tests and tuning; all zero = the defaults.  Results never depend
     * on them: every layout returns the same hits.  (No upstream counterpart.) */
    uint32_t flags;                /* RT_SCENE_* */
    int32_t linear_max_tris;       /* triangle count up to which the linear tables are kept (0 = 48; at most 4096) */
    int32_t bvh_leaf_size;         /* leaf target of a hierarchy built by the library (0 = 2) */
    int32_t build_threads;         /* threads of that build (0 = hardware threads, at most 16) */
} rt_scene_desc;

enum {
    RT_SCENE_NO_PAIR_ROWS = 1,     /* every triangle a table row of its own (no parallelogram rows) */
    RT_SCENE_NO_COMPACT_ROWS = 2,  /* no compact rows for parallelograms in axis-perpendicular planes */
    RT_SCENE_NO_LINEAR_SPLIT = 4,  /* hierarchy scenes: scene-sized triangles go into the hierarchy too */
    RT_SCENE_TINY_WALK_STACK = 8,  /* hierarchy scenes: a walk may stack six nodes instead of 24, so that most walks fall back to
                                    * the stackless walk of last resort (tests of that walk) */
    RT_SCENE_NO_BOX_ROWS = 16,     /* no box rows (axis-aligned boxes whose faces are tested with one slab test) */
    /* Moved meshes (upstream: Mesh.update_transformed_vertices, src/mesh/mesh.py:178-209, which rebuilds
     * nothing because it has no acceleration structure): when the context's current hierarchy was built
     * by rt_set_scene for the same meshes (same triangle counts per mesh, same leaf size), keep its
     * topology and leaf order and only refit the boxes to the new vertices; every table is uploaded
     * again.  Hits are those of a rebuilt hierarchy (any valid hierarchy gives the same hits); a refitted
     * one may be slower to walk after large deformations.  Ignored when there is nothing to refit. */
    RT_SCENE_REFIT = 32
};

/* Depth-first ("threaded") BVH node: if the ray misses the box, or after a leaf has been
 * processed, traversal continues at `skip`; otherwise at the next node in memory. */
typedef struct rt_bvh_node {
    float lo[3];
    int32_t skip;                  /* index of the node to continue with on a miss; -1 = done */
    float hi[3];
    int32_t tri_begin_count;       /* leaf: (first << 4) | count (count 1..15), first indexes bvh_tri_index; inner: -1 */
} rt_bvh_node;

typedef struct rt_render_params {
    uint32_t struct_size;
    int32_t width, height;         /* RenderSettings.WIDTH/HEIGHT (render_settings.py:15-16) */
    int32_t x0, x1, y0, y1;        /* pixel rectangle, as the `area` tuple of render_area (renderer.py:90-91) */
    int32_t sample_begin;          /* global index of the first sample (pass) of this call */
    int32_t sample_count;          /* number of samples (passes) to add per pixel */
    int32_t max_bounces;           /* RenderSettings.MAX_BOUNCES (render_settings.py:19) */
    uint64_t seed;                 /* mixed into the Philox key with (pixel, sample) */
    uint32_t flags;                /* RT_RENDER_*; 0 = defaults */
    int32_t wavefront_paths;       /* RT_RENDER_WAVEFRONT: paths in flight per batch (0 = 16 Mi) */
} rt_render_params;

enum {
    /* Hierarchy scenes: the wavefront form of the render (generate / extend / shade / resolve kernels
     * over path queues in memory) instead of the single persistent kernel. */
    RT_RENDER_WAVEFRONT = 1,
    /* The default since ABI version 3, kept as a name: a float32 hit depth is accepted only when its error
     * BOUND is below 9.5e-6 of the depth, as rt_intersect_batch does; otherwise it is redone in double
     * precision (measured against upstream's doubles: every depth within 1e-6 relative).  Hit identity is
     * exact in every mode. */
    RT_RENDER_STRICT_DEPTH = 2,
    /* Render with the one kernel that holds all the code.  Without the flag rt_render runs a lean kernel
     * that sets the few paths needing double-precision settling aside, traces those again with the full
     * code and folds them in: same paths, same image bit for bit, faster.  For A/B runs and tests. */
    RT_RENDER_NO_DEFER = 4,
    /* For A/B runs: the looser depth acceptance of earlier versions - error bounds up to 2e-4 of the depth,
     * or 1e-5 of the scene size as a distance; about 2 in 10,000 depths then differ from upstream's by more
     * than 1e-5 relative (grazing hits and very short rays), none by more than the bound. */
    RT_RENDER_LOOSE_DEPTH = 8
};

typedef struct rt_stats {
    uint64_t rays;                 /* closest-hit queries = Scene.intersect calls (scene.py:23) */
    uint64_t paths;                /* top-level raytrace calls (renderer.py:105) */
    uint64_t sphere_tests, tri_tests, aabb_tests, bvh_node_tests; /* float32 filter tests executed */
    uint64_t exact_rays;           /* rays whose contenders were settled in double precision (second pass or whole-scene scan) */
    uint64_t refined_hits;         /* hits whose winning primitive was re-evaluated in double precision.  (These two are
                                      diagnostics: when rt_render gives a job up and renders it again - its list of set-aside
                                      paths was full - what the job had counted here before is counted twice.) */
    uint64_t sky_lookups, shaded_hits;
    double kernel_ms;              /* device time of the call's kernels (rt_render: everything the call launches - the render
                                      kernel, the launches over set-aside paths and given-up jobs, the fold), summed over launches */
    uint64_t launches;
    double kernel_ms_min, kernel_ms_max;   /* shortest and longest single launch (0 without launches): shows an outlier a mean hides */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* Library / lifetime ----------------------------------------------------------------------- */
int rt_abi_version(void);
/* Create a context bound to HIP device `device_id`.  Takes over nothing upstream (the upstream
 * renderer has no device); pairs with Renderer.__init__ (renderer.py:15-19). */
int rt_create(int device_id, rt_ctx **out);
int rt_destroy(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx); /* ctx may be NULL: last error of a failed rt_create */

/* Host-side hierarchy builder (binned surface-area heuristic; CPU only, needs no context or GPU).
 * No upstream counterpart (upstream has no acceleration structure).  tri_verts = [n_tris][9] as in
 * rt_scene_desc; leaf_size = triangles per leaf to aim for (<= 0: 2).  Writes the leaf-order
 * triangle ids to tri_index_out[n_tris], the node count to *n_nodes_out and, when nodes_out is
 * non-NULL and nodes_capacity suffices (2 * n_tris always does), the nodes.  Call with
 * nodes_out = NULL to learn the count first. */
int rt_build_bvh(int32_t n_tris, const double *tri_verts, int32_t leaf_size, rt_bvh_node *nodes_out,
                 int32_t nodes_capacity, int32_t *tri_index_out, int32_t *n_nodes_out);
/* Threads rt_build_bvh uses (process-wide; 0 = hardware threads, at most 16).  The result does not
 * depend on it. */
int rt_build_bvh_threads(int32_t n_threads);

/* The FOUR-WIDE form of a threaded hierarchy, as rt_set_scene makes it for the kernels (CPU only; tests and
 * tooling; no upstream counterpart).  A node is 128 bytes = 32 words:
 *   [0..23]  float: centre and half-extent of each child's box, per axis: cx[4] hx[4] cy[4] hy[4] cz[4] hz[4]
 *            ([c - h, c + h] contains the child's box; an empty slot has h = -1e30);
 *   [24..27] int32: a LEAF child's triangle range (first << 4 | count), -1 for an inner child;
 *   [28,29]  per octant of ray directions (bit k = direction negative along axis k) one byte of four 2-bit slot
 *            numbers, near to far: octant o is byte o & 3 of word 28 + (o >> 2);
 *   [30]     int32: index of the first inner child; inner children follow one another in slot order;
 *   [31]     bits 0-3: slots holding inner nodes, bits 4-7: slots holding anything.
 * `nodes` must be a valid hierarchy in rt_build_bvh's layout.  Writes the node count to *n_wide_out, the stack
 * entries the deepest walk can need to *stack_need_out and, when wide_out is non-NULL and wide_capacity suffices
 * (n_nodes always does), the nodes. */
int rt_collapse_bvh(int32_t n_nodes, const rt_bvh_node *nodes, void *wide_out, int32_t wide_capacity, int32_t *n_wide_out,
                    int32_t *stack_need_out);

/* Host-side Wavefront OBJ reader with the semantics of Mesh.load_from_obj_file
 * (src/mesh/mesh.py:211-251): one object per "o" record, first index of each a/b/c corner, face
 * indices rebased by the vertex count of earlier objects, fan triangulation.  CPU only.
 * rt_obj_open parses the file and reports the sizes; rt_obj_read copies into caller arrays:
 * names[n_objects][name_stride] (NUL-terminated, truncated), vertex_begin / tri_begin[n_objects+1],
 * vertices[n_vertices][3], triangles[n_triangles][3] (indices local to their object); any pointer
 * may be NULL.  On failure rt_last_error(NULL) has the reason. */
typedef struct rt_obj rt_obj;
int rt_obj_open(const char *path, rt_obj **out, int32_t *n_objects, int64_t *n_vertices, int64_t *n_triangles);
int rt_obj_read(const rt_obj *obj, char *names, int32_t name_stride, int64_t *vertex_begin, int64_t *tri_begin,
                double *vertices, int32_t *triangles);
int rt_obj_close(rt_obj *obj);

/* Scene upload: what Renderer.init_render binds (renderer.py:81-88, `self.scene = ...SCENE`). */
int rt_set_scene(rt_ctx *ctx, const rt_scene_desc *scene);

/* Closest-hit query for n arbitrary rays: Scene.intersect(ray)[0] (scene.py:23-27) with
 * Sphere.intersect (sphere.py:13-41), Mesh.intersect / intersect_aabb / intersect_tri
 * (mesh.py:61-143).  Host arrays.  obj = index into scene.objects or -1; prim = face index inside
 * the mesh, or 0/1 = near/far root for a sphere; t, normal (unit, as the object returns it, not
 * flipped), pos are doubles.  `origin_prim` (may be NULL) marks rays that start ON a primitive as
 * bounce rays do: value = global primitive id as returned in `gprim` (sphere index, or n_spheres +
 * triangle index), -1 = none; NULL gives upstream's semantics unchanged.  `how` (may be NULL)
 * reports which code produced each answer: 0 = float32 filter alone, 1 = winning primitive
 * re-evaluated in double precision, 2 = every contender (or the whole scene) re-evaluated in double
 * precision. */
int rt_intersect_batch(rt_ctx *ctx, int64_t n, const double *org, const double *dir,
                       const int32_t *origin_prim, int32_t *obj, int32_t *prim, int32_t *gprim,
                       int32_t *how, double *t, double *normal, double *pos);

/* Renderer.raytrace(ray, depth=first_depth) (renderer.py:21-79) for n caller-supplied rays: the
 * radiance carried back along the path that starts with each ray, i.e. upstream's return value for
 * incoming_light = 0 and ray_color = 1.  Host arrays; org / dir are doubles as the Python objects
 * hold them and are rounded to float32, which is then the ray that is traced.  inside (may be NULL
 * = all false) is Ray.inside (ray.py:10).  first_depth > max_bounces yields 0 (renderer.py:29-30).
 * Ray i draws its random numbers from the stream of (pixel = stream_begin + i, sample) under `seed`,
 * the same keying rt_render uses, segment counter = depth + 1. */
int rt_raytrace_batch(rt_ctx *ctx, int64_t n, const double *org, const double *dir, const uint8_t *inside,
                      int32_t first_depth, int32_t max_bounces, uint64_t seed, uint32_t sample,
                      uint32_t stream_begin, float *rgb);

/* Accumulation buffer: RenderResult.arr (render_result.py:14,24-27).  The context owns one
 * float32 (W,H,3) device buffer, (re)allocated by rt_accum_resize and zeroed. */
int rt_accum_resize(rt_ctx *ctx, int32_t width, int32_t height);
int rt_accum_clear(rt_ctx *ctx);
void *rt_accum_device_ptr(rt_ctx *ctx);                 /* for zero-copy tensor hand-off; NULL if none */
int rt_accum_read(rt_ctx *ctx, float *host_w_h_3);      /* blocking copy of the whole buffer */

/* The per-pixel render path: Renderer.render_area (renderer.py:90-106) for `sample_count` passes
 * = Camera.get_ray (camera.py:39-50) + Renderer.raytrace (renderer.py:21-79) + environment lookup
 * (scenes/room.py:15-22) + RenderResult.add_to_px (render_result.py:24-27), for every pixel of the
 * rectangle.  Adds into `accum_dev` (device, float32 (W,H,3)); NULL = the context's own buffer.
 * Asynchronous on `stream`; rt_synchronize waits.  One call enqueues several kernels (see
 * RT_RENDER_NO_DEFER) that share context-owned scratch memory and counters: calls on one context go to
 * ONE stream, or are ordered by the caller (events) if they use several; contexts are independent.
 * Scratch: a call needs, in context-owned device memory that only ever grows, 33 bytes per JOB (a job is a pixel
 * and a run of at least eight of its samples: at most rectangle x sample_count / 8 jobs, about 13 per pixel for
 * calls of 256 samples at 1080p) plus 96 bytes per set-aside path for up to 1/64 of the call's paths (at least
 * 4,096, at most 2^25 records), plus once per image size 48 bytes per pixel of camera rays: 1.4 GB for 1920 x 1080
 * x 256 samples in one call, 0.5 GB for the same image in calls of 32 samples.  Several contexts on one GPU each
 * hold their own; rt_release_scratch gives it back. */
int rt_render(rt_ctx *ctx, const rt_render_params *params, float *accum_dev, void *stream);
int rt_synchronize(rt_ctx *ctx, void *stream);
/* Frees the context's grow-only scratch (rt_render's job slots and set-aside lists, the wavefront form's queues, the
 * staging of rt_intersect_batch and rt_postprocess); scene tables, accumulation buffer and camera-ray table stay.
 * Waits for the device first; later calls allocate what they need again.  (No upstream counterpart.) */
int rt_release_scratch(rt_ctx *ctx);

/* Merge across GPUs.  Upstream has one process and one buffer (render_result.py:24-27: every tile
 * adds into the same arr); with one process per GPU, each rendering its share of the samples into
 * its own buffer, the same sum is ONE all-reduce of the (W,H,3) float32 buffers - RCCL over xGMI,
 * issued by this library (PyTorch is not needed for it).
 *   rt_comm_unique_id : rank 0 makes the 128-byte id and hands it to the other ranks by any means
 *                       (a file, MPI, torch.distributed.broadcast, ...).  Needs no context.
 *   rt_comm_init      : every rank, once per context, with the context's GPU as its RCCL device.
 *                       n_ranks = 1 with a NULL id makes rt_merge a no-op without opening RCCL.
 *   rt_merge          : in-place sum over the ranks of accum_dev (NULL = the context's buffer),
 *                       asynchronous on `stream`; every rank must call it with the same size.
 * librccl.so is opened on first use; RT_ERR_UNSUPPORTED when it cannot be. */
#define RT_COMM_ID_BYTES 128
/* RT_OK when librccl.so can be opened and holds what rt_comm_* need, RT_ERR_UNSUPPORTED otherwise (reason:
 * rt_last_error(NULL)).  No upstream counterpart; needs no context and no GPU.  Lets every rank find out
 * BEFORE the collective rt_comm_init whether all of them can enter it. */
int rt_comm_available(void);
int rt_comm_unique_id(void *out_128_bytes);
int rt_comm_init(rt_ctx *ctx, int32_t n_ranks, int32_t rank, const void *unique_id_128_bytes);
int rt_merge(rt_ctx *ctx, float *accum_dev, int32_t width, int32_t height, void *stream);

/* PostProcessing.process (src/post_processing/post_processing.py:35-82) on the device, fused into two
 * passes over the image: exposure (:47), contrast / brightness with its clip (:56), saturation (:58-60),
 * global Reinhard tone mapping (:62-79; key = its `a`, tone_saturation = its `saturation`) and gamma
 * as img ** gamma (:82).  src_dev (NULL = the context's accumulation buffer) and dst_dev are
 * (W,H,3) float32 device buffers; every source value is first multiplied by `scale` (1 /
 * rendered_passes turns the accumulated sum into RenderResult.img_arr's values), and flip_y = 1
 * reads row H-1-y for output row y, which is img_arr's orientation (render_result.py:46).
 * Asynchronous on `stream`. */
typedef struct rt_postprocess_params {
    uint32_t struct_size;
    float exposure, brightness, contrast, saturation, gamma;   /* PostProcessing.__init__ (:21-33) */
    float key, tone_saturation;                                /* reinhardt_tone_mapping defaults: 0.18, 1.0 (:62-64) */
} rt_postprocess_params;
int rt_postprocess(rt_ctx *ctx, const float *src_dev, int32_t width, int32_t height, float scale, int32_t flip_y,
                   const rt_postprocess_params *params, float *dst_dev, void *stream);

/* rt_postprocess for callers that cannot allocate.  No counterpart upstream: it exists for code under
 * HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same arguments, same
 * kernels and bit-identical output; the per-block reduction slots live in a fixed device array inside
 * the library (32,784 bytes per GPU, held until the process ends) instead of a context buffer, so the
 * call makes no hipMalloc / hipFree, no synchronisation and no allocating library call - three kernel
 * launches on `stream`, nothing else.  Images of more than RT_NOALLOC_MAX_PIXELS pixels are refused
 * with RT_ERR_INVALID_ARG (use rt_postprocess).  All calls on one GPU share the one array: calls that
 * may overlap there (different streams, different contexts of the same GPU) race on it and must be
 * ordered by the caller.  Nothing in the array needs to be zero between calls. */
#define RT_NOALLOC_MAX_PIXELS (1 << 26)
int rt_postprocess_noalloc(rt_ctx *ctx, const float *src_dev, int32_t width, int32_t height, float scale, int32_t flip_y,
                           const rt_postprocess_params *params, float *dst_dev, void *stream);

/* Renderer.stop_render (renderer.py:149-150): later rt_render calls return RT_ERR_STOPPED until
 * rt_clear_stop. */
int rt_request_stop(rt_ctx *ctx);
int rt_clear_stop(rt_ctx *ctx);

int rt_get_stats(rt_ctx *ctx, rt_stats *out);   /* cumulative since rt_reset_stats; synchronises */
int rt_reset_stats(rt_ctx *ctx);

/* Known-answer hook for the counter-based RNG the kernels use (no upstream counterpart: upstream
 * draws from global MT19937 streams, renderer.py:62,64, vec3.py:14-16, camera.py:46-47).
 * out[5*s + i] = uniform i (0..4) of segment s of path (pixel, sample) under `seed`, computed ON THE
 * DEVICE: one Philox4x32-10 block per path segment, five 24-bit fields each. */
int rt_debug_uniforms(rt_ctx *ctx, uint32_t pixel, uint32_t sample, uint64_t seed, int32_t n_segments, float *out_host);

/* How rt_set_scene laid the current scene's triangles out (no upstream counterpart; tests and tuning):
 * out[0] parallelogram rows (padded), [1] of which in general position, [2..4] compact rows per axis
 * x, y, z (padded to even), [5] single-triangle rows (padded), [6] boxes, [7] hierarchy nodes;
 * n_out >= 8; with n_out >= 9, out[8] = 1 when the last rt_set_scene refitted the hierarchy (RT_SCENE_REFIT);
 * with n_out >= 10, out[9] = the stack entries the deepest walk of the wide hierarchy can ask for.
 * *box_slack (optional) = the largest distance of a true box face from its ideal one. */
int rt_debug_scene_layout(rt_ctx *ctx, int32_t *out, int32_t n_out, float *box_slack);

/* Hierarchy scenes (no upstream counterpart; tests and tuning): how many walks ran out of stack and were redone by the
 * stackless walk of last resort since the library was loaded on this context's GPU. */
int rt_debug_walk_fallbacks(rt_ctx *ctx, uint64_t *out);

/* What the last rt_render call set aside (no upstream counterpart; tests and tuning; synchronises):
 * out4[0] the job hand-out counter, [1] paths the lean kernel set aside for the full code, [2] jobs it gave
 * up (no room left in the list), [3] the hand-out counter of the launch that renders those jobs again.  [0] and [3]
 * are the counters' final values: every wave overshoots by the window it asked for last, so [3] is non-zero even
 * when no job was given up. */
int rt_debug_render_counters(rt_ctx *ctx, uint32_t *out4);

/* Parity / diagnostic hook: traces n_paths whole paths, one device thread each, with the same
 * device functions the render kernel runs (Camera.get_ray + Renderer.raytrace, camera.py:39-50,
 * renderer.py:21-79), and records every segment.  xys = [n_paths][3] (x, y, sample index).  Outputs
 * are host arrays; per-segment arrays hold max_bounces + 1 entries per path: hit object / primitive
 * (-1 = miss), which code produced the hit (see rt_intersect_batch), depth, ray origin, direction.
 * render_flags: 0 or RT_RENDER_LOOSE_DEPTH, as rt_render takes it.  seg_org_residue (optional): a ray that leaves a hit
 * the double-precision code produced starts at that hit's double-precision position (renderer.py:73-76); seg_org holds
 * its float32 rounding and seg_org_residue what that rounding dropped (zero for every other ray). */
int rt_debug_trace_paths(rt_ctx *ctx, int32_t n_paths, const int32_t *xys, int32_t width, int32_t height,
                         int32_t max_bounces, uint64_t seed, uint32_t render_flags, float *rgb, int32_t *n_seg, int32_t *seg_obj,
                         int32_t *seg_prim, int32_t *seg_how, float *seg_t, float *seg_org, float *seg_dir, float *seg_org_residue);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACING_AMD_H */
