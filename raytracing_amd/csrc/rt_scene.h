// Device-side scene tables (plain structs shared by the host table builder and the kernels).
//
// Two copies of the geometry live in HBM:
//   * float32 "filter" rows, read with wave-uniform (scalar) loads inside the brute-force loops;
//     every row carries conservative error-bound constants so that the float32 test can never
//     reject a primitive the double-precision arithmetic of the upstream code would accept;
//   * the float64 tables exactly as the host objects hold them, read only by the exact code that
//     resolves whatever the float32 filter could not decide.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace rt {

constexpr float kU = 5.9604644775390625e-08f;  // 2^-24, unit roundoff of binary32

// One live triangle, 16 dwords = one s_load_dwordx16.
// t = -(N.o + kN) / (N.d);  p = o + t d;  u = e1.p + k1;  v = e2.p + k2;  w = 1 - u - v
// (u, v are upstream's barycentrics of mesh.py:131-133 expressed at the hit point).
struct TriRow {
    float nx, ny, nz, kn;
    float e1x, e1y, e1z, k1;
    float e2x, e2y, e2z, k2;
    float cb;       // barycentric error bound = g * cb,  g = cr * |1/(N.d)|
    float ct;       // t error bound           = g * ct
    float det_lo;   // |N.d| below this can never pass upstream's |det| >= 1e-6 cut (mesh.py:123)
    int32_t gprim;  // global primitive id = n_spheres + triangle index
};

// A parallelogram row whose plane is perpendicular to a coordinate axis a (walls, floors, box faces),
// in compact form, 12 dwords: t = (c0 - o_a) / d_a needs no dot products and shares one reciprocal
// per ray and axis; (b, c) are the other two axes in increasing order.  The two functionals are
// CENTRED coordinates of the crossing point p in the parallelogram, in the first triangle's
// barycentrics (u = weight of b, v = weight of c):
//     vq = v - 1/2     = e2b p_b + e2c p_c + k2h
//     sq = u + v - 1/2 = esb p_b + esc p_c + ksh
// so that the point is inside exactly when max(|vq|, |sq|) <= 1/2 (one instruction with free
// absolute-value modifiers instead of four subtractions and two minima), and u = sq - vq names the
// triangle.  Bound constants as in TriRow, for a unit plane normal; cb also covers the rounding of
// the two shifted constants and of 1/2 - max.
struct SlabRow {
    float c0, e2b, e2c, k2h;
    float esb, esc, ksh, cb;
    float ct;
    float idf;          // (float)(id1 + 1): lets the ray-origin test be arithmetic (see slab_row_test)
    int32_t id1;        // the first triangle's id = the id the row's candidate carries
    int32_t gprim;      // id word of the parallelogram row (see kQuadFlag)
};

// An axis-aligned BOX at least four of whose faces are parallelogram rows of the scene (a room's
// walls, a lamp housing): all its faces are tested at once with one slab test - entry and exit depth
// of the ray through the box - instead of one compact row each.  8 dwords = one s_load_dwordx8.  The
// test only NOMINATES faces: a nominated face is judged from its own parallelogram row like any other
// candidate, so the box may idealise the true faces by up to DevScene::box_slack (a wall whose
// vertices wobble by 1e-6 still joins its box).  Faces are numbered code = 2 * axis + side (side 0:
// the low plane, 1: the high one); `present` has bit `code` set when the scene holds that face.  A
// face's candidate id is base + code with base = n_prims + 1 + 8 * box: ids ABOVE every primitive id,
// translated to the face's first triangle after the loops through DevScene::box_face_id.  The id
// n_prims itself means "a face of some box, which one is within the error band": every candidate of
// such a ray is then settled in double precision.
struct BoxRow {
    float lox, loy, loz, hix;
    float hiy, hiz;
    uint32_t present;
    uint32_t base;
};

// Data needed once a triangle has won: unit normal (mesh.py:140), owner, certain-accept cut.
struct TriAux {
    float nx, ny, nz;   // normalize(ab x ac), rounded from double
    float det_hi;       // |N.d| at or above this certainly passes the 1e-6 cut
    int32_t obj;        // index into scene.objects
    int32_t face;       // face index inside its mesh
    uint32_t tag;       // what a ray that starts on this triangle needs to know about it (see kTag*): the id of the first
                        // triangle of its parallelogram row, which of the row's two triangles it is, the row's box face; 0: none
    int32_t pair_row;   // index into quad_rows of the parallelogram row this triangle heads (it is the row's first), else -1
};

// One small sphere, 8 dwords = one s_load_dwordx8.
struct SphRow {
    float cx, cy, cz, r2;
    float qs;        // >= 0: |c - c_true|, how far the float32 centre lies from the double-precision one (0 when it is exact);
                     // < 0 (by-id table only): a huge sphere, evaluated in double precision
    float cn;        // |c|, for the bound on ro.d
    float r;
    int32_t gprim;   // = sphere index
};

// The same sphere as the search loop reads it (search_spheres): the discriminant in its expanded form needs the centre
// and ks = |c|^2 - r^2 (from the double-precision centre and radius, rounded once).  4 dwords (four rows come with one
// s_load_dwordx16), dead rows: ks = +inf; the sphere's id is looked up only when a lane may hit it (DevScene::sph_loop_id).
struct SphLoopRow {
    float cx, cy, cz, ks;
};

// Ray-origin tag (TriAux::tag, carried from a hit to the ray that leaves it):
//   bits 0-12   id + 1 of the first triangle of the origin's parallelogram row (0: the origin is in no such row)
//   bits 14-23  1 + 8 * box + code when that row is a face of a box (0: it is not)
//   bit  31     the origin is the row's SECOND triangle (the side u < 0 of the diagonal)
constexpr uint32_t kTagIdMask = 0x1FFFu, kTagBoxShift = 14, kTagBoxMask = 0x3FFu, kTagSecond = 0x80000000u;
constexpr int kMaxBoxes = 64;

struct MatRow {      // 8 dwords per object
    float r, g, b, emission, smoothness, transmittance, ior, r0;  // r0 = ((1-ior)/(1+ior))^2
};

struct BvhNode {     // rt_bvh_node as the device holds it (threaded depth-first order, one copy): the walk of last resort
    float lo[3]; int32_t skip;             // reads it (a lane whose stack of wide nodes ran full); inner nodes:
    float hi[3]; int32_t tri_begin_count;  // tri_begin_count < 0, first child = the next node
};

// One node of the WIDE hierarchy the kernels walk: up to four children, every child's box in the node itself, so
// one visit (eight 16-byte loads from one 128-byte line) answers four box tests and a ray needs a third of the
// dependent loads a two-child hierarchy costs it.  Boxes are stored as centre and half-extent per axis: with
// tc = c / d - o / d and th = h / |d| the ray is inside the slab for tc - th <= t <= tc + th whatever the sign of d,
// so one copy of the nodes serves every ray direction (no per-octant copies, no minimum / maximum per axis).
// [c - h, c + h] contains the child's true box (the host rounds h up to cover the rounding of c).  An empty slot has
// h = -1e30: its entry lies beyond its exit for every ray.
//   link[k]   a LEAF child's triangle range (first row << 4 | count, rows of DevScene::leaf_rows); unused for inner children
//   order[2]  for each octant of ray directions one byte (octant & 3 within word octant >> 2): four 2-bit slot
//             numbers, the children from near to far for rays of that octant
//   first     index of the node's first INNER child; its inner children are stored one after the other in slot order,
//             so child k is node first + popcount(inner & ((1 << k) - 1)): no table of links to index per lane
//   inner     bits 0-3: which slots hold inner nodes; bits 4-7: which slots hold anything
struct WideNode {
    float cx[4], hx[4], cy[4], hy[4], cz[4], hz[4];
    int32_t link[4];
    uint32_t order[2]; int32_t first; uint32_t inner;
};
static_assert(sizeof(WideNode) == 128, "WideNode: one 128-byte line");
// A wave's share of the walk's LDS: kWalkColumn rows of 64 dwords, a column per lane.
#ifndef RT_WALK_STACK
#define RT_WALK_STACK 20
#endif
constexpr int kWalkStack = RT_WALK_STACK;   // rows 0 .. : a lane's stack of wide nodes still to visit
constexpr int kWalkPending = 8;             // then: leaves a lane holds until their triangles are tested
constexpr int kWalkColumn = kWalkStack + kWalkPending;

// Scene data only the rare double-precision code, the depth-of-field camera and the ray-generation
// kernel read.  It lives in device memory behind DevScene::cold, so that the render kernel does not
// pin ~50 scalar registers on it for the whole launch (which the compiler paid for by spilling
// scalars into vector-register lanes inside the hot loop).
struct ColdScene {
    const double *tri64;      // [n_tris][9] vertices as the host holds them
    const double *sph64;      // [n_spheres][4], then [n_spheres] radius**2 as the host's C library computes it (exact_sphere)
    const double *aabb64;     // [n_meshes][6]
    const int32_t *mesh_tri_begin;  // [n_meshes+1]
    const int32_t *obj_kind, *obj_index;
    int32_t n_objects, n_tris, n_meshes, pad;
    double cam64[15];         // pos, right, up, forward (3 each), d, dof_strength, dof_dist
};

struct DevScene {
    // Members are ordered by where the render loop reads them, so that the members one stage needs
    // sit in one run of 16 dwords and reach the scalar registers with one load (see hot_scene_*).
    // ---- first pass (16 dwords): the float32 filter tables and their sizes
    const SphRow *sph_rows;                           // small spheres only
    // First passes that take one candidate per parallelogram walk quad_rows[0, n_quad_general) and, for
    // the rows behind those - the ones in axis-perpendicular planes - the compact table instead:
    const SlabRow *slab_rows;
    const TriRow *quad_rows;                          // parallelogram rows (two triangles each), padded like tri_rows
    const TriRow *tri_rows;                           // live triangles that head no parallelogram row, padded with dead rows:
    int32_t n_sph_rows, n_big_sph;                    // n_big_sph: spheres evaluated in float64 (indices in big_sph)
    uint32_t slab_counts;     // rows per axis: x | y << 10 | z << 20, stored in that order, each axis an even count, one dead row at the end
    int32_t n_quad_general;
    int32_t n_quad_rows_padded;
    int32_t n_tri_rows_padded;                        //   even, and 2 more rows exist past it
    uint32_t key_mask;        // low bits of a hit key that hold the primitive id
    float scene_radius;       // bound on the distance from the world origin of every vertex, small-sphere point and the camera
    // ---- winner, hit record, shading (16 dwords)
    const TriAux *tri_aux;                            // [n_tris]
    const TriRow *tri_by_id;                          // [n_tris] same rows, indexed by triangle id
    const SphRow *sph_by_id;                          // [n_spheres]
    const int32_t *sph_obj;                           // [n_spheres]
    const MatRow *materials;                          // [n_objects]
    const ColdScene *cold;                            // device memory
    int32_t n_spheres;
    int32_t n_prims;          // n_spheres + number of triangles: candidate ids are clamped below it before they index a table
    int32_t n_tri_rows;       // live triangles the linear tables stand for (a parallelogram row counts two)
    int32_t dof_on;
    // ---- camera, environment
    // camera: float32 position and its double-precision residue (pinhole rays come from k_raygen's table)
    float cam_pos[3], cam_res[3];
    const uint32_t *sky; int32_t sky_w, sky_h; float sky_scale;   // environment: RGBA8 texels, row-major
    float cam_rko;            // 1.01 u when the camera position is not float32-representable (cam_res != 0), else 0: see Ray::rko
    // ---- boxes (see BoxRow)
    const BoxRow *box_rows;        // [n_boxes]
    const int32_t *box_face_id;    // [8 * n_boxes] by 8 * box + code: global id of the face's first triangle, -1 = no such face
    const TriRow *box_face_rows;   // [8 * n_boxes] by 8 * box + code: the face's parallelogram row (a dead row where there is none)
    int32_t n_boxes;
    float box_slack;               // how far (per coordinate) a point of a true face may lie from the box's ideal face
    float box_tilt;                // bound on the sine of the angle between a true face's normal and its axis
    int32_t pad1;
    // ---- the few huge spheres, the hierarchy
    const int32_t *big_sph;                           // sphere indices evaluated in float64
    const TriRow *leaf_rows;                          // hierarchy: triangle rows in leaf order
    const BvhNode *bvh_nodes; int32_t n_bvh_nodes;    // threaded two-child hierarchy over leaf_rows (the walk of last resort); 0 = none
    int32_t walk_stack;                               // stack entries a lane may use (kWalkStack; tests set it lower)
    const WideNode *wide_nodes;                       // the same hierarchy with up to four children per node: what the kernels walk
    // Per plane, what the float32 row's rounding and the order of its multiply-adds can cost: four floats (ax, ay, az, ak)
    // per triangle (by triangle id), then per parallelogram row (by row index).  See judge_tri.
    const float *tri_tight;
    const SphLoopRow *sph_loop;                       // the small spheres again, in sph_rows' order, as the search loop reads them
    const int32_t *sph_loop_id;                       // sphere index of every row of sph_loop
};
static_assert(offsetof(DevScene, tri_aux) == 64 && offsetof(DevScene, cam_pos) == 128, "DevScene: the two hot runs are 16 dwords each");

// Device code reads the scene where the launch put it: the kernel-argument block, which is constant
// memory.  A member read through this reference is a scalar load at the point of use, not a scalar
// register held for the whole launch (DevScene is ~55 dwords; held in registers they push other
// scalars into vector-register lanes, and every v_readlane that brings one back is a vector
// instruction).  Every kernel takes the scene as its FIRST by-value argument, so it sits at offset 0.
typedef const DevScene __attribute__((address_space(4))) &SceneRef;
typedef const DevScene __attribute__((address_space(4))) *ScenePtr;
#if defined(__HIPCC__)
__device__ __forceinline__ ScenePtr kernarg_scene() { return (ScenePtr)__builtin_amdgcn_kernarg_segment_ptr(); }
#endif

// Division of a 32-bit unsigned by a launch-invariant divisor (Granlund & Montgomery): the host fills
// the constants, the device spends a multiply-high and two shifts instead of a ~30-instruction divide.
struct FastDiv { uint32_t d, m, sh1, sh2; };

// A path the lean render kernel could not finish (one of its closest-hit queries needs the double-precision
// settling code, which that kernel does not hold): the state of the path in front of that query, so that the
// full kernel can go on from there, and - filled in by the full kernel - what the path contributes.  96 bytes.
struct AsideRec {
    uint32_t pixel, sample;   // the path: x * H + y, global sample index (they key its random stream)
    uint32_t slot;            // the job (pixel of the rectangle, chunk of samples) the path belongs to
    int32_t next;             // the record the same job set aside before this one (list from RenderArgs::head), -1 = none
    float r, g, b;            // the path's radiance (throughput x terminal radiance): written by the full kernel
    uint32_t depth;           // index of the path segment whose query was handed back
    float ox, oy, oz, dx, dy, dz;   // that segment's ray
    float tr, tg, tb;         // throughput in front of it
    int32_t origin;           // primitive the ray starts on, -1 = camera
    uint32_t tag;             // TriAux::tag of that primitive
    float own_ratio;          // Hit::own_ratio of the hit the ray leaves from
    uint32_t inside;          // Ray.inside
    uint32_t pad[3];          // hierarchy scenes: the first pass's result for that query (PairBest::lo0, lo1, id0 as bits),
                              // so the full kernel judges and settles it without walking it again
};
static_assert(sizeof(AsideRec) == 96, "AsideRec: six 16-byte loads");

struct RenderArgs {
    int32_t width, height, x0, y0, rect_w, rect_h;
    // Camera-ray table built by k_raygen for this (camera, width, height): per pixel (x * height + y)
    // the float32 pinhole direction and its double-precision residue, 16 bytes each.
    const float *ray_dir, *ray_res;
    int32_t sample_begin, sample_count, chunk_samples, n_chunks;
    int32_t max_bounces;
    uint32_t seed_lo, seed_hi;
    uint32_t total_jobs;
    FastDiv div_chunks, div_rect_h;    // job -> (pixel in rectangle, chunk) -> (x, y): jobs are numbered pixel by pixel
    uint32_t *job_counter;    // device word, zero at launch
    float *accum;             // (W,H,3) running sum
    double *partial;          // [rect_w*rect_h][n_chunks][3]: every job's sum over its samples
    // paths set aside by the lean kernel, and jobs it gave up (no room left in the list): see k_render
    AsideRec *aside; uint32_t *aside_count; uint32_t aside_cap;
    uint32_t *dirty_list, *dirty_count; uint8_t *dirty;   // [total_jobs] job ids, their number, one byte per job
    int32_t *head;            // [total_jobs] the record each job set aside last, -1 = none
    const uint32_t *job_list, *job_list_count;   // when set: the jobs to run are job_list[0, *job_list_count) instead of [0, total_jobs)
    int32_t path_mode;        // full kernel only: the jobs are the records aside[0, min(*job_list_count, aside_cap)), one path each
    unsigned long long *stats;  // kStat* counters
    unsigned long long *prof;   // section-timer totals (diagnostic builds with RT_SECTION_TIMERS), else null
#ifdef RT_TUNE
    int32_t tune[4];            // tuning builds only: walk slice, lanes-through threshold for one more slice, how many more
#endif
};

enum { kStatRays = 0, kStatPaths, kStatSphTests, kStatTriTests, kStatExact, kStatRefine, kStatSky, kStatShaded,
       kStatBvhNodes, kStatCount };

}  // namespace rt
