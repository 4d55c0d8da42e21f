// Host-side builder of the flat bounding-volume hierarchy over triangles (binned surface-area
// heuristic, top-down), and the checks rt_set_scene runs on a hierarchy before trusting it.
//
// No upstream counterpart: upstream tests every face of every mesh for every ray
// (src/mesh/mesh.py:94-107).  The hierarchy only decides WHICH triangles the kernels test; it must
// never hide one a ray could hit, so every box contains its triangles (boxes are rounded outward
// to float32 here and padded again on upload by the kernels' own rounding allowance).
//
// Layout ("threaded" depth-first order): the children of an inner node i are node i + 1 and the
// node that follows the first child's subtree; `skip` is the node that follows node i's whole
// subtree (-1 after the last one).  Traversal needs no stack: on a box hit go to i + 1 (inner) or
// test the leaf's triangles and go to skip; on a miss go to skip.  skip > i always, so every walk
// terminates.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracing_amd.h"
#include "rt_scene.h"

namespace rt {
namespace bvh {

inline float round_down(double v) { float f = (float)v; return (double)f > v ? std::nextafterf(f, -INFINITY) : f; }
inline float round_up(double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, INFINITY) : f; }

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; } }
    void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const float *p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    double half_area() const {
        double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Builder {
    static constexpr int kBins = 16;          // (8 / 16 / 32 / 64 bins measured the same render time at 1.18 M triangles)
    static constexpr int kMaxLeaf = 15;       // rt_bvh_node packs the count in 4 bits
    int leaf_target;                          // split until a node holds at most this many
    std::vector<Box> tri_box;                 // per triangle
    std::vector<float> centroid;              // [n][3]
    std::vector<int32_t> order;               // triangle ids, permuted in place; final leaf order
    std::vector<rt_bvh_node> nodes;

    // `subset` (optional): the triangle ids to build over; the others appear in no leaf.
    void run(int64_t n, const double *verts, int leaf_size, const std::vector<int32_t> *subset = nullptr, int want_threads = 0) {
        leaf_target = std::max(1, std::min(leaf_size, kMaxLeaf));
        tri_box.resize(n); centroid.resize(3 * (size_t)n);
        if (subset) order = *subset;
        else { order.resize(n); for (int64_t t = 0; t < n; ++t) order[t] = (int32_t)t; }
        for (int64_t t = 0; t < n; ++t) {
            const double *v = verts + 9 * (size_t)t;
            Box b; b.reset();
            for (int k = 0; k < 3; ++k) {
                double lo = std::min({v[k], v[3 + k], v[6 + k]}), hi = std::max({v[k], v[3 + k], v[6 + k]});
                if (!std::isfinite(lo) || !std::isfinite(hi)) { lo = -3.0e38; hi = 3.0e38; }   // never culled
                b.lo[k] = round_down(lo); b.hi[k] = round_up(hi);
                centroid[3 * (size_t)t + k] = (float)(0.5 * (lo + hi));
            }
            tri_box[t] = b;
        }
        nodes.clear();
        nodes.reserve((size_t)(2 * n / std::max(1, leaf_target)) + 16);
        int threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
        if (want_threads > 0) threads = std::min(want_threads, kMaxThreads);
        if (order.size() < 65536) threads = 1;
        if (order.empty()) {}
        else if (threads > 1) build_parallel(threads);
        else build(nodes, 0, (int32_t)order.size(), 0);
        const int32_t total = (int32_t)nodes.size();
        for (auto &nd : nodes) if (nd.skip >= total) nd.skip = -1;
    }

    // Decides what the node over order[begin, end) is: its box, and either "leaf" or the split point
    // `mid` (order is partitioned in place).
    static constexpr int kMaxThreads = 16;
    static constexpr int32_t kWideRange = 1 << 17;
    int par_threads = 1;   // threads the passes over a range of kWideRange or more may use (set by build_parallel for the top of the tree)

    // fn(thread, b, e) over [begin, end) in `par_threads` contiguous parts when the range is large
    template <class F> void for_parts(int32_t begin, int32_t end, int &parts, F &&fn) {
        parts = (par_threads > 1 && end - begin >= kWideRange) ? par_threads : 1;
        if (parts == 1) { fn(0, begin, end); return; }
        std::vector<std::thread> pool;
        const int64_t len = end - begin;
        for (int t = 1; t < parts; ++t)
            pool.emplace_back([&, t]() { fn(t, begin + (int32_t)(len * t / parts), begin + (int32_t)(len * (t + 1) / parts)); });
        fn(0, begin, begin + (int32_t)(len / parts));
        for (auto &th : pool) th.join();
    }

    struct Bins { Box box[3][kBins]; int32_t n[3][kBins]; };

    // Decides what the node over order[begin, end) is: its box, and either "leaf" or the split point
    // `mid` (order is partitioned in place).
    void split(int32_t begin, int32_t end, int depth, Box &bb, bool &leaf, int32_t &mid) {
        bb.reset();
        Box cb; cb.reset();
        {
            int parts;
            Box pbb[kMaxThreads], pcb[kMaxThreads];
            for_parts(begin, end, parts, [&](int t, int32_t b, int32_t e) {
                Box x, y; x.reset(); y.reset();
                for (int32_t i = b; i < e; ++i) { x.grow(tri_box[order[i]]); y.grow(&centroid[3 * (size_t)order[i]]); }
                pbb[t] = x; pcb[t] = y;
            });
            for (int t = 0; t < parts; ++t) { bb.grow(pbb[t]); cb.grow(pcb[t]); }
        }
        const int32_t count = end - begin;
        leaf = count <= leaf_target;
        mid = begin;
        if (leaf) return;
        // binned SAH over the three axes of the centroid box: one pass over the triangles fills the
        // bins of all three axes (min / max and counts: the result does not depend on how the range
        // is cut into parts)
        float c0[3], scale[3]; bool usable[3];
        for (int axis = 0; axis < 3; ++axis) {
            c0[axis] = cb.lo[axis];
            usable[axis] = cb.hi[axis] > cb.lo[axis];
            scale[axis] = usable[axis] ? (float)kBins / (cb.hi[axis] - cb.lo[axis]) : 0.0f;
        }
        Bins one;                     // the usual case: this thread alone, no heap traffic per node
        std::vector<Bins> many;
        const bool wide = par_threads > 1 && end - begin >= kWideRange;
        if (wide) many.resize(par_threads);
        Bins *pbins = wide ? many.data() : &one;
        int parts;
        for_parts(begin, end, parts, [&](int t, int32_t b, int32_t e) {
            Bins &B = pbins[t];
            for (int a = 0; a < 3; ++a) for (int k = 0; k < kBins; ++k) { B.box[a][k].reset(); B.n[a][k] = 0; }
            for (int32_t i = b; i < e; ++i) {
                const int32_t tri = order[i];
                const Box &tb = tri_box[tri];
                for (int a = 0; a < 3; ++a) {
                    if (!usable[a]) continue;
                    int k = (int)((centroid[3 * (size_t)tri + a] - c0[a]) * scale[a]);
                    k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                    B.box[a][k].grow(tb); B.n[a][k]++;
                }
            }
        });
        for (int t = 1; t < parts; ++t)
            for (int a = 0; a < 3; ++a) for (int k = 0; k < kBins; ++k) { pbins[0].box[a][k].grow(pbins[t].box[a][k]); pbins[0].n[a][k] += pbins[t].n[a][k]; }
        double best_cost = INFINITY; int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; ++axis) {
            if (!usable[axis]) continue;
            const Box *bin_box = pbins[0].box[axis]; const int32_t *bin_n = pbins[0].n[axis];
            double right_area[kBins]; int32_t right_n[kBins];
            Box acc; acc.reset(); int32_t cnt = 0;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bin_box[b]); cnt += bin_n[b]; right_area[b] = acc.half_area(); right_n[b] = cnt; }
            acc.reset(); cnt = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bin_box[b]); cnt += bin_n[b];
                if (cnt == 0 || right_n[b + 1] == 0) continue;
                const double cost = acc.half_area() * cnt + right_area[b + 1] * right_n[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
            }
        }
        // a leaf of up to kMaxLeaf triangles is kept when no split is cheaper than testing them all
        // (box test = 1, triangle test = 2 in units of the parent's area)
        if (best_axis >= 0 && count <= kMaxLeaf) {
            const double leaf_cost = 2.0 * count * bb.half_area();
            if (1.0 * bb.half_area() + 2.0 * best_cost >= leaf_cost && count <= 2 * leaf_target) { leaf = true; return; }
        }
        if (best_axis >= 0 && depth <= 48) {   // deeper than that: halve the list instead (bounded depth)
            const float c0 = cb.lo[best_axis], scale = (float)kBins / (cb.hi[best_axis] - c0);
            auto it = std::partition(order.begin() + begin, order.begin() + end, [&](int32_t t) {
                int b = (int)((centroid[3 * (size_t)t + best_axis] - c0) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= best_bin;
            });
            mid = (int32_t)(it - order.begin());
        }
        if (mid <= begin || mid >= end) {
            // all centroids coincide (or binning failed): split the list in half, which keeps the
            // depth logarithmic; if it is short enough it becomes a leaf instead
            if (count <= kMaxLeaf) { leaf = true; return; }
            int axis = 0;
            for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
            mid = begin + count / 2;
            std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end, [&](int32_t a, int32_t b) {
                return centroid[3 * (size_t)a + axis] < centroid[3 * (size_t)b + axis];
            });
        }
    }

    static void set_box(rt_bvh_node &nd, const Box &bb) { for (int k = 0; k < 3; ++k) { nd.lo[k] = bb.lo[k]; nd.hi[k] = bb.hi[k]; } }

    // Subtree over order[begin, end) appended to `out` in threaded depth-first order; links are
    // indices into `out` (a skip equal to out.size() at return means "whatever follows the subtree").
    void build(std::vector<rt_bvh_node> &out, int32_t begin, int32_t end, int depth) {
        const int32_t idx = (int32_t)out.size();
        out.push_back(rt_bvh_node{});
        Box bb; bool leaf; int32_t mid;
        split(begin, end, depth, bb, leaf, mid);
        set_box(out[idx], bb);
        if (leaf) {
            out[idx].tri_begin_count = (begin << 4) | (end - begin);
            out[idx].skip = (int32_t)out.size();
            return;
        }
        out[idx].tri_begin_count = -1;
        build(out, begin, mid, depth + 1);
        build(out, mid, end, depth + 1);
        out[idx].skip = (int32_t)out.size();
    }

    // Large inputs: the top of the tree is split on this thread down to ranges of at most `grain`
    // triangles, those subtrees are built on worker threads into vectors of their own, and the
    // pieces are laid out in the same depth-first order a single-threaded build produces (the
    // result is identical node for node).
    struct Piece {
        bool is_task; int32_t begin, end, depth;
        rt_bvh_node node;            // top node (is_task == false)
        int32_t end_piece;           // top node: index of the piece after its subtree
        std::vector<rt_bvh_node> sub;
    };
    void top(std::vector<Piece> &pieces, int32_t begin, int32_t end, int depth, int32_t grain) {
        Piece p{}; p.begin = begin; p.end = end; p.depth = depth;
        if (end - begin <= grain) { p.is_task = true; pieces.push_back(std::move(p)); return; }
        const size_t i = pieces.size();
        pieces.push_back(std::move(p));
        Box bb; bool leaf; int32_t mid;
        split(begin, end, depth, bb, leaf, mid);
        set_box(pieces[i].node, bb);
        if (leaf) { pieces[i].node.tri_begin_count = (begin << 4) | (end - begin); pieces[i].end_piece = (int32_t)pieces.size(); return; }
        pieces[i].node.tri_begin_count = -1;
        top(pieces, begin, mid, depth + 1, grain);
        top(pieces, mid, end, depth + 1, grain);
        pieces[i].end_piece = (int32_t)pieces.size();
    }
    void build_parallel(int threads) {
        const int32_t n = (int32_t)order.size();
        std::vector<Piece> pieces;
        par_threads = threads;
        top(pieces, 0, n, 0, std::max<int32_t>(4096, n / (8 * threads)));
        par_threads = 1;   // the workers below each run their subtree's passes on their own
        std::vector<Piece *> tasks;
        for (auto &p : pieces) if (p.is_task) tasks.push_back(&p);
        std::sort(tasks.begin(), tasks.end(), [](const Piece *a, const Piece *b) { return a->end - a->begin > b->end - b->begin; });
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t k = next.fetch_add(1); k < tasks.size(); k = next.fetch_add(1)) {
                Piece &p = *tasks[k];
                p.sub.reserve((size_t)(2 * (p.end - p.begin) / std::max(1, leaf_target)) + 16);
                build(p.sub, p.begin, p.end, p.depth);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        // lay the pieces out: offsets first, then nodes with their links moved
        std::vector<int32_t> offset(pieces.size() + 1, 0);
        for (size_t i = 0; i < pieces.size(); ++i) offset[i + 1] = offset[i] + (pieces[i].is_task ? (int32_t)pieces[i].sub.size() : 1);
        nodes.resize((size_t)offset[pieces.size()]);
        for (size_t i = 0; i < pieces.size(); ++i) {
            const Piece &p = pieces[i];
            if (!p.is_task) { rt_bvh_node nd = p.node; nd.skip = offset[p.end_piece]; nodes[offset[i]] = nd; continue; }
            for (size_t k = 0; k < p.sub.size(); ++k) { rt_bvh_node nd = p.sub[k]; nd.skip += offset[i]; nodes[offset[i] + k] = nd; }
        }
    }
};

// The same hierarchy threaded for one octant of ray directions (bit k of `octant` set = direction
// negative along axis k): at every inner node the child whose box centre comes first along the ray
// is visited first, so a walk meets near geometry before far geometry and its depth bound prunes
// the rest.  Subtree sizes do not depend on the order, so skip links follow from the source's.
// `src` must have passed validate() (or come from Builder).
// Refit: the boxes of an existing hierarchy (threaded depth-first order, as Builder leaves it) for
// moved vertices, topology and leaf order unchanged.  A node's subtree is [i, skip): going through
// the nodes backwards meets every child before its parent.  Boxes are rounded outward exactly as
// Builder::run rounds them, so a refitted hierarchy is as conservative as a built one.
inline void refit(std::vector<rt_bvh_node> &nodes, const std::vector<int32_t> &order, const double *verts) {
    const int32_t n = (int32_t)nodes.size();
    for (int32_t i = n - 1; i >= 0; --i) {
        rt_bvh_node &nd = nodes[i];
        Box bb; bb.reset();
        if (nd.tri_begin_count >= 0) {
            const int32_t begin = nd.tri_begin_count >> 4, cnt = nd.tri_begin_count & 15;
            for (int32_t k = begin; k < begin + cnt; ++k) {
                const double *v = verts + 9 * (size_t)order[k];
                Box b;
                for (int a = 0; a < 3; ++a) {
                    double lo = std::min({v[a], v[3 + a], v[6 + a]}), hi = std::max({v[a], v[3 + a], v[6 + a]});
                    if (!std::isfinite(lo) || !std::isfinite(hi)) { lo = -3.0e38; hi = 3.0e38; }   // never culled
                    b.lo[a] = round_down(lo); b.hi[a] = round_up(hi);
                }
                bb.grow(b);
            }
        } else {
            const int32_t end = nd.skip < 0 ? n : nd.skip;
            for (int32_t c = i + 1; c < end; c = nodes[c].skip < 0 ? n : nodes[c].skip) {
                Box b;
                for (int a = 0; a < 3; ++a) { b.lo[a] = nodes[c].lo[a]; b.hi[a] = nodes[c].hi[a]; }
                bb.grow(b);
            }
        }
        Builder::set_box(nd, bb);
    }
}

// Sum of the half-areas of all boxes: the builder's cost measure, used to tell when a refitted topology has gone stale.
inline double area_sum(const std::vector<rt_bvh_node> &nodes) {
    double sum = 0;
    for (const rt_bvh_node &nd : nodes) {
        const double dx = (double)nd.hi[0] - nd.lo[0], dy = (double)nd.hi[1] - nd.lo[1], dz = (double)nd.hi[2] - nd.lo[2];
        if (dx >= 0 && dy >= 0 && dz >= 0 && std::isfinite(dx + dy + dz)) sum += dx * dy + dy * dz + dz * dx;
    }
    return sum;
}

inline void rethread(const std::vector<rt_bvh_node> &src, int octant, std::vector<rt_bvh_node> &dst) {
    const int32_t n = (int32_t)src.size();
    dst.clear(); dst.reserve(n);
    auto end_of = [&](int32_t i) { return src[i].skip < 0 ? n : src[i].skip; };
    std::vector<int32_t> stack;
    if (n > 0) stack.push_back(0);
    while (!stack.empty()) {
        const int32_t i = stack.back(); stack.pop_back();
        const int32_t idx = (int32_t)dst.size();
        rt_bvh_node nd = src[i];
        nd.skip = idx + (end_of(i) - i);
        dst.push_back(nd);
        if (src[i].tri_begin_count >= 0) continue;
        const int32_t l = i + 1, r = end_of(l);
        int axis = 0; float best = -1.0f, diff[3];
        for (int k = 0; k < 3; ++k) {
            diff[k] = (src[r].lo[k] + src[r].hi[k]) - (src[l].lo[k] + src[l].hi[k]);   // > 0: r lies further along +k
            if (std::fabs(diff[k]) > best) { best = std::fabs(diff[k]); axis = k; }
        }
        const bool negative = (octant >> axis) & 1;
        const bool l_first = negative ? !(diff[axis] > 0) : !(diff[axis] < 0);
        const int32_t first = l_first ? l : r, second = l_first ? r : l;
        stack.push_back(second);
        stack.push_back(first);
    }
    for (auto &nd : dst) if (nd.skip >= n) nd.skip = -1;
}

// Device order of a valid threaded hierarchy: the nodes of the top levels first (breadth-first:
// these are the ones every ray visits, and the kernels stage them in LDS), then everything below in
// the original depth-first order.  A top node can no longer find its first child at index + 1, so an
// inner node's tri_begin_count carries it explicitly as -(child + 2); -1 still means "the next node"
// (true for every node below the top, whose children moved with it).  Leaves are unchanged.
// Returns the number of top nodes (<= max_top; whole levels only).
inline int32_t reorder_top_first(const std::vector<rt_bvh_node> &src, int32_t max_top, std::vector<rt_bvh_node> &dst) {
    const int32_t n = (int32_t)src.size();
    dst.clear();
    if (n == 0) return 0;
    std::vector<int32_t> depth(n, 0);
    {   // depth of every node from the nesting of the skip links
        std::vector<int32_t> open_end;
        for (int32_t i = 0; i < n; ++i) {
            while (!open_end.empty() && open_end.back() <= i) open_end.pop_back();
            depth[i] = (int32_t)open_end.size();
            if (src[i].tri_begin_count < 0) open_end.push_back(src[i].skip < 0 ? n : src[i].skip);
        }
    }
    int32_t max_depth = 0;
    for (int32_t d : depth) max_depth = std::max(max_depth, d);
    std::vector<int32_t> per_level(max_depth + 2, 0);
    for (int32_t d : depth) per_level[d]++;
    int32_t levels = 0, n_top = 0;                    // whole levels while they fit
    while (levels <= max_depth && n_top + per_level[levels] <= max_top) { n_top += per_level[levels]; ++levels; }
    std::vector<int32_t> where(n, -1);
    int32_t next = 0;
    for (int32_t lv = 0; lv < levels; ++lv)
        for (int32_t i = 0; i < n; ++i) if (depth[i] == lv) where[i] = next++;
    for (int32_t i = 0; i < n; ++i) if (where[i] < 0) where[i] = next++;
    dst.resize(n);
    for (int32_t i = 0; i < n; ++i) {
        rt_bvh_node nd = src[i];
        nd.skip = src[i].skip < 0 ? -1 : where[src[i].skip];
        if (src[i].tri_begin_count < 0) {
            const int32_t child = where[i + 1];
            nd.tri_begin_count = (child == where[i] + 1) ? -1 : -(child + 2);
        }
        dst[where[i]] = nd;
    }
    return n_top;
}

// The wide form of a valid threaded hierarchy (see WideNode): every node takes its two children and then, while it
// has fewer than four, replaces the inner child with the largest box by that child's two children.  A node's inner
// children get consecutive indices (in slot order), and the nodes are laid out depth first by such blocks, so a walk
// down the first children reads ascending addresses.  Leaves keep their triangle ranges.  For each octant of ray
// directions the slots are ordered by the projection of the box centres on that octant's diagonal.
// `stack_need` receives the largest number of stack entries any walk can ask for (a visit pushes every inner child
// that is hit and pops one: the sum of (inner children - 1) along a root-to-node path, plus that node's own).
// Returns false when the hierarchy has more nodes than the walk's 24-bit node numbers can name.
inline bool collapse_wide(const std::vector<rt_bvh_node> &src, std::vector<WideNode> &out, int32_t *stack_need) {
    const int32_t n = (int32_t)src.size();
    out.clear();
    if (stack_need) *stack_need = 0;
    if (n == 0) return true;
    auto end_of = [&](int32_t i) { return src[i].skip < 0 ? n : src[i].skip; };
    auto is_leaf = [&](int32_t i) { return src[i].tri_begin_count >= 0; };
    auto area = [&](int32_t i) {
        const double dx = (double)src[i].hi[0] - src[i].lo[0], dy = (double)src[i].hi[1] - src[i].lo[1], dz = (double)src[i].hi[2] - src[i].lo[2];
        return (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.0;
    };
    auto up32 = [](double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, INFINITY) : f; };
    struct Work { int32_t src, wide, pending; };   // pending: stack entries in use when the walk reaches this node
    std::vector<Work> todo;
    out.push_back(WideNode{});
    todo.push_back({0, 0, 0});
    int32_t need = 0;
    while (!todo.empty()) {
        const Work w = todo.back(); todo.pop_back();
        int32_t kids[4]; int nk = 0;
        if (is_leaf(w.src)) kids[nk++] = w.src;     // (a hierarchy that is one leaf: the root holds it)
        else {
            kids[nk++] = w.src + 1; kids[nk++] = end_of(w.src + 1);
            while (nk < 4) {
                int best = -1; double best_area = -1.0;
                for (int k = 0; k < nk; ++k) if (!is_leaf(kids[k]) && area(kids[k]) > best_area) { best_area = area(kids[k]); best = k; }
                if (best < 0) break;
                const int32_t c = kids[best], l = c + 1, r = end_of(l);
                for (int k = nk; k > best + 1; --k) kids[k] = kids[k - 1];
                kids[best] = l; kids[best + 1] = r; ++nk;
            }
        }
        WideNode nd{};
        double centre[4][3] = {{0}};
        uint32_t inner = 0, used = 0;
        for (int k = 0; k < 4; ++k) {
            float c[3] = {0, 0, 0}, h[3] = {-1e30f, -1e30f, -1e30f};
            if (k < nk) {
                const rt_bvh_node &b = src[kids[k]];
                for (int a = 0; a < 3; ++a) {
                    const double mid = 0.5 * ((double)b.lo[a] + (double)b.hi[a]);
                    c[a] = std::isfinite(mid) ? (float)mid : 0.0f;
                    // the half-extent covers the rounding of the centre: [c - h, c + h] contains [lo, hi] (empty box: stays empty)
                    const double hh = std::max((double)b.hi[a] - (double)c[a], (double)c[a] - (double)b.lo[a]);
                    h[a] = hh >= 0 ? std::min(up32(hh * (1.0 + 1e-7)), 3.0e38f) : -1e30f;
                    centre[k][a] = c[a];
                }
                used |= 1u << k;
                if (is_leaf(kids[k])) nd.link[k] = b.tri_begin_count; else { inner |= 1u << k; nd.link[k] = -1; }
            } else nd.link[k] = 0;   // (an empty slot is never hit; a zero-count leaf if it ever were)
            nd.cx[k] = c[0]; nd.hx[k] = h[0]; nd.cy[k] = c[1]; nd.hy[k] = h[1]; nd.cz[k] = c[2]; nd.hz[k] = h[2];
        }
        for (int oct = 0; oct < 8; ++oct) {
            int idx[4] = {0, 1, 2, 3};
            double key[4];
            for (int k = 0; k < 4; ++k) {
                key[k] = k < nk ? 0.0 : INFINITY;   // empty slots last
                for (int a = 0; a < 3 && k < nk; ++a) key[k] += ((oct >> a) & 1) ? -centre[k][a] : centre[k][a];
            }
            std::stable_sort(idx, idx + 4, [&](int x, int y) { return key[x] < key[y]; });
            uint32_t byte = 0;
            for (int j = 0; j < 4; ++j) byte |= (uint32_t)idx[j] << (2 * j);
            nd.order[oct >> 2] |= byte << (8 * (oct & 3));
        }
        const int n_inner = __builtin_popcount(inner);
        nd.first = (int32_t)out.size();
        nd.inner = inner | (used << 4);
        need = std::max(need, w.pending + n_inner);
        for (int k = 0; k < n_inner; ++k) out.push_back(WideNode{});
        out[w.wide] = nd;
        // children are worked on in slot order (the work list is a stack: push them in reverse)
        int rank = n_inner;
        for (int k = 3; k >= 0; --k)
            if (inner & (1u << k)) { --rank; todo.push_back({kids[k], nd.first + rank, w.pending + n_inner - 1}); }
        if (out.size() >= (size_t)(1 << 24)) return false;
    }
    if (stack_need) *stack_need = need;
    return true;
}

// Structural and containment checks.  Returns an empty string when the hierarchy is safe to use:
// links only point forward and nest, leaves index [0, n_tris) and cover every triangle exactly once,
// every leaf box contains its triangles and every box lies inside its parent's.
inline std::string validate(int32_t n_nodes, const rt_bvh_node *nodes, int32_t n_tris, const int32_t *tri_index, const double *verts) {
    if (n_nodes <= 0 || !nodes || !tri_index) return "empty hierarchy";
    std::vector<uint8_t> seen((size_t)n_tris, 0);
    struct Open { int32_t end; const rt_bvh_node *box; };
    std::vector<Open> stack;
    char buf[160];
    for (int32_t i = 0; i < n_nodes; ++i) {
        const rt_bvh_node &nd = nodes[i];
        const int32_t end = nd.skip < 0 ? n_nodes : nd.skip;
        if (end <= i || end > n_nodes) { snprintf(buf, sizeof buf, "node %d: skip %d does not point forward", i, nd.skip); return buf; }
        while (!stack.empty() && stack.back().end <= i) stack.pop_back();
        if (!stack.empty()) {
            if (end > stack.back().end) { snprintf(buf, sizeof buf, "node %d: subtree crosses its parent's end", i); return buf; }
            for (int k = 0; k < 3; ++k)
                if (!(nd.lo[k] >= stack.back().box->lo[k] && nd.hi[k] <= stack.back().box->hi[k])) {
                    snprintf(buf, sizeof buf, "node %d: box not inside its parent's", i); return buf;
                }
        } else if (i != 0) { snprintf(buf, sizeof buf, "node %d is not reachable from the root", i); return buf; }
        if (nd.tri_begin_count < 0) {
            if (end == i + 1) { snprintf(buf, sizeof buf, "inner node %d has no children", i); return buf; }
            stack.push_back({end, &nd});
        } else {
            if (end != i + 1) { snprintf(buf, sizeof buf, "leaf %d: skip must be the next node", i); return buf; }
            const int32_t first = nd.tri_begin_count >> 4, cnt = nd.tri_begin_count & 15;
            if (cnt < 1 || first < 0 || (int64_t)first + cnt > n_tris) { snprintf(buf, sizeof buf, "leaf %d: bad triangle range", i); return buf; }
            for (int32_t k = first; k < first + cnt; ++k) {
                const int32_t t = tri_index[k];
                if (t < 0 || t >= n_tris || seen[t]) { snprintf(buf, sizeof buf, "leaf %d: triangle %d out of range or listed twice", i, t); return buf; }
                seen[t] = 1;
                const double *v = verts + 9 * (size_t)t;
                for (int c = 0; c < 9; ++c) {
                    if (!std::isfinite(v[c])) continue;   // such a triangle can never be hit
                    if (!(v[c] >= (double)nd.lo[c % 3] && v[c] <= (double)nd.hi[c % 3])) {
                        snprintf(buf, sizeof buf, "leaf %d: triangle %d sticks out of the box", i, t); return buf;
                    }
                }
            }
        }
    }
    for (int32_t t = 0; t < n_tris; ++t) if (!seen[t]) { snprintf(buf, sizeof buf, "triangle %d is in no leaf", t); return buf; }
    return std::string();
}

}  // namespace bvh
}  // namespace rt
