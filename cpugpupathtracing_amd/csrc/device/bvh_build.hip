// bvh_build.hip -- GPU BVH::Build / BVH::Rebuild for the reference's three BuildOptions, bit-identical to the host build.
//
// Options (ref: Source/BVH.cpp:204-297): SAH split intervals (the default, BVH.h:43-44; described below), naive split (midpoint of the
// longest axis, stop at <= 2 triangles, :208-224 -- the same level kernels with the 24-candidate sweep replaced by one plane), and
// SAH split primitives, whose cheapest_cost is never updated (SURVEY A-5), so the root stays a leaf: no kernel at all.
// Rebuild (ref: BVH.cpp:47-59) re-splits over the CURRENT triangle order (m_tri_indices is not reset), so it is the same build
// started from a caller-provided permutation instead of the identity.
//
// Reproduces ref: Source/BVH.cpp:11-45 (Build), :204-259 (Subdivide, SAH with 8 planes x 3 axes on the node bounds),
// :299-327 (EvaluateSAH), :329-366 (Split) -- same split decisions, same node numbering, same triangle order:
//   * levels run one after the other, all nodes of a level are independent.  Deep levels (many small nodes): one workgroup per
//     node does everything (subdivide_level).  The top levels (few, huge nodes -- the root alone is the whole mesh) are cut into
//     kWideBlocks-ish position-ordered pieces, one workgroup each, and run as six short kernels (wide_*) whose per-piece partial
//     results are combined IN PIECE ORDER by one workgroup per node: with one workgroup per node the first ten levels of a
//     1.31 M-triangle mesh kept one CU busy for 0.5 s;
//   * a candidate plane's cost is the reference's float expression on exact counts and exact (min/max) bounds.  Bounds are
//     reduced in triangle order (each thread owns a contiguous chunk, partial results are combined in thread order), so
//     even the sign of a zero bound is the one the sequential std::min/std::max chain leaves;
//   * the in-place swap partition (BVH.cpp:334-344) is order-sensitive; its result has a closed form (see partition_slot
//     below, checked against the sequential loop in tests/test_host.py) that is evaluated in parallel from one prefix
//     sum of the "left" flags;
//   * nodes are created in breadth-first order with an atomic counter and renumbered at the end into the reference's
//     allocation order (children of the k-th splitting node in depth-first preorder get indices 2k+1, 2k+2).
#include <hip/hip_runtime.h>

#include <exception>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cpugpupt_abi.h"

namespace cgpt {

hipStream_t CtxStream(cgpt_ctx* ctx);
int CtxDevice(cgpt_ctx* ctx);
int CtxFail(cgpt_ctx* ctx, int code, const char* fmt, ...);
cgpt_ctx* GroupFirstMemberOrNull(cgpt_ctx* ctx);
int GroupForwarded(cgpt_ctx* ctx, int rc);

namespace {

constexpr uint32_t kBuildThreads = 256;

struct F3 { float x, y, z; };
__device__ __host__ inline float min_std(float a, float b) { return (b < a) ? b : a; }   // std::min(a,b), ref: MathLib.h:95
__device__ __host__ inline float max_std(float a, float b) { return (a < b) ? b : a; }   // std::max(a,b), ref: MathLib.h:96
__device__ inline F3 f3min(F3 a, F3 b) { return { min_std(a.x, b.x), min_std(a.y, b.y), min_std(a.z, b.z) }; }
__device__ inline F3 f3max(F3 a, F3 b) { return { max_std(a.x, b.x), max_std(a.y, b.y), max_std(a.z, b.z) }; }
__device__ inline float axis_of(F3 v, uint32_t a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
__device__ inline float half_area(F3 lo, F3 hi)                               // GetAABBVolume, ref: Primitives.cpp:280-284
{
    const float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
    return ex * ey + ey * ez + ez * ex;
}

struct BuildNode {              // breadth-first working node
    F3 lo, hi;
    uint32_t first, count;      // segment of tri_indices
    uint32_t left;              // breadth-first id of the left child (right = left + 1); 0 = leaf
    uint32_t depth;
    uint32_t splits;            // splitting nodes in this subtree (renumbering)
    uint32_t rank;              // preorder rank among splitting nodes (renumbering)
    uint32_t new_id;            // index in the reference's numbering
};

struct Bounds { F3 lo, hi; };
struct WidePartial { Bounds lb, rb; uint32_t lc, rc; };
struct WideDecision { uint32_t split, axis; float pos; uint32_t n_left; };
struct WideChild { Bounds lb, rb; };

struct BuildArrays {
    const F3* tri_lo; const F3* tri_hi; const F3* centroid;   // per triangle
    uint32_t* tri_indices; uint32_t* scratch_idx;              // n_tris each
    uint32_t* sel_a; uint32_t* sel_b;                           // n_tris each: hole / tail-left position tables of the partition
    BuildNode* nodes;                                            // 2 n_tris - 1
    uint32_t* counters;                                          // [0] nodes allocated, [1] max depth
    uint32_t naive;                                              // BuildOption_NaiveSplit: one plane per node instead of the SAH sweep
    // wide levels only (index j = node of the level, b = piece of the node)
    WidePartial* partial;                                        // [j][b][24]: candidate sums of one piece
    WideDecision* decision;                                      // [j]
    uint32_t* piece_left;                                        // [j][b]: lefts before the piece (exclusive prefix over the pieces)
    uint32_t* piece_front;                                       // [j][b]: lefts of the piece that lie in the front [0, n_left)
    WideChild* piece_child;                                      // [j][b]: child bounds of the piece, new triangle order
};

// ---- per-triangle preparation: bounds and centroid (ref: Primitives.cpp:232-243, 255-258) ------------------------------------
__global__ void prepare_triangles(const cgpt_triangle* tris, uint32_t n, F3* lo, F3* hi, F3* centroid, uint32_t* idx, uint32_t identity)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cgpt_triangle& t = tris[i];
    const F3 p0 = { t.v0.pos[0], t.v0.pos[1], t.v0.pos[2] }, p1 = { t.v1.pos[0], t.v1.pos[1], t.v1.pos[2] }, p2 = { t.v2.pos[0], t.v2.pos[1], t.v2.pos[2] };
    lo[i] = f3min(f3min(p0, p1), p2);
    hi[i] = f3max(f3max(p0, p1), p2);
    centroid[i] = { ((p0.x + p1.x) + p2.x) * 0.3333f, ((p0.y + p1.y) + p2.y) * 0.3333f, ((p0.z + p1.z) + p2.z) * 0.3333f };
    if (identity) idx[i] = i;                                              // Build: ref BVH.cpp:25-29; Rebuild keeps the order it is given
}

// ---- ordered block reductions ------------------------------------------------------------------------------------------------
// Threads own contiguous chunks in position order, so "the earlier operand" is always the lower lane / lower wave:
// combining with a = earlier, b = later keeps std::min / std::max's left-most-of-equals result.
__device__ inline Bounds empty_bounds() { return { { 1e30f, 1e30f, 1e30f }, { -1e30f, -1e30f, -1e30f } }; }
__device__ inline Bounds merge(Bounds a, Bounds b) { return { f3min(a.lo, b.lo), f3max(a.hi, b.hi) }; }

__device__ inline float shfl_down_f(float v, int off) { return __shfl_down(v, off, 64); }
__device__ inline Bounds shfl_down_bounds(Bounds v, int off)
{
    return { { shfl_down_f(v.lo.x, off), shfl_down_f(v.lo.y, off), shfl_down_f(v.lo.z, off) },
             { shfl_down_f(v.hi.x, off), shfl_down_f(v.hi.y, off), shfl_down_f(v.hi.z, off) } };
}
// result valid in thread 0
__device__ inline Bounds block_reduce_bounds(Bounds v, Bounds* lds /* [4] */)
{
    for (int off = 1; off < 64; off <<= 1) {          // lane l absorbs lane l + off: earlier operand first
        const Bounds o = shfl_down_bounds(v, off);
        if ((threadIdx.x & 63) + off < 64) v = merge(v, o);
    }
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) v = merge(merge(lds[0], lds[1]), merge(lds[2], lds[3]));
    __syncthreads();
    return v;
}
__device__ inline uint32_t block_reduce_sum(uint32_t v, uint32_t* lds /* [4] */)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) v = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return v;
}
// exclusive prefix over the block of one value per thread; *total gets the block sum (all threads)
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds /* [5] */, uint32_t* total)
{
    uint32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    if ((threadIdx.x & 63) == 63) lds[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += lds[w];
    *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + incl - v;
}

// ---- group-wide primitives: a group is a whole 256-thread block (G = 256), one wavefront (G = 64) or a quarter of one (G = 16);
//      the sub-block groups use neither LDS nor barriers ------------------------------------------------------------------------
__device__ inline Bounds shfl_down_bounds_w(Bounds v, int off, int width)
{
    return { { __shfl_down(v.lo.x, off, width), __shfl_down(v.lo.y, off, width), __shfl_down(v.lo.z, off, width) },
             { __shfl_down(v.hi.x, off, width), __shfl_down(v.hi.y, off, width), __shfl_down(v.hi.z, off, width) } };
}
template <uint32_t G> __device__ inline void group_sync()                       // orders the group's global-memory writes before its reads
{
    if (G == kBuildThreads) __syncthreads();
    else { __threadfence_block(); __builtin_amdgcn_wave_barrier(); }
}
template <uint32_t G> __device__ inline Bounds group_reduce_bounds(Bounds v, Bounds* lds)       // result valid in thread 0 of the group
{
    if (G == kBuildThreads) return block_reduce_bounds(v, lds);
    for (int off = 1; off < (int)G; off <<= 1) {      // lane l absorbs lane l + off: earlier operand first
        const Bounds o = shfl_down_bounds_w(v, off, (int)G);
        if ((threadIdx.x % G) + off < G) v = merge(v, o);
    }
    return v;
}
template <uint32_t G> __device__ inline uint32_t group_reduce_sum(uint32_t v, uint32_t* lds)    // result valid in thread 0 of the group
{
    if (G == kBuildThreads) return block_reduce_sum(v, lds);
    for (int off = 1; off < (int)G; off <<= 1) {
        const uint32_t o = __shfl_down(v, off, (int)G);
        if ((threadIdx.x % G) + off < G) v += o;
    }
    return v;
}
template <uint32_t G> __device__ inline uint32_t group_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t* total)
{
    if (G == kBuildThreads) return block_exclusive_scan(v, lds, total);
    uint32_t incl = v;
    for (int off = 1; off < (int)G; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, (int)G);
        if ((int)(threadIdx.x % G) >= off) incl += o;
    }
    *total = __shfl(incl, (int)G - 1, (int)G);
    return incl - v;
}
template <uint32_t G> __device__ inline uint32_t group_broadcast(uint32_t v, uint32_t* lds_word)     // thread 0's value to the group
{
    if (G == kBuildThreads) {
        if (threadIdx.x == 0) *lds_word = v;
        __syncthreads();
        const uint32_t r = *lds_word;
        __syncthreads();
        return r;
    }
    return __shfl(v, 0, (int)G);
}

// ---- one tree level, one GROUP per node (ref: BVH.cpp:225-259 + Split :329-366) --------------------------------------------------
// G = 256: a workgroup per node.  G = 64 / 16: a wavefront / a quarter wavefront per node -- the deep levels, where a level is hundreds
// of thousands of nodes of a few triangles each and a workgroup's two hundred barriers per node were the whole cost (17 ms per level).
// Groups of one wavefront diverge at the leaf / no-split exits; a group's lanes always leave together, and no lane reads another
// group's registers.
// NaiveSplit's plane: the midpoint of the longest axis of the node bounds (ref: BVH.cpp:214-223)
__device__ inline void naive_plane(const BuildNode& node, uint32_t& axis, float& pos)
{
    const F3 extent = { node.hi.x - node.lo.x, node.hi.y - node.lo.y, node.hi.z - node.lo.z };
    axis = 0u;
    if (extent.y > extent.x) axis = 1u;
    if (extent.z > axis_of(extent, axis)) axis = 2u;
    pos = axis_of(node.lo, axis) + axis_of(extent, axis) * 0.5f;
}

template <uint32_t G>
__global__ void __launch_bounds__(kBuildThreads) subdivide_level(BuildArrays A, uint32_t level_first, uint32_t level_count)
{
    __shared__ Bounds s_bounds[4];
    __shared__ uint32_t s_u32[5];
    __shared__ uint32_t s_word;

    constexpr uint32_t groups = kBuildThreads / G;
    const uint32_t in_level = blockIdx.x * groups + threadIdx.x / G, t = threadIdx.x % G;
    if (in_level >= level_count) return;                                       // group-uniform
    const uint32_t node_id = level_first + in_level;
    BuildNode node = A.nodes[node_id];
    const uint32_t n = node.count, first = node.first;
    uint32_t* const idx = A.tri_indices + first;

    // contiguous chunk of this thread, in position order
    const uint32_t chunk = (n + G - 1) / G;
    const uint32_t c0 = min(t * chunk, n), c1 = min(c0 + chunk, n);

    // ---- SAH over 8 planes x 3 axes (split_idx outer, axis inner: the first strictly cheaper candidate wins) ----
    float cheapest_cost = 1e30f; uint32_t cheapest_axis = 0; float cheapest_pos = 0.0f;      // meaningful in thread 0
    float parent_cost = half_area(node.lo, node.hi) * (float)n;
    if (A.naive) {                                                              // ref: BVH.cpp:208-224
        naive_plane(node, cheapest_axis, cheapest_pos);
        cheapest_cost = 0.0f; parent_cost = n <= 2u ? 0.0f : 1.0f;              // "split unless <= 2 triangles" in the test below
    }
    for (uint32_t split_idx = 0; split_idx < (A.naive ? 0u : 8u); ++split_idx) {
        for (uint32_t axis = 0; axis < 3; ++axis) {
            const float axis_width = axis_of(node.hi, axis) - axis_of(node.lo, axis);
            const float split_pos = axis_width * ((float)split_idx / 8) + axis_of(node.lo, axis);
            Bounds lb = empty_bounds(), rb = empty_bounds();
            uint32_t lc = 0, rc = 0;
            for (uint32_t i = c0; i < c1; ++i) {
                const uint32_t tri = idx[i];
                const Bounds tb = { A.tri_lo[tri], A.tri_hi[tri] };
                if (axis_of(A.centroid[tri], axis) < split_pos) { ++lc; lb = merge(lb, tb); }
                else { ++rc; rb = merge(rb, tb); }
            }
            lb = group_reduce_bounds<G>(lb, s_bounds);
            rb = group_reduce_bounds<G>(rb, s_bounds);
            lc = group_reduce_sum<G>(lc, s_u32);
            rc = group_reduce_sum<G>(rc, s_u32);
            if (t == 0) {
                // an empty side has extent -2e30 -> +inf area -> 0 * inf = NaN, which the "<" rejects (ref: BVH.cpp:325)
                const float split_cost = (float)lc * half_area(lb.lo, lb.hi) + (float)rc * half_area(rb.lo, rb.hi);
                if (split_cost < cheapest_cost) { cheapest_cost = split_cost; cheapest_axis = axis; cheapest_pos = split_pos; }
            }
        }
    }
    if (t == 0) atomicMax(&A.counters[1], node.depth);                                          // m_max_depth, ref: BVH.cpp:206
    const uint32_t do_split = group_broadcast<G>((cheapest_cost >= parent_cost) ? 0u : 1u, &s_word);   // ref: BVH.cpp:253-256
    if (do_split == 0u) return;                                                                 // leaf
    const uint32_t axis = group_broadcast<G>(cheapest_axis, &s_word);
    const float split_pos = __uint_as_float(group_broadcast<G>(__float_as_uint(cheapest_pos), &s_word));

    // ---- Split: the reference's in-place swap partition (BVH.cpp:331-344), evaluated in closed form ----
    // L(p) = lefts before p.  n_left = L(n).  Front [0, n_left): a left element stays; the k-th hole (right element,
    // k = p - L(p)) receives the k-th tail-left counted from the end.  Tail [n_left, n): position q receives a[q+1] when a[q+1]
    // is a right element, otherwise (q = n-1 or a[q+1] is a left element) the k-th hole's element with k = lefts behind q, and
    // once the holes are used up the first tail element a[n_left] (the final rotation).
    uint32_t my_left = 0;
    for (uint32_t i = c0; i < c1; ++i) my_left += (axis_of(A.centroid[idx[i]], axis) < split_pos) ? 1u : 0u;
    uint32_t n_left = 0;
    const uint32_t left_before_chunk = group_exclusive_scan<G>(my_left, s_u32, &n_left);
    uint32_t* const hole_pos = A.sel_a + first;          // k-th hole -> position
    uint32_t* const tail_left_pos = A.sel_b + first;     // k-th tail-left from the end -> position
    uint32_t* const out = A.scratch_idx + first;
    uint32_t front_left = 0;
    {
        uint32_t L = left_before_chunk;
        for (uint32_t p = c0; p < c1; ++p) {
            const bool is_left = axis_of(A.centroid[idx[p]], axis) < split_pos;
            if (p < n_left) { if (is_left) ++front_left; else hole_pos[p - L] = p; }
            else if (is_left) tail_left_pos[n_left - L - 1u] = p;       // lefts at positions > p: n_left - (L + 1)
            L += is_left ? 1u : 0u;
        }
    }
    front_left = group_reduce_sum<G>(front_left, s_u32);
    group_sync<G>();                                                    // publishes the two tables
    const uint32_t holes = n_left - group_broadcast<G>(front_left, &s_word);    // rights in the front = lefts in the tail
    {
        uint32_t L = left_before_chunk;
        for (uint32_t p = c0; p < c1; ++p) {
            const bool is_left = axis_of(A.centroid[idx[p]], axis) < split_pos;
            L += is_left ? 1u : 0u;                                     // now: lefts at positions <= p
            uint32_t value;
            if (p < n_left) {
                value = is_left ? idx[p] : idx[tail_left_pos[p - L]];                  // hole number = rights before p
            } else if (p + 1u == n || axis_of(A.centroid[idx[p + 1u]], axis) < split_pos) {
                const uint32_t k = n_left - L;                                              // lefts at positions > p
                value = k < holes ? idx[hole_pos[k]] : idx[n_left];
            } else {
                value = idx[p + 1u];
            }
            out[p] = value;
        }
    }
    group_sync<G>();                                                    // every read of the old order is done
    for (uint32_t p = c0; p < c1; ++p) idx[p] = out[p];
    group_sync<G>();

    if (n_left == 0u || n_left == n) return;                                                // ref: BVH.cpp:346-348 (stays a leaf)

    // ---- children (ref: BVH.cpp:350-362): bounds in the NEW triangle order (CalculateNodeBounds) ----
    Bounds lb = empty_bounds(), rb = empty_bounds();
    for (uint32_t p = c0; p < c1; ++p) {
        const uint32_t tri = idx[p];
        const Bounds tb = { A.tri_lo[tri], A.tri_hi[tri] };
        if (p < n_left) lb = merge(lb, tb); else rb = merge(rb, tb);
    }
    lb = group_reduce_bounds<G>(lb, s_bounds);
    rb = group_reduce_bounds<G>(rb, s_bounds);
    if (t == 0) {
        const uint32_t left_id = atomicAdd(&A.counters[0], 2u);
        BuildNode l{}, r{};
        l.lo = lb.lo; l.hi = lb.hi; l.first = first; l.count = n_left; l.depth = node.depth + 1u;
        r.lo = rb.lo; r.hi = rb.hi; r.first = first + n_left; r.count = n - n_left; r.depth = node.depth + 1u;
        A.nodes[left_id] = l; A.nodes[left_id + 1u] = r;
        A.nodes[node_id].left = left_id;
    }
}

// ---- the top levels: a node's index range cut into `pieces` position-ordered pieces, one workgroup each --------------------------
// Same arithmetic as subdivide_level, split where that kernel has a block-wide dependency: candidate sums per piece ->
// (wide_decide) combine in piece order, choose the plane, prefix of the left counts -> partition tables per piece -> the closed-form
// partition per piece -> copy back + child bounds per piece -> (wide_children) combine in piece order, create the children.
// min_std / max_std keep the EARLIER operand on ties and every combination below has the earlier piece on the left, so the result is
// the one the sequential chain over the node's triangles leaves (same argument as for the thread-ordered block reduction).
constexpr uint32_t kWideBlocks = 1024;      // pieces of a level at most (nodes x pieces per node)
constexpr uint32_t kCandidates = 24;        // 8 planes x 3 axes, split_idx outer (ref: BVH.cpp:229-251)

struct Piece { uint32_t j, b, n, first, c0, c1; };       // node of the level, piece; node size / first index; this thread's positions
__device__ inline Piece piece_of(const BuildArrays& A, uint32_t level_first, uint32_t pieces, BuildNode& node)
{
    Piece q;
    q.j = blockIdx.x / pieces; q.b = blockIdx.x - q.j * pieces;
    node = A.nodes[level_first + q.j];
    q.n = node.count; q.first = node.first;
    const uint32_t per_piece = (q.n + pieces - 1u) / pieces;
    const uint32_t p0 = min(q.b * per_piece, q.n), p1 = min(p0 + per_piece, q.n);
    const uint32_t chunk = (p1 - p0 + kBuildThreads - 1u) / kBuildThreads;
    q.c0 = min(p0 + threadIdx.x * chunk, p1); q.c1 = min(q.c0 + chunk, p1);
    return q;
}
__device__ inline float candidate_pos(const BuildNode& node, uint32_t split_idx, uint32_t axis)      // ref: BVH.cpp:233-234
{
    const float axis_width = axis_of(node.hi, axis) - axis_of(node.lo, axis);
    return axis_width * ((float)split_idx / 8) + axis_of(node.lo, axis);
}

__global__ void __launch_bounds__(kBuildThreads) wide_sah_partial(BuildArrays A, uint32_t level_first, uint32_t pieces)
{
    __shared__ Bounds s_bounds[4];
    __shared__ uint32_t s_u32[5];
    BuildNode node;
    const Piece q = piece_of(A, level_first, pieces, node);
    const uint32_t* const idx = A.tri_indices + q.first;
    WidePartial* const out = A.partial + ((size_t)q.j * pieces + q.b) * kCandidates;
    uint32_t naive_axis = 0; float naive_pos = 0.0f;
    if (A.naive) naive_plane(node, naive_axis, naive_pos);                      // one candidate, kept in slot 0
    for (uint32_t split_idx = 0; split_idx < (A.naive ? 1u : 8u); ++split_idx) {
        for (uint32_t axis = (A.naive ? naive_axis : 0u); axis < (A.naive ? naive_axis + 1u : 3u); ++axis) {
            const float split_pos = A.naive ? naive_pos : candidate_pos(node, split_idx, axis);
            Bounds lb = empty_bounds(), rb = empty_bounds();
            uint32_t lc = 0, rc = 0;
            for (uint32_t i = q.c0; i < q.c1; ++i) {
                const uint32_t tri = idx[i];
                const Bounds tb = { A.tri_lo[tri], A.tri_hi[tri] };
                if (axis_of(A.centroid[tri], axis) < split_pos) { ++lc; lb = merge(lb, tb); }
                else { ++rc; rb = merge(rb, tb); }
            }
            lb = block_reduce_bounds(lb, s_bounds);
            rb = block_reduce_bounds(rb, s_bounds);
            lc = block_reduce_sum(lc, s_u32);
            rc = block_reduce_sum(rc, s_u32);
            if (threadIdx.x == 0) out[A.naive ? 0u : split_idx * 3u + axis] = { lb, rb, lc, rc };
        }
    }
}

__global__ void __launch_bounds__(64) wide_decide(BuildArrays A, uint32_t level_first, uint32_t pieces)
{
    __shared__ WidePartial s_c[kCandidates];
    const uint32_t j = blockIdx.x;
    const BuildNode node = A.nodes[level_first + j];
    if (threadIdx.x < (A.naive ? 1u : kCandidates)) {
        WidePartial acc = { empty_bounds(), empty_bounds(), 0u, 0u };
        for (uint32_t b = 0; b < pieces; ++b) {                                 // piece order = position order
            const WidePartial w = A.partial[((size_t)j * pieces + b) * kCandidates + threadIdx.x];
            acc.lb = merge(acc.lb, w.lb); acc.rb = merge(acc.rb, w.rb); acc.lc += w.lc; acc.rc += w.rc;
        }
        s_c[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float cheapest_cost = 1e30f; uint32_t cheapest = 0;
    for (uint32_t c = 0; c < (A.naive ? 0u : kCandidates); ++c) {               // split_idx outer, axis inner: the first strictly cheaper wins
        const WidePartial& w = s_c[c];
        const float split_cost = (float)w.lc * half_area(w.lb.lo, w.lb.hi) + (float)w.rc * half_area(w.rb.lo, w.rb.hi);
        if (split_cost < cheapest_cost) { cheapest_cost = split_cost; cheapest = c; }
    }
    const float parent_cost = half_area(node.lo, node.hi) * (float)node.count;
    WideDecision d;
    d.split = (cheapest_cost >= parent_cost) ? 0u : 1u;                         // ref: BVH.cpp:253-256
    d.axis = cheapest % 3u;
    d.pos = candidate_pos(node, cheapest / 3u, d.axis);
    if (A.naive) { d.split = node.count <= 2u ? 0u : 1u; naive_plane(node, d.axis, d.pos); }   // ref: BVH.cpp:211-223
    uint32_t run = 0;
    for (uint32_t b = 0; b < pieces; ++b) {                                     // lefts before each piece, for the partition
        A.piece_left[(size_t)j * pieces + b] = run;
        run += A.partial[((size_t)j * pieces + b) * kCandidates + cheapest].lc;
    }
    d.n_left = run;
    A.decision[j] = d;
    atomicMax(&A.counters[1], node.depth);                                      // m_max_depth, ref: BVH.cpp:206
}

// the partition's position tables (see subdivide_level); leaves every piece's count of front lefts
__global__ void __launch_bounds__(kBuildThreads) wide_tables(BuildArrays A, uint32_t level_first, uint32_t pieces)
{
    __shared__ uint32_t s_u32[5];
    BuildNode node;
    const Piece q = piece_of(A, level_first, pieces, node);
    const WideDecision d = A.decision[q.j];
    if (d.split == 0u) return;
    const uint32_t* const idx = A.tri_indices + q.first;
    uint32_t my_left = 0;
    for (uint32_t i = q.c0; i < q.c1; ++i) my_left += (axis_of(A.centroid[idx[i]], d.axis) < d.pos) ? 1u : 0u;
    uint32_t piece_total = 0;
    uint32_t L = block_exclusive_scan(my_left, s_u32, &piece_total) + A.piece_left[(size_t)q.j * pieces + q.b];
    uint32_t* const hole_pos = A.sel_a + q.first;
    uint32_t* const tail_left_pos = A.sel_b + q.first;
    uint32_t front_left = 0;
    for (uint32_t p = q.c0; p < q.c1; ++p) {
        const bool is_left = axis_of(A.centroid[idx[p]], d.axis) < d.pos;
        if (p < d.n_left) { if (is_left) ++front_left; else hole_pos[p - L] = p; }
        else if (is_left) tail_left_pos[d.n_left - L - 1u] = p;
        L += is_left ? 1u : 0u;
    }
    front_left = block_reduce_sum(front_left, s_u32);
    if (threadIdx.x == 0) A.piece_front[(size_t)q.j * pieces + q.b] = front_left;
}

__global__ void __launch_bounds__(kBuildThreads) wide_write(BuildArrays A, uint32_t level_first, uint32_t pieces)
{
    __shared__ uint32_t s_u32[5];
    BuildNode node;
    const Piece q = piece_of(A, level_first, pieces, node);
    const WideDecision d = A.decision[q.j];
    if (d.split == 0u) return;
    const uint32_t n = q.n, n_left = d.n_left;
    const uint32_t* const idx = A.tri_indices + q.first;
    uint32_t fronts = 0;
    for (uint32_t b = threadIdx.x; b < pieces; b += kBuildThreads) fronts += A.piece_front[(size_t)q.j * pieces + b];
    fronts = block_reduce_sum(fronts, s_u32);
    if (threadIdx.x == 0) s_u32[4] = fronts;
    __syncthreads();
    const uint32_t holes = n_left - s_u32[4];                                   // rights in the front = lefts in the tail
    __syncthreads();
    uint32_t my_left = 0;
    for (uint32_t i = q.c0; i < q.c1; ++i) my_left += (axis_of(A.centroid[idx[i]], d.axis) < d.pos) ? 1u : 0u;
    uint32_t piece_total = 0;
    uint32_t L = block_exclusive_scan(my_left, s_u32, &piece_total) + A.piece_left[(size_t)q.j * pieces + q.b];
    const uint32_t* const hole_pos = A.sel_a + q.first;
    const uint32_t* const tail_left_pos = A.sel_b + q.first;
    uint32_t* const out = A.scratch_idx + q.first;
    for (uint32_t p = q.c0; p < q.c1; ++p) {
        const bool is_left = axis_of(A.centroid[idx[p]], d.axis) < d.pos;
        L += is_left ? 1u : 0u;                                                 // lefts at positions <= p
        uint32_t value;
        if (p < n_left) {
            value = is_left ? idx[p] : idx[tail_left_pos[p - L]];
        } else if (p + 1u == n || axis_of(A.centroid[idx[p + 1u]], d.axis) < d.pos) {
            const uint32_t k = n_left - L;
            value = k < holes ? idx[hole_pos[k]] : idx[n_left];
        } else {
            value = idx[p + 1u];
        }
        out[p] = value;
    }
}

// new order back into tri_indices; the piece's share of the children's bounds (CalculateNodeBounds in the NEW order, ref: BVH.cpp:350-362)
__global__ void __launch_bounds__(kBuildThreads) wide_finish_piece(BuildArrays A, uint32_t level_first, uint32_t pieces)
{
    __shared__ Bounds s_bounds[4];
    BuildNode node;
    const Piece q = piece_of(A, level_first, pieces, node);
    const WideDecision d = A.decision[q.j];
    if (d.split == 0u) return;
    uint32_t* const idx = A.tri_indices + q.first;
    const uint32_t* const out = A.scratch_idx + q.first;
    for (uint32_t p = q.c0; p < q.c1; ++p) idx[p] = out[p];
    if (d.n_left == 0u || d.n_left == q.n) return;                              // ref: BVH.cpp:346-348 (stays a leaf, in the new order)
    Bounds lb = empty_bounds(), rb = empty_bounds();
    for (uint32_t p = q.c0; p < q.c1; ++p) {
        const uint32_t tri = idx[p];
        const Bounds tb = { A.tri_lo[tri], A.tri_hi[tri] };
        if (p < d.n_left) lb = merge(lb, tb); else rb = merge(rb, tb);
    }
    lb = block_reduce_bounds(lb, s_bounds);
    rb = block_reduce_bounds(rb, s_bounds);
    if (threadIdx.x == 0) A.piece_child[(size_t)q.j * pieces + q.b] = { lb, rb };
}

__global__ void __launch_bounds__(64) wide_children(BuildArrays A, uint32_t level_first, uint32_t pieces)
{
    const uint32_t j = blockIdx.x, node_id = level_first + j;
    if (threadIdx.x != 0) return;
    const BuildNode node = A.nodes[node_id];
    const WideDecision d = A.decision[j];
    if (d.split == 0u || d.n_left == 0u || d.n_left == node.count) return;
    Bounds lb = empty_bounds(), rb = empty_bounds();
    for (uint32_t b = 0; b < pieces; ++b) {
        const WideChild w = A.piece_child[(size_t)j * pieces + b];
        lb = merge(lb, w.lb); rb = merge(rb, w.rb);
    }
    const uint32_t left_id = atomicAdd(&A.counters[0], 2u);
    BuildNode l{}, r{};
    l.lo = lb.lo; l.hi = lb.hi; l.first = node.first; l.count = d.n_left; l.depth = node.depth + 1u;
    r.lo = rb.lo; r.hi = rb.hi; r.first = node.first + d.n_left; r.count = node.count - d.n_left; r.depth = node.depth + 1u;
    A.nodes[left_id] = l; A.nodes[left_id + 1u] = r;
    A.nodes[node_id].left = left_id;
}

// ---- renumbering into the reference's allocation order --------------------------------------------------------------------------
__global__ void count_splits_level(BuildNode* nodes, uint32_t level_first, uint32_t level_count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= level_count) return;
    BuildNode& n = nodes[level_first + i];
    n.splits = n.left ? 1u + nodes[n.left].splits + nodes[n.left + 1u].splits : 0u;
}
__global__ void assign_ids_level(BuildNode* nodes, uint32_t level_first, uint32_t level_count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= level_count) return;
    const BuildNode& n = nodes[level_first + i];
    if (!n.left) return;
    BuildNode& l = nodes[n.left]; BuildNode& r = nodes[n.left + 1u];
    l.new_id = 1u + 2u * n.rank; r.new_id = l.new_id + 1u;       // children of the k-th splitting node: 2k+1, 2k+2
    l.rank = n.rank + 1u;                                         // preorder: the left subtree's splits come first
    r.rank = n.rank + 1u + l.splits;
}
__global__ void emit_nodes(const BuildNode* nodes, uint32_t n_nodes, cgpt_bvh_node* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const BuildNode& n = nodes[i];
    cgpt_bvh_node o;
    o.aabb_min[0] = n.lo.x; o.aabb_min[1] = n.lo.y; o.aabb_min[2] = n.lo.z;
    o.aabb_max[0] = n.hi.x; o.aabb_max[1] = n.hi.y; o.aabb_max[2] = n.hi.z;
    if (n.left) { o.left_first = nodes[n.left].new_id; o.prim_count = 0u; }
    else { o.left_first = n.first; o.prim_count = n.count; }
    out[n.new_id] = o;
}

// root bounds (CalculateNodeBounds over all triangles in index order, ref: BVH.cpp:43,188-202): one block, ordered
__global__ void __launch_bounds__(kBuildThreads) root_bounds(BuildArrays A, uint32_t n)
{
    __shared__ Bounds s_bounds[4];
    const uint32_t chunk = (n + kBuildThreads - 1) / kBuildThreads;
    const uint32_t c0 = min(threadIdx.x * chunk, n), c1 = min(c0 + chunk, n);
    Bounds b = empty_bounds();
    for (uint32_t i = c0; i < c1; ++i) { const uint32_t tri = A.tri_indices[i]; b = merge(b, Bounds{ A.tri_lo[tri], A.tri_hi[tri] }); }   // index order (a Rebuild starts from a permutation)
    b = block_reduce_bounds(b, s_bounds);
    if (threadIdx.x == 0) {
        BuildNode root{};
        root.lo = b.lo; root.hi = b.hi; root.first = 0; root.count = n; root.depth = 0; root.rank = 0; root.new_id = 0;
        A.nodes[0] = root;
        A.counters[0] = 1u; A.counters[1] = 0u;
    }
}

float HostTriangleArea(const cgpt_triangle& t)                                // Heron, ref: Primitives.cpp:270-278
{
    auto len = [](const float a[3], const float b[3]) { const float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2]; return sqrtf(x * x + y * y + z * z); };
    const float a = len(t.v1.pos, t.v0.pos), b = len(t.v2.pos, t.v0.pos), c = len(t.v2.pos, t.v1.pos);
    const float s = (a + b + c) / 2.0f;
    return sqrtf(s * (s - a) * (s - b) * (s - c));
}

}  // namespace

}  // namespace cgpt

using namespace cgpt;

static int BuildOnDevice(cgpt_ctx* ctx, const cgpt_triangle* triangles, uint32_t n_tris, uint32_t build_option, const uint32_t* initial_tri_indices,
                         cgpt_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* tri_indices_out, uint32_t* max_depth_out, float* total_area_out);

extern "C" int cgpt_bvh_build_ex(cgpt_ctx* ctx, const cgpt_triangle* triangles, uint32_t n_tris, uint32_t build_option, const uint32_t* initial_tri_indices,
                                 cgpt_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* tri_indices_out, uint32_t* max_depth_out, float* total_area_out)
{
    if (!ctx) return CGPT_ERR_INVALID;
    cgpt_ctx* const first = GroupFirstMemberOrNull(ctx);                       // a multi-device context builds on its first device
    cgpt_ctx* const target = first ? first : ctx;
    int rc;
    try { rc = BuildOnDevice(target, triangles, n_tris, build_option, initial_tri_indices, nodes_out, n_nodes_out, tri_indices_out, max_depth_out, total_area_out); }
    catch (const std::exception& e) { rc = CtxFail(target, CGPT_ERR_INVALID, "cgpt_bvh_build: %s", e.what()); }   // host vectors: nothing unwinds through the C ABI
    return first ? GroupForwarded(ctx, rc) : rc;
}

extern "C" int cgpt_bvh_build(cgpt_ctx* ctx, const cgpt_triangle* triangles, uint32_t n_tris, cgpt_bvh_node* nodes_out, uint32_t* n_nodes_out,
                              uint32_t* tri_indices_out, uint32_t* max_depth_out, float* total_area_out)
{
    return cgpt_bvh_build_ex(ctx, triangles, n_tris, CGPT_BUILD_SAH_SPLIT_INTERVALS, nullptr, nodes_out, n_nodes_out, tri_indices_out, max_depth_out, total_area_out);
}

static int BuildOnDevice(cgpt_ctx* ctx, const cgpt_triangle* triangles, uint32_t n_tris, uint32_t build_option, const uint32_t* initial_tri_indices,
                         cgpt_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* tri_indices_out, uint32_t* max_depth_out, float* total_area_out)
{
    if (!triangles || n_tris == 0 || !nodes_out || !n_nodes_out || !tri_indices_out || !max_depth_out || !total_area_out)
        return CtxFail(ctx, CGPT_ERR_INVALID, "cgpt_bvh_build: null argument or empty mesh");
    if (n_tris > 0x3FFFFFFFu) return CtxFail(ctx, CGPT_ERR_INVALID, "cgpt_bvh_build: too many triangles");
    if (build_option > CGPT_BUILD_SAH_SPLIT_PRIMITIVES) return CtxFail(ctx, CGPT_ERR_INVALID, "cgpt_bvh_build: unknown build option %u", build_option);
    if (initial_tri_indices) {                                                // a Rebuild's starting order must be a permutation: the kernels index with it
        std::vector<uint8_t> seen(n_tris, 0);
        for (uint32_t i = 0; i < n_tris; ++i) {
            const uint32_t t = initial_tri_indices[i];
            if (t >= n_tris || seen[t]) return CtxFail(ctx, CGPT_ERR_INVALID, "cgpt_bvh_build: initial_tri_indices[%u] = %u: not a permutation of 0..%u", i, t, n_tris - 1u);
            seen[t] = 1;
        }
    }
    hipStream_t stream = CtxStream(ctx);
    if (hipSetDevice(CtxDevice(ctx)) != hipSuccess) return CtxFail(ctx, CGPT_ERR_HIP, "cgpt_bvh_build: hipSetDevice failed");

#define BV_TRY(expr)                                                                                                          \
    do {                                                                                                                      \
        hipError_t e_ = (expr);                                                                                               \
        if (e_ != hipSuccess) { rc = CtxFail(ctx, CGPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)

    int rc = CGPT_OK;
    cgpt_triangle* d_tris = nullptr; F3 *d_lo = nullptr, *d_hi = nullptr, *d_c = nullptr;
    uint32_t *d_idx = nullptr, *d_scratch = nullptr, *d_sa = nullptr, *d_sb = nullptr, *d_counters = nullptr;
    BuildNode* d_nodes = nullptr; cgpt_bvh_node* d_out = nullptr;
    WidePartial* d_partial = nullptr; WideDecision* d_decision = nullptr; uint32_t *d_piece_left = nullptr, *d_piece_front = nullptr; WideChild* d_piece_child = nullptr;
    uint32_t quarter_tris = 32;                                               // ... and one quarter wavefront per node
    uint32_t wave_tris = 2048;                                                // average triangles per node up to which a level runs one wavefront per node
    uint32_t piece_tris = 512;                                                // average triangles per piece below which a level goes to one workgroup per node
    const uint32_t max_nodes = 2u * n_tris - 1u;
    std::vector<uint32_t> level_first;     // breadth-first ranges of the levels
    uint32_t counters[2] = { 0, 0 };
    uint32_t n_nodes = 0;
    BuildArrays A{};
    {
        BV_TRY(hipMalloc((void**)&d_tris, (size_t)n_tris * sizeof(cgpt_triangle)));
        BV_TRY(hipMalloc((void**)&d_lo, (size_t)n_tris * sizeof(F3)));
        BV_TRY(hipMalloc((void**)&d_hi, (size_t)n_tris * sizeof(F3)));
        BV_TRY(hipMalloc((void**)&d_c, (size_t)n_tris * sizeof(F3)));
        BV_TRY(hipMalloc((void**)&d_idx, (size_t)n_tris * 4)); BV_TRY(hipMalloc((void**)&d_scratch, (size_t)n_tris * 4));
        BV_TRY(hipMalloc((void**)&d_sa, (size_t)n_tris * 4)); BV_TRY(hipMalloc((void**)&d_sb, (size_t)n_tris * 4));
        BV_TRY(hipMalloc((void**)&d_counters, 2 * 4));
        BV_TRY(hipMalloc((void**)&d_nodes, (size_t)max_nodes * sizeof(BuildNode)));
        BV_TRY(hipMalloc((void**)&d_out, (size_t)max_nodes * sizeof(cgpt_bvh_node)));
        BV_TRY(hipMalloc((void**)&d_partial, (size_t)kWideBlocks * kCandidates * sizeof(WidePartial)));
        BV_TRY(hipMalloc((void**)&d_decision, (size_t)kWideBlocks * sizeof(WideDecision)));
        BV_TRY(hipMalloc((void**)&d_piece_left, (size_t)kWideBlocks * 4)); BV_TRY(hipMalloc((void**)&d_piece_front, (size_t)kWideBlocks * 4));
        BV_TRY(hipMalloc((void**)&d_piece_child, (size_t)kWideBlocks * sizeof(WideChild)));
        if (const char* e = getenv("CGPT_BVH_WAVE_TRIS")) wave_tris = (uint32_t)std::max(0l, strtol(e, nullptr, 10));     // tests: 0 = workgroups only
        if (const char* e = getenv("CGPT_BVH_QUARTER_TRIS")) quarter_tris = (uint32_t)std::max(0l, strtol(e, nullptr, 10));
        if (const char* e = getenv("CGPT_BVH_PIECE_TRIS")) piece_tris = (uint32_t)std::max(1l, strtol(e, nullptr, 10));   // tests: the wide path on small meshes
        BV_TRY(hipMemcpyAsync(d_tris, triangles, (size_t)n_tris * sizeof(cgpt_triangle), hipMemcpyHostToDevice, stream));
        if (initial_tri_indices) BV_TRY(hipMemcpyAsync(d_idx, initial_tri_indices, (size_t)n_tris * 4, hipMemcpyHostToDevice, stream));   // Rebuild: the current order
        A.naive = build_option == CGPT_BUILD_NAIVE_SPLIT ? 1u : 0u;
        A.tri_lo = d_lo; A.tri_hi = d_hi; A.centroid = d_c; A.tri_indices = d_idx; A.scratch_idx = d_scratch; A.sel_a = d_sa; A.sel_b = d_sb;
        A.nodes = d_nodes; A.counters = d_counters;
        A.partial = d_partial; A.decision = d_decision; A.piece_left = d_piece_left; A.piece_front = d_piece_front; A.piece_child = d_piece_child;
        hipLaunchKernelGGL(prepare_triangles, dim3((n_tris + 255u) / 256u), dim3(256), 0, stream, d_tris, n_tris, d_lo, d_hi, d_c, d_idx, initial_tri_indices ? 0u : 1u);
        hipLaunchKernelGGL(root_bounds, dim3(1), dim3(kBuildThreads), 0, stream, A, n_tris);
        // SAH split primitives (ref: BVH.cpp:260-297): cheapest_cost stays 1e30 because the loop never assigns it (SURVEY A-5), so
        // "cheapest_cost >= parent_cost" ends the build at the root whenever the root's cost is an ordinary number.  A root cost
        // of NaN or beyond 1e30 (bounds near the float limit) would take the reference into Split with the last candidate plane:
        // that corner is left to the host build.
        bool root_only = false;
        if (build_option == CGPT_BUILD_SAH_SPLIT_PRIMITIVES) {
            BuildNode root{};
            BV_TRY(hipMemcpyAsync(&root, d_nodes, sizeof(root), hipMemcpyDeviceToHost, stream));
            BV_TRY(hipStreamSynchronize(stream));
            const float ex = root.hi.x - root.lo.x, ey = root.hi.y - root.lo.y, ez = root.hi.z - root.lo.z;
            const float parent_cost = (ex * ey + ey * ez + ez * ex) * (float)n_tris;
            if (!(1e30f >= parent_cost)) { rc = CtxFail(ctx, CGPT_ERR_UNSUPPORTED, "cgpt_bvh_build: SAH-split-primitives on bounds whose cost is not below 1e30: use the host build"); goto done; }
            root_only = true;
        }

        // level by level: the nodes created while level L is processed are exactly level L + 1
        uint32_t first = 0, count = 1;
        if (root_only) { level_first.push_back(0); counters[0] = 1; counters[1] = 0; count = 0; }
        while (count > 0) {
            level_first.push_back(first);
            // few, big nodes: pieces of ~piece_tris triangles or more, at most kWideBlocks of them per level
            const uint32_t pieces = std::min(kWideBlocks / std::min(count, kWideBlocks), std::max(1u, n_tris / count / piece_tris));
            if (count <= kWideBlocks && pieces >= 2u) {
                const dim3 grid(count * pieces), per_node(count), block(kBuildThreads);
                hipLaunchKernelGGL(wide_sah_partial, grid, block, 0, stream, A, first, pieces);
                hipLaunchKernelGGL(wide_decide, per_node, dim3(64), 0, stream, A, first, pieces);
                hipLaunchKernelGGL(wide_tables, grid, block, 0, stream, A, first, pieces);
                hipLaunchKernelGGL(wide_write, grid, block, 0, stream, A, first, pieces);
                hipLaunchKernelGGL(wide_finish_piece, grid, block, 0, stream, A, first, pieces);
                hipLaunchKernelGGL(wide_children, per_node, dim3(64), 0, stream, A, first, pieces);
            } else if ((uint64_t)count * wave_tris < n_tris) {                  // big nodes on average: a workgroup per node
                hipLaunchKernelGGL(subdivide_level<kBuildThreads>, dim3(count), dim3(kBuildThreads), 0, stream, A, first, count);
            } else if ((uint64_t)count * quarter_tris < n_tris) {               // the deep levels: a wavefront per node
                hipLaunchKernelGGL(subdivide_level<64>, dim3((count + 3u) / 4u), dim3(kBuildThreads), 0, stream, A, first, count);
            } else {                                                          // the deepest: a handful of triangles per node, 16 lanes each
                hipLaunchKernelGGL(subdivide_level<16>, dim3((count + 15u) / 16u), dim3(kBuildThreads), 0, stream, A, first, count);
            }
            BV_TRY(hipMemcpyAsync(counters, d_counters, sizeof(counters), hipMemcpyDeviceToHost, stream));
            BV_TRY(hipStreamSynchronize(stream));
            first += count;
            count = counters[0] - first;
            if (level_first.size() > 4096) { rc = CtxFail(ctx, CGPT_ERR_INVALID, "cgpt_bvh_build: tree deeper than 4096 levels"); goto done; }
        }
        n_nodes = counters[0];
        level_first.push_back(n_nodes);
        // splitting-node counts bottom-up, preorder ranks and final ids top-down
        for (size_t l = level_first.size() - 1; l-- > 0;) {
            const uint32_t c = level_first[l + 1] - level_first[l];
            hipLaunchKernelGGL(count_splits_level, dim3((c + 255u) / 256u), dim3(256), 0, stream, d_nodes, level_first[l], c);
        }
        for (size_t l = 0; l + 1 < level_first.size(); ++l) {
            const uint32_t c = level_first[l + 1] - level_first[l];
            hipLaunchKernelGGL(assign_ids_level, dim3((c + 255u) / 256u), dim3(256), 0, stream, d_nodes, level_first[l], c);
        }
        hipLaunchKernelGGL(emit_nodes, dim3((n_nodes + 255u) / 256u), dim3(256), 0, stream, d_nodes, n_nodes, d_out);
        BV_TRY(hipGetLastError());
        BV_TRY(hipMemcpyAsync(nodes_out, d_out, (size_t)n_nodes * sizeof(cgpt_bvh_node), hipMemcpyDeviceToHost, stream));
        BV_TRY(hipMemcpyAsync(tri_indices_out, d_idx, (size_t)n_tris * 4, hipMemcpyDeviceToHost, stream));
        BV_TRY(hipStreamSynchronize(stream));
        *n_nodes_out = n_nodes;
        *max_depth_out = counters[1];
        float area = 0.0f;                                                    // m_total_area: a sequential float sum (ref: BVH.cpp:22)
        for (uint32_t i = 0; i < n_tris; ++i) area += HostTriangleArea(triangles[i]);
        *total_area_out = area;
    }
done:
    (void)hipFree(d_tris); (void)hipFree(d_lo); (void)hipFree(d_hi); (void)hipFree(d_c); (void)hipFree(d_idx); (void)hipFree(d_scratch);
    (void)hipFree(d_sa); (void)hipFree(d_sb); (void)hipFree(d_counters); (void)hipFree(d_nodes); (void)hipFree(d_out);
    (void)hipFree(d_partial); (void)hipFree(d_decision); (void)hipFree(d_piece_left); (void)hipFree(d_piece_front); (void)hipFree(d_piece_child);
#undef BV_TRY
    return rc;
}
