// mesh_bvh.cpp -- host BVH build reproducing the reference tree bit for bit (see mesh_bvh.h).
//
// Same decisions as ref: Source/BVH.cpp:11-59,188-366, organised differently: an explicit depth-first
// work list instead of recursion (children are numbered in the reference's allocation order: left,
// right, then the whole left subtree before the right one) and cached per-triangle bounds (min/max are
// exact and return the left-most tied operand under any grouping, so no bit changes).
#include "mesh_bvh.h"

#include <utility>

namespace cgpt {

namespace {

inline Vec3 P(const float p[3]) { return { p[0], p[1], p[2] }; }

// Heron's formula, ref: Source/Primitives.cpp:270-278
float TriangleArea(const cgpt_triangle& t)
{
    float a = length(P(t.v1.pos) - P(t.v0.pos));
    float b = length(P(t.v2.pos) - P(t.v0.pos));
    float c = length(P(t.v2.pos) - P(t.v1.pos));
    float s = (a + b + c) / 2.0f;
    return sqrtf(s * (s - a) * (s - b) * (s - c));
}

// the reference's SAH "volume" is the half surface area, ref: Source/Primitives.cpp:280-284 (SURVEY A-6)
inline float HalfArea(const Vec3& lo, const Vec3& hi)
{
    Vec3 e = hi - lo;
    return e.x * e.y + e.y * e.z + e.z * e.x;
}

}  // namespace

bool MeshBVH::SetTriangles(const std::vector<cgpt_vertex>& vertices, const std::vector<uint32_t>& indices)
{
    nodes_.clear(); triangles_.clear(); tri_indices_.clear(); centroids_.clear(); tri_bounds_.clear();
    nodes_used_ = 0; max_depth_ = 0; total_area_ = 0.0f;

    const size_t n = indices.size() / 3;
    if (n == 0) return false;
    for (size_t k = 0; k < n * 3; ++k)
        if (indices[k] >= vertices.size()) return false;

    triangles_.resize(n);
    tri_indices_.resize(n);
    centroids_.resize(n);
    tri_bounds_.resize(n);
    for (size_t i = 0; i < n; ++i) {
        cgpt_triangle& t = triangles_[i];
        t.v0 = vertices[indices[3 * i]];
        t.v1 = vertices[indices[3 * i + 1]];
        t.v2 = vertices[indices[3 * i + 2]];
        total_area_ += TriangleArea(t);                                       // ref: BVH.cpp:22
        tri_indices_[i] = (uint32_t)i;
        Vec3 p0 = P(t.v0.pos), p1 = P(t.v1.pos), p2 = P(t.v2.pos);
        centroids_[i] = (p0 + p1 + p2) * 0.3333f;                             // ref: Primitives.cpp:255-258 (SURVEY A-10)
        tri_bounds_[i].lo = vmin(vmin(p0, p1), p2);                           // ref: Primitives.cpp:232-243
        tri_bounds_[i].hi = vmax(vmax(p0, p1), p2);
    }
    nodes_.assign(2 * n - 1, cgpt_bvh_node{});                                // ref: BVH.cpp:37
    return true;
}

bool MeshBVH::Build(const std::vector<cgpt_vertex>& vertices, const std::vector<uint32_t>& indices, BuildOption option)
{
    option_ = option;
    if (!SetTriangles(vertices, indices)) return false;
    BuildTree();
    return true;
}

// adopt only a well-formed tree: a malformed one would send the device traversal out of bounds
bool MeshBVH::WellFormed(const cgpt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* tri_indices, uint32_t n)
{
    if (n_nodes < 1 || n_nodes > 2 * n - 1) return false;
    std::vector<uint8_t> seen(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        if (tri_indices[i] >= n || seen[tri_indices[i]]) return false;
        seen[tri_indices[i]] = 1;
    }
    for (uint32_t i = 0; i < n_nodes; ++i) {
        const cgpt_bvh_node& node = nodes[i];
        if (node.prim_count == 0) { if (!(node.left_first > i && node.left_first + 1 < n_nodes)) return false; }
        else if (!(node.left_first < n && node.prim_count <= n - node.left_first)) return false;
    }
    return true;
}

bool MeshBVH::BuildWith(const std::vector<cgpt_vertex>& vertices, const std::vector<uint32_t>& indices, BuildOption option, const TreeBuilder& build)
{
    option_ = option;
    if (!SetTriangles(vertices, indices)) return false;
    const uint32_t n = (uint32_t)triangles_.size();
    uint32_t n_nodes = 0, depth = 0;
    const bool ok = build(triangles_.data(), n, (int)option, nullptr, nodes_.data(), &n_nodes, tri_indices_.data(), &depth) &&
                    WellFormed(nodes_.data(), n_nodes, tri_indices_.data(), n);
    if (!ok) {
        nodes_.clear(); triangles_.clear(); tri_indices_.clear(); centroids_.clear(); tri_bounds_.clear();
        nodes_used_ = 0; max_depth_ = 0; total_area_ = 0.0f;
        return false;
    }
    nodes_used_ = n_nodes;
    max_depth_ = depth;
    return true;
}

bool MeshBVH::RebuildWith(BuildOption option, const TreeBuilder& build)       // ref: BVH.cpp:47-59
{
    if (triangles_.empty()) return false;
    const uint32_t n = (uint32_t)triangles_.size();
    std::vector<cgpt_bvh_node> nodes(2 * n - 1);
    std::vector<uint32_t> order(n);
    uint32_t n_nodes = 0, depth = 0;
    if (!build(triangles_.data(), n, (int)option, tri_indices_.data(), nodes.data(), &n_nodes, order.data(), &depth) ||
        !WellFormed(nodes.data(), n_nodes, order.data(), n))
        return false;
    option_ = option;
    nodes_.swap(nodes); tri_indices_.swap(order);
    nodes_used_ = n_nodes;
    max_depth_ = depth;                                                       // m_total_area is not recomputed by Rebuild
    return true;
}

void MeshBVH::Rebuild(BuildOption option)
{
    if (triangles_.empty()) return;
    option_ = option;
    nodes_used_ = 0;
    max_depth_ = 0;
    BuildTree();
}

void MeshBVH::FitNode(uint32_t node_index)                                    // ref: BVH.cpp:188-202
{
    cgpt_bvh_node& node = nodes_[node_index];
    Vec3 lo(1e30f), hi(-1e30f);
    for (uint32_t i = node.left_first; i < node.left_first + node.prim_count; ++i) {
        const Bounds& tb = tri_bounds_[tri_indices_[i]];
        lo = vmin(lo, tb.lo);
        hi = vmax(hi, tb.hi);
    }
    node.aabb_min[0] = lo.x; node.aabb_min[1] = lo.y; node.aabb_min[2] = lo.z;
    node.aabb_max[0] = hi.x; node.aabb_max[1] = hi.y; node.aabb_max[2] = hi.z;
}

float MeshBVH::SplitCost(const cgpt_bvh_node& node, uint32_t axis, float pos) const  // ref: BVH.cpp:299-327
{
    Bounds left, right;
    uint32_t n_left = 0, n_right = 0;
    for (uint32_t i = node.left_first; i < node.left_first + node.prim_count; ++i) {
        const uint32_t tri = tri_indices_[i];
        const Bounds& tb = tri_bounds_[tri];
        if (centroids_[tri][axis] < pos) { ++n_left; left.lo = vmin(left.lo, tb.lo); left.hi = vmax(left.hi, tb.hi); }
        else { ++n_right; right.lo = vmin(right.lo, tb.lo); right.hi = vmax(right.hi, tb.hi); }
    }
    // an empty side has extent -2e30 -> +inf area -> 0 * inf = NaN, which every "<" below rejects
    return (float)n_left * HalfArea(left.lo, left.hi) + (float)n_right * HalfArea(right.lo, right.hi);
}

bool MeshBVH::ChooseSplit(uint32_t node_index, uint32_t& axis, float& pos) const
{
    const cgpt_bvh_node& node = nodes_[node_index];
    const Vec3 lo = P(node.aabb_min), hi = P(node.aabb_max);

    if (option_ == BuildOption_NaiveSplit) {                                  // ref: BVH.cpp:208-224
        if (node.prim_count <= 2) return false;
        Vec3 extent = hi - lo;
        axis = 0;
        if (extent.y > extent.x) axis = 1;
        if (extent.z > extent[axis]) axis = 2;
        pos = lo[axis] + extent[axis] * 0.5f;
        return true;
    }

    const float parent_cost = HalfArea(lo, hi) * (float)node.prim_count;
    float best_cost = 1e30f;
    axis = 0; pos = 0.0f;

    if (option_ == BuildOption_SAHSplitIntervals) {                           // ref: BVH.cpp:225-259
        for (uint32_t k = 0; k < 8; ++k) {
            for (uint32_t a = 0; a < 3; ++a) {
                float width = hi[a] - lo[a];
                float candidate = width * ((float)k / 8) + lo[a];
                float cost = SplitCost(node, a, candidate);
                if (cost < best_cost) { best_cost = cost; axis = a; pos = candidate; }
            }
        }
    } else {                                                                  // ref: BVH.cpp:260-296
        for (uint32_t i = node.left_first; i < node.left_first + node.prim_count; ++i) {
            const Vec3& c = centroids_[tri_indices_[i]];
            for (uint32_t a = 0; a < 3; ++a) {
                float cost = SplitCost(node, a, c[a]);
                // the reference records axis/pos but never the cost (SURVEY A-5), so best_cost stays 1e30
                if (cost < best_cost) { axis = a; pos = c[a]; }
            }
        }
    }
    return !(best_cost >= parent_cost);                                       // ref: BVH.cpp:253,290
}

uint32_t MeshBVH::Partition(const cgpt_bvh_node& node, uint32_t axis, float pos)  // ref: BVH.cpp:331-344
{
    int32_t i = (int32_t)node.left_first;
    int32_t j = i + (int32_t)node.prim_count - 1;
    while (i <= j) {
        if (centroids_[tri_indices_[i]][axis] < pos) ++i;
        else std::swap(tri_indices_[i], tri_indices_[j--]);
    }
    return (uint32_t)i;
}

void MeshBVH::BuildTree()
{
    cgpt_bvh_node& root = nodes_[nodes_used_++];                              // ref: BVH.cpp:39-44
    root.left_first = 0;
    root.prim_count = (uint32_t)triangles_.size();
    FitNode(0);

    struct Work { uint32_t node, depth; };
    std::vector<Work> todo;
    todo.push_back({ 0, 0 });
    while (!todo.empty()) {
        const Work w = todo.back();
        todo.pop_back();
        if (w.depth > max_depth_) max_depth_ = w.depth;                       // ref: BVH.cpp:206

        uint32_t axis; float pos;
        if (!ChooseSplit(w.node, axis, pos)) continue;

        cgpt_bvh_node& node = nodes_[w.node];
        const uint32_t mid = Partition(node, axis, pos);
        const uint32_t n_left = mid - node.left_first;
        if (n_left == 0 || n_left == node.prim_count) continue;               // ref: BVH.cpp:346-348

        const uint32_t left = nodes_used_++, right = nodes_used_++;           // ref: BVH.cpp:350-359
        nodes_[left].left_first = node.left_first;
        nodes_[left].prim_count = n_left;
        nodes_[right].left_first = mid;
        nodes_[right].prim_count = node.prim_count - n_left;
        node.left_first = left;
        node.prim_count = 0;
        FitNode(left);
        FitNode(right);
        todo.push_back({ right, w.depth + 1 });                               // left subtree is numbered first
        todo.push_back({ left, w.depth + 1 });
    }
}

}  // namespace cgpt
