// mesh_bvh.h -- host BVH over a triangle mesh, producing the reference's exact tree.
//
// Mirrors the public surface of the reference's BVH class (ref: Include/BVH.h:15-26: Build, Rebuild,
// GetTriangle, NumTriangles, GetMaxDepth, GetTotalArea) and adds the flatten accessors a device upload
// needs (the reference keeps m_nodes / m_tri_indices private, SURVEY 8b "Ownership").  Traversal is not
// here: it runs on the GPU (csrc/device).  The tree (node numbering, bounds, triangle order) is
// bit-identical to the reference's for all three build options, including the option that never splits
// (SURVEY A-5) -- it defines traversal order and therefore image parity.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

#include "cpugpupt_abi.h"
#include "vec.h"

namespace cgpt {

struct Mesh {  // ref: Include/Primitives.h:24-28
    std::vector<cgpt_vertex> vertices;
    std::vector<uint32_t> indices;
};

class MeshBVH {
public:
    enum BuildOption : int {  // ref: Include/BVH.h:7-13
        BuildOption_NaiveSplit = 0,
        BuildOption_SAHSplitIntervals = 1,
        BuildOption_SAHSplitPrimitives = 2,
        BuildOption_NumOptions
    };

    // ref: BVH.cpp:11-45.  Returns false (and leaves the BVH empty) on an empty mesh or an out-of-range index.
    bool Build(const std::vector<cgpt_vertex>& vertices, const std::vector<uint32_t>& indices, BuildOption option);
    // Same as Build(..., option) with the tree construction delegated (the GPU build, cgpt_bvh_build_ex):
    // build(triangles, n, option, initial_tri_indices or nullptr, nodes[2n-1], &n_nodes, tri_indices[n], &max_depth) returns false on
    // failure.  The returned arrays are validated (child links, leaf ranges, permutation) before they are adopted.
    using TreeBuilder = std::function<bool(const cgpt_triangle*, uint32_t, int, const uint32_t*, cgpt_bvh_node*, uint32_t*, uint32_t*, uint32_t*)>;
    bool BuildWith(const std::vector<cgpt_vertex>& vertices, const std::vector<uint32_t>& indices, BuildOption option, const TreeBuilder& build);
    // Rebuild (below) with the re-split delegated: the builder starts from the current triangle order.  On failure the tree is unchanged.
    bool RebuildWith(BuildOption option, const TreeBuilder& build);
    // ref: BVH.cpp:47-59: re-split over the current triangle order (the order is NOT reset, as in the reference)
    void Rebuild(BuildOption option);

    const cgpt_triangle& GetTriangle(uint32_t index) const { return triangles_[index]; }
    uint32_t NumTriangles() const { return (uint32_t)triangles_.size(); }
    uint32_t GetMaxDepth() const { return max_depth_; }
    float GetTotalArea() const { return total_area_; }

    // flatten accessors (new): nodes in use, triangle order, triangles
    uint32_t NumNodes() const { return nodes_used_; }
    const cgpt_bvh_node* Nodes() const { return nodes_.data(); }
    const uint32_t* TriIndices() const { return tri_indices_.data(); }
    const cgpt_triangle* Triangles() const { return triangles_.data(); }
    BuildOption CurrentBuildOption() const { return option_; }

private:
    struct Bounds { Vec3 lo{ 1e30f }, hi{ -1e30f }; };
    bool SetTriangles(const std::vector<cgpt_vertex>& vertices, const std::vector<uint32_t>& indices);
    static bool WellFormed(const cgpt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* tri_indices, uint32_t n);
    void BuildTree();
    void FitNode(uint32_t node_index);
    bool ChooseSplit(uint32_t node_index, uint32_t& axis, float& pos) const;
    float SplitCost(const cgpt_bvh_node& node, uint32_t axis, float pos) const;
    uint32_t Partition(const cgpt_bvh_node& node, uint32_t axis, float pos);

    BuildOption option_ = BuildOption_SAHSplitIntervals;
    std::vector<cgpt_bvh_node> nodes_;
    uint32_t nodes_used_ = 0;
    uint32_t max_depth_ = 0;
    float total_area_ = 0.0f;
    std::vector<cgpt_triangle> triangles_;
    std::vector<uint32_t> tri_indices_;
    std::vector<Vec3> centroids_;
    std::vector<Bounds> tri_bounds_;  // per-triangle AABB: min/max are exact, so caching them changes no bit
};

}  // namespace cgpt
