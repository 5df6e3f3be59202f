// device_scene.h -- how the scene lives in HBM (gfx950), shared by the upload code and the kernels.
//
// The C ABI hands over the reference's own AoS layouts (32-byte BVHNode, 72-byte Triangle, 56-byte Material).
// The upload step (cgpt_abi.hip: BuildDeviceScene) re-lays them for per-lane gathers:
//
//  node_pairs  float4[4 * n_pairs]  one 64-byte, 64-byte-aligned record per INNER node = its two children, left / right
//                                   interleaved per component:
//                                   {lmin.x, rmin.x, lmin.y, rmin.y | lmin.z, rmin.z, lmax.x, rmax.x | lmax.y, rmax.y, lmax.z, rmax.z | -, -, lcode, rcode}.
//                                   (the codes sit at byte 56: an 8-byte load there is only 8-byte aligned, so the compiler cannot
//                                   widen it to 16 bytes -- the texture data path is paid per byte returned)
//                                   The reference reads nodes[left_first] and nodes[left_first+1] (adjacent, ref: BVH.cpp:93-94);
//                                   here that is one aligned fetch of 56 useful bytes, each {left, right} pair sits in an even
//                                   VGPR pair so the slab test's subtract and multiply are v_pk_add_f32 / v_pk_mul_f32 on both
//                                   children at once, and the child's {left_first, prim_count} pair is pre-folded into a
//                                   32-bit traversal code so a stack entry is one LDS dword.
//  tri_leaf    float4[3 * n_tris]   triangles in LEAF order (m_tri_indices order, ref: BVH.cpp:76), 48 B each:
//                                   {v0.xyz, e1.x | e1.yz, e2.xy | -, e2.z, tri_idx, last_in_leaf} (the last three are one
//                                   12-byte load at byte 36); e1 = v1-v0, e2 = v2-v0 are
//                                   the reference's own first two subtractions (ref: Primitives.cpp:9-10), done once at upload
//                                   (same IEEE operation, same bits).  The indirection through m_tri_indices disappears.
//  tri_orig    float4[3 * n_tris]   triangles in ORIGINAL order for GetTriangle() users (ref: BVH.cpp:129-132): shading normal
//                                   = v0.normal (ref: Primitives.cpp:148-151) and mesh-light sampling (ref: Primitives.cpp:170-186):
//                                   {p0.xyz, n0.x | p1.xyz, n0.y | p2.xyz, n0.z}
//  tri_normal  float4[n_tris]       {n0.xyz, -} in ORIGINAL order: the shading normal of a hit is one 16-byte load
//  materials   float4[4 * n_mat]    {albedo.xyz, specular | refractivity, absorption.xyz | ior, emissive.xyz | intensity, is_light, -, -}
//  objects     DevObject[n]         read with wave-uniform indices (scalar loads)
//  obj_trace   float4[2 * n]        what IntersectScene's object loop needs of object i, for per-lane object indices:
//                                   {kind, p0, p1, p2 | p3, p4, p5, -}: mesh p0 = root code; sphere p0..2 = centre, p3 = radius^2;
//                                   plane p0..2 = normal, p3..5 = point
//
// record order: a child-pair record's index is only a name (the codes inside the records and the root codes are the only
// references to it), so the upload renumbers them: records [0, n_top_records) are the top levels of all meshes' trees in
// breadth-first order -- the part of the tree every ray walks; the trace kernel keeps a copy of them in LDS -- and the rest
// follow in the reference's depth-first allocation order (a parent next to its left subtree).
// Leaf-triangle records: the triangles of SMALL meshes (at most kSmallMeshTris triangles each, kLdsTrisMax in total -- the ground quad
// of the reference scene, ref: Main.cpp:789-800, whose two triangles every ray of the scene tests) come first, records
// [0, n_small_tris); the voted trace kernels read those from an LDS copy, which takes two of the ~2.8 triangle fetches per ray off
// the vector-memory path.  Larger meshes follow in object order.
//
// traversal code: bit 31 clear -> index of a child-pair record; bit 31 set -> index (into tri_leaf records) of the first
// triangle of a leaf, whose last triangle carries last_in_leaf = 1.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace cgpt {

static constexpr uint32_t kLeafBit = 0x80000000u;
static constexpr uint32_t kNoHit = 0xFFFFFFFFu;
#ifndef CGPT_TOP_RECORDS_MAX
#define CGPT_TOP_RECORDS_MAX 256
#endif
static constexpr uint32_t kSmallMeshTris = 8;    // a mesh of at most this many triangles is "small" ...
static constexpr uint32_t kLdsTrisMax = 16;      // ... and at most this many leaf-triangle records are mirrored in LDS
static constexpr uint32_t kTopRecords = CGPT_TOP_RECORDS_MAX;   // most records renumbered to the front in breadth-first order (8 full levels of one tree)

struct DevObject {
    uint32_t kind;        // cgpt_object_kind
    uint32_t mat_index;
    uint32_t root_code;   // traversal code of the root (a leaf code when the root never split)
    uint32_t tri_base;    // first record of this mesh in tri_orig (object-local tri_idx + tri_base)
    uint32_t n_tris;
    float total_area;
    float sphere_radius, sphere_radius_sq;
    float sphere_center[3];
    float plane_normal[3];
    float plane_point[3];
    uint32_t pad_;
};

struct DevScene {
    const float4* node_pairs;
    const float4* tri_leaf;
    const float4* tri_orig;
    const float4* tri_normal;
    const float4* materials;
    const DevObject* objects;
    const float4* obj_trace;
    const uint32_t* lights;
    uint32_t n_objects;
    uint32_t n_lights;
    uint32_t stack_depth;  // LDS stack entries per lane (max BVH depth + 1 over all meshes)
    uint32_t n_top_records; // records [0, n_top_records) are the breadth-first top of the trees
    uint32_t n_pair_records; // child-pair records in node_pairs (the plane stride of the -DCGPT_NODE_SOA experiment build)
    uint32_t n_small_tris;  // tri_leaf records [0, n_small_tris) are the triangles of the scene's small meshes (device_scene.h "record order")
};

struct DevCamera { float pos[3], top_left[3], top_right[3], bottom_left[3]; };

struct DevSettings {
    int32_t max_ray_depth;
    uint32_t nee, cosine, rr;
    uint32_t render_mode, debug_mode;
};

__host__ __device__ inline uint32_t GlobalRow(uint32_t l, uint32_t band_first, uint32_t band_h, uint32_t band_stride)
{
    return band_first + (l / band_h) * band_stride + l % band_h;
}

struct DevCounters {  // device-side totals, 64-bit atomics
    unsigned long long traced_rays, inner_steps, tri_tests, bvh_depth_sum, closest_hits;
    double total_energy;
};

struct DevRenderArgs {
    DevScene scene;
    DevCamera camera;
    DevSettings settings;
    uint32_t width, height;
    // rows of this context: local row l (0 <= l < n_rows) is global row band_first + (l / band_h) * band_stride + l % band_h.
    // Contiguous band [row_begin,row_end): band_first = row_begin, band_h = n_rows.  Interleaved: band_h = h, band_stride = R*h.
    uint32_t n_rows, band_first, band_h, band_stride;
    uint32_t first_sample, n_samples, seed;
    float4* accumulator;   // band-local: n_rows x width
    uint32_t* pixels;
    DevCounters* counters;
};

}  // namespace cgpt
