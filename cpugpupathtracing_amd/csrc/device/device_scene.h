// device_scene.h -- how the scene lives in HBM (gfx950), shared by the upload code and the kernels.
//
// The C ABI hands over the reference's own AoS layouts (32-byte BVHNode, 72-byte Triangle, 56-byte Material).
// The upload step (cgpt_abi.hip: BuildDeviceScene) re-lays them for per-lane gathers:
//
//  node_pairs  float4[4 * n_pairs]  one 64-byte, 64-byte-aligned record per INNER node = its two children
//                                   {lmin.xyz, lcode | lmax.xyz, - | rmin.xyz, rcode | rmax.xyz, -}.  The reference reads
//                                   nodes[left_first] and nodes[left_first+1] (adjacent, ref: BVH.cpp:93-94); here that is one
//                                   aligned 64-B fetch, and the child's {left_first, prim_count} pair is pre-folded into a
//                                   32-bit traversal code so a stack entry is one LDS dword.
//  tri_leaf    float4[3 * n_tris]   triangles in LEAF order (m_tri_indices order, ref: BVH.cpp:76), 48 B each:
//                                   {v0.xyz, e1.x | e1.yz, e2.xy | e2.z, tri_idx, last_in_leaf, -}; e1 = v1-v0, e2 = v2-v0 are
//                                   the reference's own first two subtractions (ref: Primitives.cpp:9-10), done once at upload
//                                   (same IEEE operation, same bits).  The indirection through m_tri_indices disappears.
//  tri_orig    float4[3 * n_tris]   triangles in ORIGINAL order for GetTriangle() users (ref: BVH.cpp:129-132): shading normal
//                                   = v0.normal (ref: Primitives.cpp:148-151) and mesh-light sampling (ref: Primitives.cpp:170-186):
//                                   {p0.xyz, n0.x | p1.xyz, n0.y | p2.xyz, n0.z}
//  tri_normal  float4[n_tris]       {n0.xyz, -} in ORIGINAL order: the shading normal of a hit is one 16-byte load
//  materials   float4[4 * n_mat]    {albedo.xyz, specular | refractivity, absorption.xyz | ior, emissive.xyz | intensity, is_light, -, -}
//  objects     DevObject[n]         read with wave-uniform indices (scalar loads)
//
// traversal code: bit 31 clear -> index of a child-pair record; bit 31 set -> index (into tri_leaf records) of the first
// triangle of a leaf, whose last triangle carries last_in_leaf = 1.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace cgpt {

static constexpr uint32_t kLeafBit = 0x80000000u;
static constexpr uint32_t kNoHit = 0xFFFFFFFFu;

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
    const uint32_t* lights;
    uint32_t n_objects;
    uint32_t n_lights;
    uint32_t stack_depth;  // LDS stack entries per lane (max BVH depth + 1 over all meshes)
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
