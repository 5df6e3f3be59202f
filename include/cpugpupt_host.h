/*
 * cpugpupt_host.h -- C entry points of the host-side mirror (scene structs, BVH build, glTF load path,
 * synthetic meshes, framebuffer dump).  Pure CPU code, no HIP calls: this is the part of the reference that
 * "stays host C++" (ref: Source/Main.cpp:51-275,775-819; Source/BVH.cpp:11-59,188-366; Source/GLTFLoader.cpp),
 * exported with a C ABI so non-C++ hosts (the Python tests and bench) can build scenes and hand the
 * flattened cgpt_scene_desc to cgpt_scene_upload() in cpugpupt_abi.h.
 *
 * Status convention as in cpugpupt_abi.h: 0 = ok, message from cgpth_last_error() (thread-local).
 */
#ifndef CPUGPUPT_HOST_H
#define CPUGPUPT_HOST_H

#include "cpugpupt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref: Include/BVH.h:7-13 */
enum cgpth_build_option { CGPTH_BUILD_NAIVE = 0, CGPTH_BUILD_SAH_INTERVALS = 1, CGPTH_BUILD_SAH_PRIMITIVES = 2 };

typedef struct cgpth_mesh cgpth_mesh;    /* ref: Include/Primitives.h:24-28 (Mesh) */
typedef struct cgpth_scene cgpth_scene;  /* ref: Source/Main.cpp:200-236 (the parts of `data` Render() reads) */

typedef struct cgpth_bvh_info {
    uint32_t num_triangles, nodes_used, num_leaves, max_leaf_size, max_depth;
    float total_area;
} cgpth_bvh_info;

const char* cgpth_last_error(void);

/* ---- meshes ---- */
/* GLTFLoader::Load (ref: Source/GLTFLoader.cpp:19-89); NULL + error on failure */
cgpth_mesh* cgpth_mesh_load_gltf(const char* path);
cgpth_mesh* cgpth_mesh_from_arrays(const cgpt_vertex* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices);
/* synthetic dragon stand-in: icosphere `level` -> 20*4^level triangles inside the dragon's AABB */
cgpth_mesh* cgpth_mesh_dragon_standin(uint32_t level);
cgpth_mesh* cgpth_mesh_bumpy_icosphere(uint32_t level, const float center[3], const float radii[3], float bump);
int cgpth_mesh_save_gltf(const cgpth_mesh* mesh, const char* gltf_path);
uint32_t cgpth_mesh_num_vertices(const cgpth_mesh* mesh);
uint32_t cgpth_mesh_num_indices(const cgpth_mesh* mesh);
const cgpt_vertex* cgpth_mesh_vertices(const cgpth_mesh* mesh);
const uint32_t* cgpth_mesh_indices(const cgpth_mesh* mesh);
void cgpth_mesh_free(cgpth_mesh* mesh);

/* ---- scene ---- */
cgpth_scene* cgpth_scene_new(void);
void cgpth_scene_free(cgpth_scene* scene);
/* the shipped scene (ref: Main.cpp:777-819) with `mesh` in place of the dragon and material `mesh_material` on it */
cgpth_scene* cgpth_scene_reference_layout(const cgpth_mesh* mesh, uint32_t mesh_material, float aspect, int build_option);
int cgpth_scene_add_material(cgpth_scene* scene, const cgpt_material* material);            /* returns index */
int cgpth_scene_set_material(cgpth_scene* scene, uint32_t index, const cgpt_material* material);
int cgpth_scene_add_mesh(cgpth_scene* scene, const cgpth_mesh* mesh, uint32_t mat_index, int build_option);  /* Object ctor, ref: Main.cpp:247-251; returns object index */
/* same result as cgpth_scene_add_mesh(..., CGPTH_BUILD_SAH_INTERVALS) with the tree built on the GPU by cgpt_bvh_build
 * (bit-identical tree; the host keeps validating and owning it) */
int cgpth_scene_add_mesh_device_built(cgpth_scene* scene, const cgpth_mesh* mesh, uint32_t mat_index, cgpt_ctx* ctx);
/* any BuildOption on the GPU (cgpt_bvh_build_ex; ref: BVH.cpp:208-297) */
int cgpth_scene_add_mesh_device_built_ex(cgpth_scene* scene, const cgpth_mesh* mesh, uint32_t mat_index, cgpt_ctx* ctx, int build_option);
int cgpth_scene_add_sphere(cgpth_scene* scene, const float center[3], float radius, uint32_t mat_index);
int cgpth_scene_add_plane(cgpth_scene* scene, const float normal[3], const float point[3], uint32_t mat_index);
int cgpth_scene_add_light(cgpth_scene* scene, uint32_t obj_index);                            /* ref: Main.cpp:817 */
int cgpth_scene_set_camera(cgpth_scene* scene, const float pos[3], const float view_dir[3], float fov_deg, float aspect);
int cgpth_scene_set_settings(cgpth_scene* scene, const cgpt_settings* settings);
int cgpth_scene_rebuild_bvh(cgpth_scene* scene, uint32_t obj_index, int build_option);       /* ref: BVH.cpp:47-59 */
/* BVH::Rebuild with the re-split on the GPU: starts from the tree's current triangle order, as the reference does (ref: BVH.cpp:47-59);
 * on failure the BVH is left unchanged */
int cgpth_scene_rebuild_bvh_device(cgpth_scene* scene, uint32_t obj_index, int build_option, cgpt_ctx* ctx);
int cgpth_scene_bvh_info(const cgpth_scene* scene, uint32_t obj_index, cgpth_bvh_info* out);
/* nodes: nodes_used x cgpt_bvh_node; tri_indices: num_triangles */
int cgpth_scene_bvh_export(const cgpth_scene* scene, uint32_t obj_index, cgpt_bvh_node* nodes, uint32_t* tri_indices);
/* flatten for cgpt_scene_upload; pointers stay valid until the scene is changed or freed */
int cgpth_scene_flatten(cgpth_scene* scene, cgpt_scene_desc* out);
int cgpth_scene_get_camera(const cgpth_scene* scene, cgpt_camera* out);
int cgpth_scene_get_settings(const cgpth_scene* scene, cgpt_settings* out);

/* ---- framebuffer dump (replaces DX12 present, ref: Source/DX12.cpp:277-322) ---- */
int cgpth_write_ppm(const char* path, const uint32_t* pixels, uint32_t width, uint32_t height);
int cgpth_write_pfm(const char* path, const float* accumulator_rgba, uint32_t num_accumulated, uint32_t width, uint32_t height);
int cgpth_write_accumulator(const char* path, const float* accumulator_rgba, uint32_t num_accumulated, uint32_t width, uint32_t height);
int cgpth_read_accumulator(const char* path, float* accumulator_rgba, uint32_t* num_accumulated, uint32_t width, uint32_t height);

/* ---- self-check hook ---- */
/* the exact division-by-a-launch-constant the kernels use to map a path id to its pixel (csrc/device/fast_div.h), evaluated on
 * the host: equals n / d for every 32-bit n and every d >= 1 (tests/test_host.py checks it against integer division) */
uint32_t cgpth_fast_div(uint32_t n, uint32_t d);

#ifdef __cplusplus
}
#endif
#endif
