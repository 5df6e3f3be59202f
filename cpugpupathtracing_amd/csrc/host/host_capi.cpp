// host_capi.cpp -- C entry points of the host mirror (include/cpugpupt_host.h).
#include <cstring>
#include <exception>
#include <new>
#include <string>

#include "cpugpupt_host.h"
#include "fast_div.h"
#include "gltf_loader.h"
#include "image_io.h"
#include "mesh_gen.h"
#include "scene.h"

using namespace cgpt;

struct cgpth_mesh { Mesh mesh; };
struct cgpth_scene {
    Scene scene;
    Scene::FlatStorage flat;
};

namespace {
thread_local std::string g_error;
int Fail(const std::string& msg) { g_error = msg; return CGPT_ERR_INVALID; }
Vec3 V(const float p[3]) { return { p[0], p[1], p[2] }; }

Material FromAbi(const cgpt_material& m)
{
    Material out;
    out.albedo = V(m.albedo); out.specular = m.specular; out.refractivity = m.refractivity;
    out.absorption = V(m.absorption); out.ior = m.ior; out.emissive = V(m.emissive);
    out.intensity = m.intensity; out.is_light = m.is_light != 0;
    return out;
}
bool ValidOption(int o) { return o >= 0 && o < MeshBVH::BuildOption_NumOptions; }

// Nothing may unwind through the C ABI: a corrupt file or an allocation failure inside the C++ mirror (std::bad_alloc,
// std::length_error) becomes an error status and a message, never std::terminate in the host application.
template <class R, class F>
R Guarded(R on_error, F&& body) noexcept
{
    try {
        return body();
    } catch (const std::exception& e) {
        try { g_error = std::string("exception in the host mirror: ") + e.what(); } catch (...) {}
    } catch (...) {
        try { g_error = "unknown exception in the host mirror"; } catch (...) {}
    }
    return on_error;
}
}  // namespace

extern "C" {

const char* cgpth_last_error(void) { return g_error.c_str(); }

cgpth_mesh* cgpth_mesh_load_gltf(const char* path)
{
    return Guarded<cgpth_mesh*>(nullptr, [&]() -> cgpth_mesh* {
        if (!path) { Fail("null path"); return nullptr; }
        cgpth_mesh* m = new (std::nothrow) cgpth_mesh;
        if (!m) { Fail("out of memory"); return nullptr; }
        std::string err;
        if (!GLTFLoader::Load(path, m->mesh, err)) { Fail(err); delete m; return nullptr; }
        return m;
    });
}

cgpth_mesh* cgpth_mesh_from_arrays(const cgpt_vertex* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices)
{
    return Guarded<cgpth_mesh*>(nullptr, [&]() -> cgpth_mesh* {
        if ((!vertices && n_vertices) || (!indices && n_indices)) { Fail("null array"); return nullptr; }
        cgpth_mesh* m = new (std::nothrow) cgpth_mesh;
        if (!m) { Fail("out of memory"); return nullptr; }
        m->mesh.vertices.assign(vertices, vertices + n_vertices);
        m->mesh.indices.assign(indices, indices + n_indices);
        return m;
    });
}

cgpth_mesh* cgpth_mesh_dragon_standin(uint32_t level)
{
    return Guarded<cgpth_mesh*>(nullptr, [&]() -> cgpth_mesh* {
        if (level > 9) { Fail("icosphere level > 9 (5.2 M triangles) refused"); return nullptr; }
        cgpth_mesh* m = new (std::nothrow) cgpth_mesh;
        if (!m) { Fail("out of memory"); return nullptr; }
        m->mesh = MakeDragonStandIn(level);
        return m;
    });
}

cgpth_mesh* cgpth_mesh_bumpy_icosphere(uint32_t level, const float center[3], const float radii[3], float bump)
{
    return Guarded<cgpth_mesh*>(nullptr, [&]() -> cgpth_mesh* {
        if (level > 9 || !center || !radii) { Fail("bad icosphere arguments"); return nullptr; }
        cgpth_mesh* m = new (std::nothrow) cgpth_mesh;
        if (!m) { Fail("out of memory"); return nullptr; }
        m->mesh = MakeBumpyIcosphere(level, center, radii, bump);
        return m;
    });
}

int cgpth_mesh_save_gltf(const cgpth_mesh* mesh, const char* gltf_path)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!mesh || !gltf_path) return Fail("null argument");
        std::string err;
        if (!GLTFLoader::Save(gltf_path, mesh->mesh, err)) return Fail(err);
        return CGPT_OK;
    });
}

uint32_t cgpth_mesh_num_vertices(const cgpth_mesh* mesh) { return mesh ? (uint32_t)mesh->mesh.vertices.size() : 0; }
uint32_t cgpth_mesh_num_indices(const cgpth_mesh* mesh) { return mesh ? (uint32_t)mesh->mesh.indices.size() : 0; }
const cgpt_vertex* cgpth_mesh_vertices(const cgpth_mesh* mesh) { return mesh ? mesh->mesh.vertices.data() : nullptr; }
const uint32_t* cgpth_mesh_indices(const cgpth_mesh* mesh) { return mesh ? mesh->mesh.indices.data() : nullptr; }
void cgpth_mesh_free(cgpth_mesh* mesh) { delete mesh; }

cgpth_scene* cgpth_scene_new(void)
{
    return Guarded<cgpth_scene*>(nullptr, [&]() -> cgpth_scene* {
        cgpth_scene* s = new (std::nothrow) cgpth_scene;
        if (!s) Fail("out of memory");
        return s;
    });
}
void cgpth_scene_free(cgpth_scene* scene) { delete scene; }

cgpth_scene* cgpth_scene_reference_layout(const cgpth_mesh* mesh, uint32_t mesh_material, float aspect, int build_option)
{
    return Guarded<cgpth_scene*>(nullptr, [&]() -> cgpth_scene* {
        if (!mesh || !ValidOption(build_option) || mesh_material > 3) { Fail("bad argument to cgpth_scene_reference_layout"); return nullptr; }
        cgpth_scene* s = cgpth_scene_new();
        if (!s) return nullptr;
        s->scene = MakeReferenceScene(mesh->mesh, mesh_material, aspect, (MeshBVH::BuildOption)build_option);
        if (!s->scene.objects[0].valid) { Fail("mesh is empty or has out-of-range indices"); delete s; return nullptr; }
        return s;
    });
}

int cgpth_scene_add_material(cgpth_scene* scene, const cgpt_material* material)
{
    return Guarded<int>(-CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !material) return -Fail("null argument");
        scene->scene.materials.push_back(FromAbi(*material));
        return (int)scene->scene.materials.size() - 1;
    });
}

int cgpth_scene_set_material(cgpth_scene* scene, uint32_t index, const cgpt_material* material)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !material || index >= scene->scene.materials.size()) return Fail("bad material index");
        scene->scene.materials[index] = FromAbi(*material);
        return CGPT_OK;
    });
}

int cgpth_scene_add_mesh(cgpth_scene* scene, const cgpth_mesh* mesh, uint32_t mat_index, int build_option)
{
    return Guarded<int>(-CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !mesh || !ValidOption(build_option)) return -Fail("bad argument to cgpth_scene_add_mesh");
        scene->scene.objects.emplace_back("mesh", mesh->mesh, mat_index, (MeshBVH::BuildOption)build_option);
        if (!scene->scene.objects.back().valid) {
            scene->scene.objects.pop_back();
            return -Fail("mesh is empty or has out-of-range indices");
        }
        return (int)scene->scene.objects.size() - 1;
    });
}

// the device builder behind MeshBVH::BuildWith / RebuildWith: cgpt_bvh_build_ex on ctx; keeps the device's message
static MeshBVH::TreeBuilder DeviceBuilder(cgpt_ctx* ctx, std::string& device_error)
{
    return [ctx, &device_error](const cgpt_triangle* tris, uint32_t n, int option, const uint32_t* initial, cgpt_bvh_node* nodes, uint32_t* n_nodes,
                                uint32_t* tri_indices, uint32_t* depth) {
        float area = 0.0f;
        if (cgpt_bvh_build_ex(ctx, tris, n, (uint32_t)option, initial, nodes, n_nodes, tri_indices, depth, &area) == CGPT_OK) return true;
        device_error = cgpt_last_error(ctx);
        return false;
    };
}

int cgpth_scene_add_mesh_device_built_ex(cgpth_scene* scene, const cgpth_mesh* mesh, uint32_t mat_index, cgpt_ctx* ctx, int build_option)
{
    return Guarded<int>(-CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !mesh || !ctx || !ValidOption(build_option)) return -Fail("bad argument to cgpth_scene_add_mesh_device_built");
        std::string device_error;
        scene->scene.objects.emplace_back("mesh", mesh->mesh, mat_index, (MeshBVH::BuildOption)build_option, DeviceBuilder(ctx, device_error));
        if (!scene->scene.objects.back().valid) {
            scene->scene.objects.pop_back();
            return -Fail(device_error.empty() ? std::string("mesh is empty, has out-of-range indices, or the device returned a malformed tree")
                                              : "device BVH build failed: " + device_error);
        }
        return (int)scene->scene.objects.size() - 1;
    });
}

int cgpth_scene_add_mesh_device_built(cgpth_scene* scene, const cgpth_mesh* mesh, uint32_t mat_index, cgpt_ctx* ctx)
{
    return cgpth_scene_add_mesh_device_built_ex(scene, mesh, mat_index, ctx, (int)MeshBVH::BuildOption_SAHSplitIntervals);
}

int cgpth_scene_rebuild_bvh_device(cgpth_scene* scene, uint32_t obj_index, int build_option, cgpt_ctx* ctx)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !ctx || obj_index >= scene->scene.objects.size() || !scene->scene.objects[obj_index].has_bvh || !ValidOption(build_option))
            return Fail("bad argument to cgpth_scene_rebuild_bvh_device");
        std::string device_error;
        if (!scene->scene.objects[obj_index].bvh.RebuildWith((MeshBVH::BuildOption)build_option, DeviceBuilder(ctx, device_error)))
            return Fail(device_error.empty() ? std::string("the device returned a malformed tree (the BVH is unchanged)") : "device BVH rebuild failed: " + device_error);
        return CGPT_OK;
    });
}

int cgpth_scene_add_sphere(cgpth_scene* scene, const float center[3], float radius, uint32_t mat_index)
{
    return Guarded<int>(-CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !center) return -Fail("null argument");
        scene->scene.objects.emplace_back("sphere", Sphere{ V(center), radius }, mat_index);
        return (int)scene->scene.objects.size() - 1;
    });
}

int cgpth_scene_add_plane(cgpth_scene* scene, const float normal[3], const float point[3], uint32_t mat_index)
{
    return Guarded<int>(-CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !normal || !point) return -Fail("null argument");
        scene->scene.objects.emplace_back("plane", Plane{ V(normal), V(point) }, mat_index);
        return (int)scene->scene.objects.size() - 1;
    });
}

int cgpth_scene_add_light(cgpth_scene* scene, uint32_t obj_index)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || obj_index >= scene->scene.objects.size()) return Fail("bad object index");
        const Object& o = scene->scene.objects[obj_index];
        // ref: Main.cpp:371-384: only meshes and sphere primitives can be sampled, anything else EXCEPTs
        if (!o.has_bvh && o.kind != CGPT_OBJECT_SPHERE) { g_error = "only meshes and spheres can be light sources"; return CGPT_ERR_UNSUPPORTED; }
        scene->scene.light_source_indices.push_back(obj_index);
        return CGPT_OK;
    });
}

int cgpth_scene_set_camera(cgpth_scene* scene, const float pos[3], const float view_dir[3], float fov_deg, float aspect)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !pos || !view_dir) return Fail("null argument");
        scene->scene.camera = Camera(V(pos), V(view_dir), fov_deg, aspect);
        return CGPT_OK;
    });
}

int cgpth_scene_set_settings(cgpth_scene* scene, const cgpt_settings* s)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !s) return Fail("null argument");
        scene->scene.settings.max_ray_depth = s->max_ray_depth;
        scene->scene.settings.next_event_estimation_enabled = s->next_event_estimation_enabled != 0;
        scene->scene.settings.cosine_weighted_diffuse_reflection_enabled = s->cosine_weighted_diffuse_reflection_enabled != 0;
        scene->scene.settings.russian_roulette_enabled = s->russian_roulette_enabled != 0;
        scene->scene.render_mode = s->render_mode;
        scene->scene.debug_render_mode = s->debug_render_mode;
        return CGPT_OK;
    });
}

int cgpth_scene_rebuild_bvh(cgpth_scene* scene, uint32_t obj_index, int build_option)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || obj_index >= scene->scene.objects.size() || !scene->scene.objects[obj_index].has_bvh || !ValidOption(build_option))
            return Fail("bad argument to cgpth_scene_rebuild_bvh");
        scene->scene.objects[obj_index].bvh.Rebuild((MeshBVH::BuildOption)build_option);
        return CGPT_OK;
    });
}

int cgpth_scene_bvh_info(const cgpth_scene* scene, uint32_t obj_index, cgpth_bvh_info* out)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !out || obj_index >= scene->scene.objects.size() || !scene->scene.objects[obj_index].has_bvh) return Fail("not a mesh object");
        const MeshBVH& b = scene->scene.objects[obj_index].bvh;
        *out = cgpth_bvh_info{};
        out->num_triangles = b.NumTriangles(); out->nodes_used = b.NumNodes(); out->max_depth = b.GetMaxDepth(); out->total_area = b.GetTotalArea();
        for (uint32_t i = 0; i < b.NumNodes(); ++i) {
            const uint32_t c = b.Nodes()[i].prim_count;
            if (c > 0) { out->num_leaves++; if (c > out->max_leaf_size) out->max_leaf_size = c; }
        }
        return CGPT_OK;
    });
}

int cgpth_scene_bvh_export(const cgpth_scene* scene, uint32_t obj_index, cgpt_bvh_node* nodes, uint32_t* tri_indices)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !nodes || !tri_indices || obj_index >= scene->scene.objects.size() || !scene->scene.objects[obj_index].has_bvh) return Fail("not a mesh object");
        const MeshBVH& b = scene->scene.objects[obj_index].bvh;
        memcpy(nodes, b.Nodes(), sizeof(cgpt_bvh_node) * b.NumNodes());
        memcpy(tri_indices, b.TriIndices(), sizeof(uint32_t) * b.NumTriangles());
        return CGPT_OK;
    });
}

int cgpth_scene_flatten(cgpth_scene* scene, cgpt_scene_desc* out)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !out) return Fail("null argument");
        *out = scene->scene.Flatten(scene->flat);
        return CGPT_OK;
    });
}

int cgpth_scene_get_camera(const cgpth_scene* scene, cgpt_camera* out)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !out) return Fail("null argument");
        *out = scene->scene.camera.Abi();
        return CGPT_OK;
    });
}

int cgpth_scene_get_settings(const cgpth_scene* scene, cgpt_settings* out)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        if (!scene || !out) return Fail("null argument");
        *out = scene->scene.AbiSettings();
        return CGPT_OK;
    });
}

int cgpth_write_ppm(const char* path, const uint32_t* pixels, uint32_t width, uint32_t height)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        std::string err;
        if (!path || !pixels) return Fail("null argument");
        return WritePPM(path, pixels, width, height, err) ? CGPT_OK : Fail(err);
    });
}
int cgpth_write_pfm(const char* path, const float* acc, uint32_t n, uint32_t width, uint32_t height)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        std::string err;
        if (!path || !acc) return Fail("null argument");
        return WritePFM(path, acc, n, width, height, err) ? CGPT_OK : Fail(err);
    });
}
int cgpth_write_accumulator(const char* path, const float* acc, uint32_t n, uint32_t width, uint32_t height)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        std::string err;
        if (!path || !acc) return Fail("null argument");
        return WriteAccumulator(path, acc, n, width, height, err) ? CGPT_OK : Fail(err);
    });
}
int cgpth_read_accumulator(const char* path, float* acc, uint32_t* n, uint32_t width, uint32_t height)
{
    return Guarded<int>((int)CGPT_ERR_INVALID, [&]() -> int {
        std::string err;
        if (!path || !acc || !n) return Fail("null argument");
        return ReadAccumulator(path, acc, n, width, height, err) ? CGPT_OK : Fail(err);
    });
}

uint32_t cgpth_fast_div(uint32_t n, uint32_t d) { return d == 0u ? 0u : fast_div(n, MakeFastDiv(d)); }

}  // extern "C"
