// scene.h -- host-side mirror of the reference's scene structs, feeding the C ABI.
//
// Same names and meaning as the reference's Material / Camera / Object / Data::Settings
// (ref: Source/Main.cpp:51-69, 94-170, 228-235, 245-275) minus their ImGui methods; `Scene` is the part
// of the file-static `data` (ref: Main.cpp:200-236) that Render() reads.  Scene::Flatten() produces the
// POD cgpt_scene_desc the device library uploads.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "cpugpupt_abi.h"
#include "mesh_bvh.h"
#include "vec.h"

namespace cgpt {

struct Material {  // ref: Main.cpp:51-69
    Vec3 albedo{ 0.0f };
    float specular = 0.0f;
    float refractivity = 0.0f;
    Vec3 absorption{ 0.0f };
    float ior = 1.0f;
    Vec3 emissive{ 0.0f };
    float intensity = 0.0f;
    bool is_light = false;

    Material() = default;
    Material(const Vec3& albedo_, float spec) : albedo(albedo_), specular(spec) {}
    Material(const Vec3& albedo_, float spec, float refract, const Vec3& absorption_, float ior_)
        : albedo(albedo_), specular(spec), refractivity(refract), absorption(absorption_), ior(ior_) {}
    Material(const Vec3& emissive_, float intensity_, bool light) : emissive(emissive_), intensity(intensity_), is_light(light) {}

    cgpt_material ToAbi() const;
};

struct Sphere {  // ref: Include/Primitives.h:36-44
    Vec3 center{ 0.0f };
    float radius = 0.0f;
};
struct Plane {  // ref: Include/Primitives.h:30-34
    Vec3 normal{ 0.0f };
    Vec3 point{ 0.0f };
};

class Camera {  // ref: Main.cpp:94-170 (GetRay runs on the device; Update()'s input half is out of scope)
public:
    Camera() { UpdateScreenPlane(); }
    Camera(const Vec3& pos, const Vec3& view_dir, float fov_deg, float aspect)
        : pos_(pos), view_dir_(view_dir), fov_(fov_deg * kPi / 180.0f), aspect_(aspect) { UpdateScreenPlane(); }
    // translate like the WASD handler (ref: Main.cpp:116-118); returns true if the view changed
    bool Move(float right, float up, float forward);
    const cgpt_camera& Abi() const { return abi_; }

private:
    void UpdateScreenPlane();  // ref: Main.cpp:143-149
    Vec3 pos_{ 0.0f };
    Vec3 view_dir_{ 0.0f, 0.0f, -1.0f };
    float fov_ = 60.0f * kPi / 180.0f;
    float aspect_ = 16.0f / 9.0f;
    cgpt_camera abi_{};
};

struct Object {  // ref: Main.cpp:245-275
    Object(const char* name_, const Mesh& mesh, uint32_t mat, MeshBVH::BuildOption option)
        : name(name_), mat_index(mat), has_bvh(true) { valid = bvh.Build(mesh.vertices, mesh.indices, option); }
    Object(const char* name_, const Mesh& mesh, uint32_t mat, MeshBVH::BuildOption build_option, const MeshBVH::TreeBuilder& builder)
        : name(name_), mat_index(mat), has_bvh(true) { valid = bvh.BuildWith(mesh.vertices, mesh.indices, build_option, builder); }
    Object(const char* name_, const Sphere& s, uint32_t mat) : name(name_), mat_index(mat), kind(CGPT_OBJECT_SPHERE), sphere(s) {}
    Object(const char* name_, const Plane& p, uint32_t mat) : name(name_), mat_index(mat), kind(CGPT_OBJECT_PLANE), plane(p) {}

    std::string name;
    uint32_t mat_index = 0;
    bool has_bvh = false;
    bool valid = true;
    MeshBVH bvh;
    uint32_t kind = CGPT_OBJECT_MESH;
    Sphere sphere;
    Plane plane;
};

struct Settings {  // ref: Main.cpp:228-235
    int32_t max_ray_depth = 5;
    bool next_event_estimation_enabled = true;
    bool cosine_weighted_diffuse_reflection_enabled = true;
    bool russian_roulette_enabled = true;
};

struct Scene {  // ref: Main.cpp:209-216,228-235
    std::vector<Object> objects;
    std::vector<uint32_t> light_source_indices;
    std::vector<Material> materials;
    Camera camera;
    Settings settings;
    uint32_t render_mode = CGPT_MODE_COMPARISON;       // the reference's default (ref: Main.cpp:215)
    uint32_t debug_render_mode = CGPT_DEBUG_NONE;

    // Packs objects/BVHs/materials/lights into the POD arrays of cgpt_scene_desc.  The returned desc points
    // into `storage`, which must outlive its use.
    struct FlatStorage {
        std::vector<cgpt_object> objects;
        std::vector<cgpt_bvh_node> nodes;
        std::vector<cgpt_triangle> triangles;
        std::vector<uint32_t> tri_indices;
        std::vector<cgpt_material> materials;
        std::vector<uint32_t> lights;
    };
    cgpt_scene_desc Flatten(FlatStorage& storage) const;
    cgpt_settings AbiSettings() const;
};

// The shipped scene layout (ref: Main.cpp:777-819): camera (0,0,8)->-z fov 60, the 4 materials, the given mesh as
// "Dragon" with `mesh_material`, the ground quad y=-3, two sphere lights r=5.
Scene MakeReferenceScene(const Mesh& dragon_mesh, uint32_t mesh_material, float aspect, MeshBVH::BuildOption option);

}  // namespace cgpt
