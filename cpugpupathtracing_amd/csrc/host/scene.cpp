// scene.cpp -- Material/Camera/Scene helpers and the flatten step (see scene.h).
#include "scene.h"

#include <cstring>

namespace cgpt {

cgpt_material Material::ToAbi() const
{
    cgpt_material m{};
    m.albedo[0] = albedo.x; m.albedo[1] = albedo.y; m.albedo[2] = albedo.z;
    m.specular = specular;
    m.refractivity = refractivity;
    m.absorption[0] = absorption.x; m.absorption[1] = absorption.y; m.absorption[2] = absorption.z;
    m.ior = ior;
    m.emissive[0] = emissive.x; m.emissive[1] = emissive.y; m.emissive[2] = emissive.z;
    m.intensity = intensity;
    m.is_light = is_light ? 1u : 0u;
    return m;
}

void Camera::UpdateScreenPlane()
{
    // same arithmetic as cgpt_camera_from_view (device library) with the already-converted fov
    Vec3 center = pos_ + fov_ * view_dir_;                                   // ref: Main.cpp:145 (distance = fov in radians)
    Vec3 tl = center + Vec3(-aspect_, 1.0f, 0.0f);
    Vec3 tr = center + Vec3(aspect_, 1.0f, 0.0f);
    Vec3 bl = center + Vec3(-aspect_, -1.0f, 0.0f);
    abi_.pos[0] = pos_.x; abi_.pos[1] = pos_.y; abi_.pos[2] = pos_.z;
    abi_.top_left[0] = tl.x; abi_.top_left[1] = tl.y; abi_.top_left[2] = tl.z;
    abi_.top_right[0] = tr.x; abi_.top_right[1] = tr.y; abi_.top_right[2] = tr.z;
    abi_.bottom_left[0] = bl.x; abi_.bottom_left[1] = bl.y; abi_.bottom_left[2] = bl.z;
}

bool Camera::Move(float right, float up, float forward)
{
    pos_.x -= right;                                                         // ref: Main.cpp:116-118
    pos_.y += up;
    pos_.z -= forward;
    const bool changed = right != 0.0f || up != 0.0f || forward != 0.0f;
    if (changed) UpdateScreenPlane();
    return changed;
}

cgpt_scene_desc Scene::Flatten(FlatStorage& st) const
{
    st = FlatStorage{};
    for (const Object& o : objects) {
        cgpt_object d{};
        d.mat_index = o.mat_index;
        if (o.has_bvh) {
            d.kind = CGPT_OBJECT_MESH;
            d.node_offset = (uint32_t)st.nodes.size();
            d.node_count = o.bvh.NumNodes();
            d.tri_offset = (uint32_t)st.triangles.size();
            d.tri_count = o.bvh.NumTriangles();
            d.max_depth = o.bvh.GetMaxDepth();
            d.total_area = o.bvh.GetTotalArea();
            st.nodes.insert(st.nodes.end(), o.bvh.Nodes(), o.bvh.Nodes() + d.node_count);
            st.triangles.insert(st.triangles.end(), o.bvh.Triangles(), o.bvh.Triangles() + d.tri_count);
            st.tri_indices.insert(st.tri_indices.end(), o.bvh.TriIndices(), o.bvh.TriIndices() + d.tri_count);
        } else if (o.kind == CGPT_OBJECT_SPHERE) {
            d.kind = CGPT_OBJECT_SPHERE;
            d.sphere_center[0] = o.sphere.center.x; d.sphere_center[1] = o.sphere.center.y; d.sphere_center[2] = o.sphere.center.z;
            d.sphere_radius = o.sphere.radius;
        } else {
            d.kind = CGPT_OBJECT_PLANE;
            d.plane_normal[0] = o.plane.normal.x; d.plane_normal[1] = o.plane.normal.y; d.plane_normal[2] = o.plane.normal.z;
            d.plane_point[0] = o.plane.point.x; d.plane_point[1] = o.plane.point.y; d.plane_point[2] = o.plane.point.z;
        }
        st.objects.push_back(d);
    }
    for (const Material& m : materials) st.materials.push_back(m.ToAbi());
    st.lights = light_source_indices;

    cgpt_scene_desc desc{};
    desc.objects = st.objects.data(); desc.n_objects = (uint32_t)st.objects.size();
    desc.nodes = st.nodes.data(); desc.n_nodes = (uint32_t)st.nodes.size();
    desc.triangles = st.triangles.data(); desc.n_triangles = (uint32_t)st.triangles.size();
    desc.tri_indices = st.tri_indices.data();
    desc.materials = st.materials.data(); desc.n_materials = (uint32_t)st.materials.size();
    desc.light_indices = st.lights.data(); desc.n_lights = (uint32_t)st.lights.size();
    return desc;
}

cgpt_settings Scene::AbiSettings() const
{
    cgpt_settings s{};
    s.max_ray_depth = settings.max_ray_depth;
    s.next_event_estimation_enabled = settings.next_event_estimation_enabled;
    s.cosine_weighted_diffuse_reflection_enabled = settings.cosine_weighted_diffuse_reflection_enabled;
    s.russian_roulette_enabled = settings.russian_roulette_enabled;
    s.render_mode = render_mode;
    s.debug_render_mode = debug_render_mode;
    return s;
}

Scene MakeReferenceScene(const Mesh& dragon_mesh, uint32_t mesh_material, float aspect, MeshBVH::BuildOption option)
{
    Scene scene;
    scene.camera = Camera(Vec3(0.0f, 0.0f, 8.0f), Vec3(0.0f, 0.0f, -1.0f), 60.0f, aspect);   // ref: Main.cpp:777

    scene.materials.emplace_back(Vec3(0.2f, 0.2f, 0.8f), 0.0f);                                 // ref: Main.cpp:779-782
    scene.materials.emplace_back(Vec3(1.0f), 0.0f);
    scene.materials.emplace_back(Vec3(1.0f, 0.95f, 0.8f), 10.0f, true);
    scene.materials.emplace_back(Vec3(1.0f), 0.0f, 1.0f, Vec3(0.2f, 0.8f, 0.8f), 1.517f);

    scene.objects.emplace_back("Dragon", dragon_mesh, mesh_material, option);                   // ref: Main.cpp:787

    Mesh ground;                                                                                // ref: Main.cpp:789-800
    ground.indices = { 0, 1, 2, 2, 3, 0 };
    ground.vertices.push_back({ { -1000.0f, -3.0f, 1000.0f }, { 0.0f, 1.0f, 0.0f } });
    ground.vertices.push_back({ { -1000.0f, -3.0f, -1000.0f }, { 0.0f, 1.0f, 0.0f } });
    ground.vertices.push_back({ { 1000.0f, -3.0f, -1000.0f }, { 0.0f, 1.0f, 0.0f } });
    ground.vertices.push_back({ { 1000.0f, -3.0f, 1000.0f }, { 0.0f, 1.0f, 0.0f } });
    scene.objects.emplace_back("Ground", ground, 1, MeshBVH::BuildOption_SAHSplitIntervals);

    scene.objects.emplace_back("Spherical light0", Sphere{ Vec3(10.0f, 10.0f, 10.0f), 5.0f }, 2);   // ref: Main.cpp:816-819
    scene.light_source_indices.push_back((uint32_t)scene.objects.size() - 1);
    scene.objects.emplace_back("Spherical light1", Sphere{ Vec3(-10.0f, 10.0f, -10.0f), 5.0f }, 2);
    scene.light_source_indices.push_back((uint32_t)scene.objects.size() - 1);

    scene.render_mode = CGPT_MODE_ADVANCED;
    return scene;
}

}  // namespace cgpt
