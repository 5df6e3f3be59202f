// mesh_gen.cpp -- see mesh_gen.h.
#include "mesh_gen.h"

#include <cmath>
#include <unordered_map>

namespace cgpt {

namespace {
struct D3 { double x, y, z; };
inline D3 Normalized(D3 v) { double l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); return { v.x / l, v.y / l, v.z / l }; }
}  // namespace

Mesh MakeBumpyIcosphere(uint32_t level, const float center[3], const float radii[3], float bump)
{
    const double t = (1.0 + std::sqrt(5.0)) / 2.0;
    std::vector<D3> dirs = {
        { -1, t, 0 }, { 1, t, 0 }, { -1, -t, 0 }, { 1, -t, 0 }, { 0, -1, t }, { 0, 1, t },
        { 0, -1, -t }, { 0, 1, -t }, { t, 0, -1 }, { t, 0, 1 }, { -t, 0, -1 }, { -t, 0, 1 } };
    for (D3& d : dirs) d = Normalized(d);
    std::vector<uint32_t> idx = {
        0, 11, 5, 0, 5, 1, 0, 1, 7, 0, 7, 10, 0, 10, 11, 1, 5, 9, 5, 11, 4, 11, 10, 2, 10, 7, 6, 7, 1, 8,
        3, 9, 4, 3, 4, 2, 3, 2, 6, 3, 6, 8, 3, 8, 9, 4, 9, 5, 2, 4, 11, 6, 2, 10, 8, 6, 7, 9, 8, 1 };

    for (uint32_t l = 0; l < level; ++l) {
        std::unordered_map<uint64_t, uint32_t> midpoint;
        midpoint.reserve(idx.size());
        auto mid = [&](uint32_t a, uint32_t b) -> uint32_t {
            const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
            auto it = midpoint.find(key);
            if (it != midpoint.end()) return it->second;
            D3 m = Normalized({ dirs[a].x + dirs[b].x, dirs[a].y + dirs[b].y, dirs[a].z + dirs[b].z });
            dirs.push_back(m);
            const uint32_t id = (uint32_t)dirs.size() - 1;
            midpoint.emplace(key, id);
            return id;
        };
        std::vector<uint32_t> next;
        next.reserve(idx.size() * 4);
        for (size_t k = 0; k < idx.size(); k += 3) {
            const uint32_t a = idx[k], b = idx[k + 1], c = idx[k + 2];
            const uint32_t ab = mid(a, b), bc = mid(b, c), ca = mid(c, a);
            const uint32_t tri[12] = { a, ab, ca, b, bc, ab, c, ca, bc, ab, bc, ca };
            next.insert(next.end(), tri, tri + 12);
        }
        idx.swap(next);
    }

    Mesh mesh;
    mesh.indices = std::move(idx);
    mesh.vertices.resize(dirs.size());
    for (size_t v = 0; v < dirs.size(); ++v) {
        const D3 n = dirs[v];
        const double r = 1.0 + (double)bump * std::sin(7.0 * n.x + 1.0) * std::sin(5.0 * n.y + 2.0) * std::sin(6.0 * n.z + 3.0);
        cgpt_vertex& out = mesh.vertices[v];
        out.pos[0] = (float)(center[0] + radii[0] * r * n.x);
        out.pos[1] = (float)(center[1] + radii[1] * r * n.y);
        out.pos[2] = (float)(center[2] + radii[2] * r * n.z);
        out.normal[0] = (float)n.x; out.normal[1] = (float)n.y; out.normal[2] = (float)n.z;
    }
    return mesh;
}

Mesh MakeDragonStandIn(uint32_t level)
{
    // dragon AABB x[-7.05,7.05] y[-3.15,3.15] z[-9.94,0]; the bump (15 %) is taken out of the radii so the
    // displaced surface stays inside the box
    const float bump = 0.15f;
    const float center[3] = { 0.0f, 0.0f, -4.97f };
    const float radii[3] = { 7.05f / (1.0f + bump), 3.15f / (1.0f + bump), 4.97f / (1.0f + bump) };
    return MakeBumpyIcosphere(level, center, radii, bump);
}

}  // namespace cgpt
