// cgpt_abi.hip -- implementation of the C ABI in include/cpugpupt_abi.h on HIP (gfx950).
// Context management, the AoS -> device-layout upload (device_scene.h), kernel launches, statistics.
// There is no CPU fallback anywhere in this file: without a gfx950 device every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "cpugpupt_abi.h"
#include "device_scene.h"

namespace cgpt {
hipError_t LaunchMegakernel(const DevRenderArgs& args, bool count, hipStream_t stream);                    // path_kernels.hip
hipError_t LaunchIntersectRays(const DevScene& sc, const float* origins, const float* dirs, const float* tmax, uint32_t n, float* out_t,
                               uint32_t* out_obj, uint32_t* out_tri, uint32_t* out_depth, DevCounters* counters, hipStream_t stream);
int LaunchWavefront(struct ::cgpt_ctx* ctx, const DevRenderArgs& args, bool count);                       // wavefront_kernels.hip
void WavefrontFree(void* state);
void WavefrontCollectTiming(void* state, double* trace_ms, uint32_t* trace_launches, double* round0_ms, uint32_t* round0_launches);
int WavefrontSetTuning(struct ::cgpt_ctx* ctx, const char* name, uint32_t value);
uint32_t WavefrontTraceWavesPerSimd(void* state);
int LaunchPersistent(struct ::cgpt_ctx* ctx, const DevRenderArgs& args, bool count);                          // persistent_kernel.hip
void PersistentFree(void* state);
void PersistentCollectTiming(void* state, double* ms, uint32_t* launches, uint32_t* waves_per_simd);
int PersistentSetTuning(struct ::cgpt_ctx* ctx, const char* name, uint32_t value, bool* known);
uint32_t MegakernelWavesPerSimd(const DevRenderArgs& args);                                                   // path_kernels.hip
}  // namespace cgpt

using namespace cgpt;

namespace {
std::string g_create_error = "";
}

#include "ctx_internal.h"

namespace cgpt {
// accessors for the other translation units
hipStream_t CtxStream(cgpt_ctx* ctx) { return ctx->stream; }
int CtxDevice(cgpt_ctx* ctx) { return ctx->device; }
void** CtxWavefrontSlot(cgpt_ctx* ctx) { return &ctx->wavefront_state; }
void** CtxPersistentSlot(cgpt_ctx* ctx) { return &ctx->persistent_state; }
// the launchers of the multi-launch kernels record the render's start event themselves, after their one-time host setup
// (allocations, occupancy queries), so that cgpt_stats.kernel_ms of a first call is device time
hipEvent_t CtxStartEvent(cgpt_ctx* ctx) { return ctx->ev_start; }
int CreateFail(int code, const char* fmt, ...)                               // failure of cgpt_ctx_create: there is no context to hold the text
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_create_error = buf;
    return code;
}
int CtxFail(cgpt_ctx* ctx, int code, const char* fmt, ...)
{
    if (!ctx) return code;
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    ctx->error = buf;
    return code;
}
}  // namespace cgpt

namespace {

int Fail(cgpt_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (ctx) ctx->error = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                        \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return Fail((ctx), CGPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

// Calls into the multi-device context (multi_gpu.hip: host vectors, worker threads): nothing may unwind through the C ABI
#define GROUP_CALL(ctx, expr)                                                                                     \
    do {                                                                                                          \
        try { return (expr); }                                                                                    \
        catch (const std::exception& e_) { return Fail((ctx), CGPT_ERR_INVALID, "%s: %s", __func__, e_.what()); } \
        catch (...) { return Fail((ctx), CGPT_ERR_INVALID, "%s: unknown exception", __func__); }                  \
    } while (0)

template <typename T>
int UploadArray(cgpt_ctx* ctx, T** dst, const std::vector<T>& src)
{
    if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
    const size_t bytes = sizeof(T) * (src.empty() ? 1 : src.size());
    HIP_TRY(ctx, hipMalloc((void**)dst, bytes));
    if (!src.empty()) HIP_TRY(ctx, hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    return CGPT_OK;
}

void FreeScene(cgpt_ctx* ctx)
{
    (void)hipFree(ctx->d_node_pairs); (void)hipFree(ctx->d_tri_leaf); (void)hipFree(ctx->d_tri_orig); (void)hipFree(ctx->d_tri_normal);
    (void)hipFree(ctx->d_materials); (void)hipFree(ctx->d_objects); (void)hipFree(ctx->d_obj_trace); (void)hipFree(ctx->d_lights);
    ctx->d_node_pairs = ctx->d_tri_leaf = ctx->d_tri_orig = ctx->d_tri_normal = ctx->d_materials = nullptr;
    ctx->d_objects = nullptr; ctx->d_obj_trace = nullptr; ctx->d_lights = nullptr;
    ctx->has_scene = false;
}

void FreeFramebuffer(cgpt_ctx* ctx)
{
    (void)hipFree(ctx->d_accumulator);
    (void)hipFree(ctx->d_pixels);
    ctx->d_accumulator = nullptr; ctx->d_pixels = nullptr;
}

float4 F4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
float AsFloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

void PackMaterial(const cgpt_material& m, float4 out[4])
{
    out[0] = F4(m.albedo[0], m.albedo[1], m.albedo[2], m.specular);
    out[1] = F4(m.refractivity, m.absorption[0], m.absorption[1], m.absorption[2]);
    out[2] = F4(m.ior, m.emissive[0], m.emissive[1], m.emissive[2]);
    out[3] = F4(m.intensity, AsFloat(m.is_light ? 1u : 0u), 0.0f, 0.0f);
}

// Re-lays the reference's AoS scene into the device layout of device_scene.h, validating everything a kernel will
// index with (a malformed tree must fail here, not fault on the GPU).
int BuildDeviceScene(cgpt_ctx* ctx, const cgpt_scene_desc& sd)
{
    if (sd.n_objects == 0 || !sd.objects) return Fail(ctx, CGPT_ERR_INVALID, "scene has no objects");
    if (sd.n_materials == 0 || !sd.materials) return Fail(ctx, CGPT_ERR_INVALID, "scene has no materials");
    if (sd.n_lights && !sd.light_indices) return Fail(ctx, CGPT_ERR_INVALID, "light_indices is null");

    std::vector<float4> pairs, tri_leaf, tri_orig, tri_normal, mats;
    std::vector<DevObject> objs(sd.n_objects);
    uint32_t max_tree_depth = 0;

    // leaf-record order (device_scene.h): the triangles of the small meshes first, then the rest in object order
    std::vector<uint32_t> leaf_base_of(sd.n_objects, 0);
    uint32_t n_small_tris = 0;
    {
        uint64_t total = 0;
        std::vector<uint8_t> small(sd.n_objects, 0);
        for (uint32_t oi = 0; oi < sd.n_objects; ++oi) {
            const cgpt_object& o = sd.objects[oi];
            if (o.kind != CGPT_OBJECT_MESH) continue;
            total += o.tri_count;
            if (o.tri_count > 0 && o.tri_count <= kSmallMeshTris && n_small_tris + o.tri_count <= kLdsTrisMax) { small[oi] = 1; leaf_base_of[oi] = n_small_tris; n_small_tris += o.tri_count; }
        }
        if (total >= (1u << 26)) return Fail(ctx, CGPT_ERR_INVALID, "scene too large: more than 2^26 triangles or inner nodes");
        uint32_t next = n_small_tris;
        for (uint32_t oi = 0; oi < sd.n_objects; ++oi)
            if (sd.objects[oi].kind == CGPT_OBJECT_MESH && !small[oi]) { leaf_base_of[oi] = next; next += sd.objects[oi].tri_count; }
        tri_leaf.resize(3 * (size_t)total);
    }

    for (uint32_t oi = 0; oi < sd.n_objects; ++oi) {
        const cgpt_object& o = sd.objects[oi];
        DevObject& d = objs[oi];
        memset(&d, 0, sizeof(d));
        d.kind = o.kind; d.mat_index = o.mat_index;
        if (o.mat_index >= sd.n_materials) return Fail(ctx, CGPT_ERR_INVALID, "object %u: mat_index %u out of range", oi, o.mat_index);
        if (o.kind == CGPT_OBJECT_SPHERE) {
            memcpy(d.sphere_center, o.sphere_center, 12);
            d.sphere_radius = o.sphere_radius;
            d.sphere_radius_sq = o.sphere_radius * o.sphere_radius;                  // Sphere ctor, ref: Primitives.h:38-39
            continue;
        }
        if (o.kind == CGPT_OBJECT_PLANE) {
            memcpy(d.plane_normal, o.plane_normal, 12);
            memcpy(d.plane_point, o.plane_point, 12);
            continue;
        }
        if (o.kind != CGPT_OBJECT_MESH)
            return Fail(ctx, CGPT_ERR_UNSUPPORTED, "object %u: primitive kind %u has no intersector (the reference EXCEPTs too, Primitives.cpp:304)", oi, o.kind);

        // ---- mesh: validate the slices ----
        if (!sd.nodes || !sd.triangles || !sd.tri_indices) return Fail(ctx, CGPT_ERR_INVALID, "mesh object %u but nodes/triangles/tri_indices is null", oi);
        if (o.node_count == 0 || (uint64_t)o.node_offset + o.node_count > sd.n_nodes) return Fail(ctx, CGPT_ERR_INVALID, "object %u: node slice out of range", oi);
        if (o.tri_count == 0 || (uint64_t)o.tri_offset + o.tri_count > sd.n_triangles) return Fail(ctx, CGPT_ERR_INVALID, "object %u: triangle slice out of range", oi);
        if ((o.node_count & 1u) == 0) return Fail(ctx, CGPT_ERR_INVALID, "object %u: a binary BVH has an odd node count, got %u", oi, o.node_count);
        const cgpt_bvh_node* nodes = sd.nodes + o.node_offset;
        const cgpt_triangle* tris = sd.triangles + o.tri_offset;
        const uint32_t* tidx = sd.tri_indices + o.tri_offset;

        const uint32_t pair_base = (uint32_t)(pairs.size() / 4);
        const uint32_t leaf_base = leaf_base_of[oi];
        const uint32_t orig_base = (uint32_t)(tri_orig.size() / 3);
        // record byte offsets are computed in 32 bits on the device (64-byte pairs, 48-byte leaf triangles)
        if ((uint64_t)leaf_base + o.tri_count >= (1u << 26) || (uint64_t)pair_base + o.node_count / 2 >= (1u << 26))
            return Fail(ctx, CGPT_ERR_INVALID, "scene too large: more than 2^26 triangles or inner nodes");

        auto code_of = [&](uint32_t node_index, uint32_t& code) -> bool {
            const cgpt_bvh_node& n = nodes[node_index];
            if (n.prim_count > 0) {
                if ((uint64_t)n.left_first + n.prim_count > o.tri_count) return false;
                code = kLeafBit | (leaf_base + n.left_first);
                return true;
            }
            // children were allocated as a pair after the parent: odd index, both in range, both beyond the parent
            if ((n.left_first & 1u) == 0 || n.left_first <= node_index || (uint64_t)n.left_first + 1 >= o.node_count) return false;
            code = pair_base + (n.left_first - 1) / 2;
            return true;
        };

        uint32_t root_code;
        if (!code_of(0, root_code)) return Fail(ctx, CGPT_ERR_INVALID, "object %u: malformed BVH root", oi);
        d.root_code = root_code; d.tri_base = orig_base; d.n_tris = o.tri_count; d.total_area = o.total_area;

        // leaf-ordered triangle records
        float4* leaf = tri_leaf.data() + 3 * (size_t)leaf_base;            // sized above
        for (uint32_t i = 0; i < o.tri_count; ++i) {
            const uint32_t t = tidx[i];
            if (t >= o.tri_count) return Fail(ctx, CGPT_ERR_INVALID, "object %u: tri_indices[%u] = %u out of range", oi, i, t);
            const cgpt_triangle& tr = tris[t];
            const float e1[3] = { tr.v1.pos[0] - tr.v0.pos[0], tr.v1.pos[1] - tr.v0.pos[1], tr.v1.pos[2] - tr.v0.pos[2] };   // ref: Primitives.cpp:9
            const float e2[3] = { tr.v2.pos[0] - tr.v0.pos[0], tr.v2.pos[1] - tr.v0.pos[1], tr.v2.pos[2] - tr.v0.pos[2] };   // ref: Primitives.cpp:10
            leaf[3 * (size_t)i + 0] = F4(tr.v0.pos[0], tr.v0.pos[1], tr.v0.pos[2], e1[0]);
            leaf[3 * (size_t)i + 1] = F4(e1[1], e1[2], e2[0], e2[1]);
            leaf[3 * (size_t)i + 2] = F4(0.0f, e2[2], AsFloat(t), AsFloat(0u));
        }
        // original-order records for GetTriangle users
        tri_orig.resize(tri_orig.size() + 3 * (size_t)o.tri_count);
        float4* orig = tri_orig.data() + 3 * (size_t)orig_base;
        for (uint32_t t = 0; t < o.tri_count; ++t) {
            const cgpt_triangle& tr = tris[t];
            orig[3 * (size_t)t + 0] = F4(tr.v0.pos[0], tr.v0.pos[1], tr.v0.pos[2], tr.v0.normal[0]);
            orig[3 * (size_t)t + 1] = F4(tr.v1.pos[0], tr.v1.pos[1], tr.v1.pos[2], tr.v0.normal[1]);
            orig[3 * (size_t)t + 2] = F4(tr.v2.pos[0], tr.v2.pos[1], tr.v2.pos[2], tr.v0.normal[2]);
            tri_normal.push_back(F4(tr.v0.normal[0], tr.v0.normal[1], tr.v0.normal[2], 0.0f));   // TriangleNormal, ref: Primitives.cpp:148-151
        }

        // child-pair records + leaf terminators; iterative DFS from the root also measures the real depth
        pairs.resize(pairs.size() + 4 * (size_t)(o.node_count / 2), F4(0, 0, 0, 0));
        float4* pr = pairs.data() + 4 * (size_t)pair_base;
        std::vector<uint8_t> covered(o.tri_count, 0);
        struct Item { uint32_t node, depth; };
        std::vector<Item> todo;
        todo.push_back({ 0, 0 });
        uint32_t visited = 0;
        while (!todo.empty()) {
            const Item it = todo.back(); todo.pop_back();
            if (++visited > o.node_count) return Fail(ctx, CGPT_ERR_INVALID, "object %u: BVH is not a tree", oi);
            if (it.depth > max_tree_depth) max_tree_depth = it.depth;
            const cgpt_bvh_node& n = nodes[it.node];
            if (n.prim_count > 0) {
                if ((uint64_t)n.left_first + n.prim_count > o.tri_count) return Fail(ctx, CGPT_ERR_INVALID, "object %u: leaf %u out of range", oi, it.node);
                for (uint32_t i = n.left_first; i < n.left_first + n.prim_count; ++i) {
                    if (covered[i]) return Fail(ctx, CGPT_ERR_INVALID, "object %u: triangle slot %u is in two leaves", oi, i);
                    covered[i] = 1;
                }
                leaf[3 * (size_t)(n.left_first + n.prim_count - 1) + 2].w = AsFloat(1u);     // last_in_leaf
                continue;
            }
            uint32_t lc, rc, dummy;
            if (!code_of(it.node, dummy)) return Fail(ctx, CGPT_ERR_INVALID, "object %u: malformed inner node %u", oi, it.node);
            const uint32_t L = n.left_first;
            if (!code_of(L, lc) || !code_of(L + 1, rc)) return Fail(ctx, CGPT_ERR_INVALID, "object %u: malformed children of node %u", oi, it.node);
            float4* rec = pr + 4 * (size_t)((L - 1) / 2);
            const cgpt_bvh_node& l = nodes[L]; const cgpt_bvh_node& r = nodes[L + 1];
            // left / right interleaved per component: one packed-f32 instruction handles both children (device_scene.h)
            rec[0] = F4(l.aabb_min[0], r.aabb_min[0], l.aabb_min[1], r.aabb_min[1]);
            rec[1] = F4(l.aabb_min[2], r.aabb_min[2], l.aabb_max[0], r.aabb_max[0]);
            rec[2] = F4(l.aabb_max[1], r.aabb_max[1], l.aabb_max[2], r.aabb_max[2]);
            rec[3] = F4(0.0f, 0.0f, AsFloat(lc), AsFloat(rc));
            todo.push_back({ L + 1, it.depth + 1 });
            todo.push_back({ L, it.depth + 1 });
        }
    }

    for (uint32_t i = 0; i < sd.n_lights; ++i) {
        const uint32_t li = sd.light_indices[i];
        if (li >= sd.n_objects) return Fail(ctx, CGPT_ERR_INVALID, "light_indices[%u] = %u out of range", i, li);
        if (sd.objects[li].kind != CGPT_OBJECT_MESH && sd.objects[li].kind != CGPT_OBJECT_SPHERE)
            return Fail(ctx, CGPT_ERR_UNSUPPORTED, "light %u: only mesh and sphere lights can be sampled (the reference EXCEPTs, Main.cpp:383)", i);
    }

    const uint32_t stack_depth = max_tree_depth + 1;
    if (stack_depth > 64) return Fail(ctx, CGPT_ERR_UNSUPPORTED, "BVH depth %u exceeds the traversal stack of 64 (ref: BVH.cpp:66)", max_tree_depth);

    mats.resize(4 * (size_t)sd.n_materials);
    for (uint32_t i = 0; i < sd.n_materials; ++i) PackMaterial(sd.materials[i], mats.data() + 4 * (size_t)i);
    std::vector<uint32_t> lights(sd.light_indices, sd.light_indices + sd.n_lights);

    // ---- record order (device_scene.h: "record order") ----
    // The reference allocates nodes depth-first; a record's index is only a name here (codes are rewritten), so the records are
    // renumbered: the first kTopRecords in breadth-first order over all meshes (the top of every tree, which every ray walks:
    // the trace kernel mirrors records [0, n_top_records) in LDS), the rest in the reference's order (CGPT_NODE_ORDER=bfs:
    // everything breadth-first; =dfs: nothing renumbered, for experiments).
    const uint32_t n_records = (uint32_t)(pairs.size() / 4);
    uint32_t n_top_records = 0;
    {
        const char* mode_env = getenv("CGPT_NODE_ORDER");
        const std::string mode = mode_env ? mode_env : "top";
        std::vector<uint32_t> bfs; bfs.reserve(n_records);
        for (uint32_t oi = 0; oi < sd.n_objects; ++oi)
            if (objs[oi].kind == CGPT_OBJECT_MESH && (objs[oi].root_code & kLeafBit) == 0u) bfs.push_back(objs[oi].root_code);
        const size_t bfs_limit = mode == "bfs" ? n_records : std::min<size_t>(n_records, kTopRecords);
        for (size_t head = 0; head < bfs.size() && bfs.size() < n_records; ++head) {
            if (mode != "bfs" && bfs.size() >= bfs_limit + 2 * kTopRecords) break;   // enough: only the first bfs_limit are used
            const float4& cc = pairs[4 * (size_t)bfs[head] + 3];
            uint32_t lc, rc; memcpy(&lc, &cc.z, 4); memcpy(&rc, &cc.w, 4);
            if ((lc & kLeafBit) == 0u) bfs.push_back(lc);
            if ((rc & kLeafBit) == 0u) bfs.push_back(rc);
        }
        if (mode != "dfs" && n_records > 0) {
            std::vector<uint32_t> perm(n_records, 0xFFFFFFFFu);
            uint32_t next = 0;
            for (size_t i = 0; i < bfs.size() && i < bfs_limit; ++i) perm[bfs[i]] = next++;
            for (uint32_t r = 0; r < n_records; ++r) if (perm[r] == 0xFFFFFFFFu) perm[r] = next++;
            std::vector<float4> moved(pairs.size());
            for (uint32_t r = 0; r < n_records; ++r) {
                float4* dst = moved.data() + 4 * (size_t)perm[r];
                const float4* src = pairs.data() + 4 * (size_t)r;
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
                uint32_t lc, rc; memcpy(&lc, &src[3].z, 4); memcpy(&rc, &src[3].w, 4);
                if ((lc & kLeafBit) == 0u) lc = perm[lc];
                if ((rc & kLeafBit) == 0u) rc = perm[rc];
                dst[3].z = AsFloat(lc); dst[3].w = AsFloat(rc);
            }
            pairs.swap(moved);
            for (uint32_t oi = 0; oi < sd.n_objects; ++oi)
                if (objs[oi].kind == CGPT_OBJECT_MESH && (objs[oi].root_code & kLeafBit) == 0u) objs[oi].root_code = perm[objs[oi].root_code];
            n_top_records = (uint32_t)std::min<size_t>(bfs.size(), std::min<size_t>(n_records, kTopRecords));
        }
    }

    // per-object records for the trace kernel's object phase (device_scene.h: obj_trace)
    std::vector<float4> obj_trace(2 * (size_t)sd.n_objects);
    for (uint32_t oi = 0; oi < sd.n_objects; ++oi) {
        const DevObject& d = objs[oi];
        float4& q0 = obj_trace[2 * (size_t)oi]; float4& q1 = obj_trace[2 * (size_t)oi + 1];
        q0 = F4(AsFloat(d.kind), 0.0f, 0.0f, 0.0f); q1 = F4(0.0f, 0.0f, 0.0f, 0.0f);
        if (d.kind == CGPT_OBJECT_MESH) q0.y = AsFloat(d.root_code);
        else if (d.kind == CGPT_OBJECT_SPHERE) { q0.y = d.sphere_center[0]; q0.z = d.sphere_center[1]; q0.w = d.sphere_center[2]; q1.x = d.sphere_radius_sq; }
        else { q0.y = d.plane_normal[0]; q0.z = d.plane_normal[1]; q0.w = d.plane_normal[2]; q1.x = d.plane_point[0]; q1.y = d.plane_point[1]; q1.z = d.plane_point[2]; }
    }

#ifdef CGPT_NODE_SOA
    {   // experiment build: component planes instead of 64-byte records (rt_device.hpp: load_pair_soa)
        std::vector<float4> planes(pairs.size());
        const float* src = reinterpret_cast<const float*>(pairs.data());
        float* dst = reinterpret_cast<float*>(planes.data());
        for (uint32_t r = 0; r < n_records; ++r)
            for (uint32_t c = 0; c < 16u; ++c) dst[(size_t)c * n_records + r] = src[(size_t)r * 16u + c];
        pairs.swap(planes);
    }
#endif
    FreeScene(ctx);
    int rc;
    if ((rc = UploadArray(ctx, &ctx->d_node_pairs, pairs)) != CGPT_OK) return rc;
    if ((rc = UploadArray(ctx, &ctx->d_tri_leaf, tri_leaf)) != CGPT_OK) return rc;
    if ((rc = UploadArray(ctx, &ctx->d_tri_orig, tri_orig)) != CGPT_OK) return rc;
    if ((rc = UploadArray(ctx, &ctx->d_tri_normal, tri_normal)) != CGPT_OK) return rc;
    if ((rc = UploadArray(ctx, &ctx->d_materials, mats)) != CGPT_OK) return rc;
    if ((rc = UploadArray(ctx, &ctx->d_objects, objs)) != CGPT_OK) return rc;
    if ((rc = UploadArray(ctx, &ctx->d_obj_trace, obj_trace)) != CGPT_OK) return rc;
    if ((rc = UploadArray(ctx, &ctx->d_lights, lights)) != CGPT_OK) return rc;

    ctx->scene.node_pairs = ctx->d_node_pairs; ctx->scene.tri_leaf = ctx->d_tri_leaf; ctx->scene.tri_orig = ctx->d_tri_orig; ctx->scene.tri_normal = ctx->d_tri_normal;
    ctx->scene.materials = ctx->d_materials; ctx->scene.objects = ctx->d_objects; ctx->scene.obj_trace = ctx->d_obj_trace; ctx->scene.lights = ctx->d_lights;
    ctx->scene.n_objects = sd.n_objects; ctx->scene.n_lights = sd.n_lights; ctx->scene.stack_depth = stack_depth; ctx->scene.n_top_records = n_top_records; ctx->scene.n_pair_records = n_records; ctx->scene.n_small_tris = n_small_tris;
    ctx->n_materials = sd.n_materials;
    ctx->has_scene = true;
    return CGPT_OK;
}

int EnsureFramebuffer(cgpt_ctx* ctx, uint32_t W, uint32_t H, uint32_t n_rows, const uint32_t key[5])
{
    if (ctx->d_accumulator && ctx->width == W && ctx->height == H && memcmp(ctx->band_key, key, sizeof(ctx->band_key)) == 0) return CGPT_OK;
    FreeFramebuffer(ctx);
    const size_t n = (size_t)W * n_rows;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_accumulator, n * sizeof(float4)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_pixels, n * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_accumulator, 0, n * sizeof(float4), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_pixels, 0, n * sizeof(uint32_t), ctx->stream));
    ctx->width = W; ctx->height = H; ctx->n_rows = n_rows; memcpy(ctx->band_key, key, sizeof(ctx->band_key));
    ctx->num_accumulated = 0;
    return CGPT_OK;
}

// rows of a context: a contiguous band, or interleaved bands of interleave_rows rows (multi-GPU load balance)
struct Band { uint32_t n_rows, first, h, stride; uint32_t key[5]; };
int ResolveBand(cgpt_ctx* ctx, const cgpt_render_params& p, Band& b)
{
    if (p.width == 0 || p.height == 0 || p.row_begin >= p.row_end || p.row_end > p.height)
        return Fail(ctx, CGPT_ERR_INVALID, "bad framebuffer/rows: %ux%u rows [%u,%u)", p.width, p.height, p.row_begin, p.row_end);
    if ((uint64_t)p.width * p.height > 0xFFFFFFFFull) return Fail(ctx, CGPT_ERR_INVALID, "framebuffer too large");
    if (p.interleave_rows == 0 && p.interleave_count == 0) {
        b.n_rows = p.row_end - p.row_begin; b.first = p.row_begin; b.h = b.n_rows; b.stride = 0;
    } else {
        const uint32_t h = p.interleave_rows, R = p.interleave_count, r = p.interleave_index;
        if (h == 0 || R == 0 || r >= R || p.row_begin != 0 || p.row_end != p.height)
            return Fail(ctx, CGPT_ERR_INVALID, "bad interleave: rows %u count %u index %u (row_begin/row_end must be 0/height)", h, R, r);
        b.first = r * h; b.h = h; b.stride = R * h;
        b.n_rows = 0;
        for (uint64_t first = b.first; first < p.height; first += b.stride) b.n_rows += std::min<uint32_t>(h, p.height - (uint32_t)first);
        if (b.n_rows == 0) return Fail(ctx, CGPT_ERR_INVALID, "interleave index %u owns no rows of a %u-row image", r, p.height);
    }
    const uint32_t key[5] = { p.row_begin, p.row_end, p.interleave_rows, p.interleave_count, p.interleave_index };
    memcpy(b.key, key, sizeof(key));
    return CGPT_OK;
}

}  // namespace

extern "C" {

uint32_t cgpt_abi_version(void) { return CGPT_ABI_VERSION; }

const char* cgpt_last_error(const cgpt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int cgpt_ctx_create(const int* device_ids, int n_devices, uint32_t flags, cgpt_ctx** out)
{
    if (!out) return Fail(nullptr, CGPT_ERR_INVALID, "out is null");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 8) return Fail(nullptr, CGPT_ERR_INVALID, "n_devices %d outside [1, 8] (one node)", n_devices);
    if (n_devices > 1 || (flags & CGPT_CTX_FORCE_COLLECTIVE)) {
        try { return GroupCreate(device_ids, n_devices, flags, out); }
        catch (const std::exception& e) { return Fail(nullptr, CGPT_ERR_INVALID, "cgpt_ctx_create: %s", e.what()); }
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return Fail(nullptr, CGPT_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    const int dev = device_ids ? device_ids[0] : 0;
    if (dev < 0 || dev >= count) return Fail(nullptr, CGPT_ERR_INVALID, "device id %d out of range (%d devices)", dev, count);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return Fail(nullptr, CGPT_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return Fail(nullptr, CGPT_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 (MI355X) only", dev, prop.gcnArchName);

    cgpt_ctx* ctx = new (std::nothrow) cgpt_ctx;
    if (!ctx) return Fail(nullptr, CGPT_ERR_INVALID, "out of host memory");
    ctx->device = dev;
    if ((e = hipSetDevice(dev)) != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_start)) != hipSuccess || (e = hipEventCreate(&ctx->ev_stop)) != hipSuccess ||
        (e = hipMalloc((void**)&ctx->d_counters, sizeof(DevCounters))) != hipSuccess ||
        (e = hipMemset(ctx->d_counters, 0, sizeof(DevCounters))) != hipSuccess) {
        int rc = Fail(nullptr, CGPT_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return CGPT_OK;
}

int cgpt_ctx_destroy(cgpt_ctx* ctx)
{
    if (!ctx) return CGPT_OK;
    if (ctx->group) { GroupDestroy(ctx); delete ctx; return CGPT_OK; }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    FreeScene(ctx);
    FreeFramebuffer(ctx);
    (void)hipFree(ctx->d_counters);
    WavefrontFree(ctx->wavefront_state);
    PersistentFree(ctx->persistent_state);
    (void)hipEventDestroy(ctx->ev_start); (void)hipEventDestroy(ctx->ev_stop);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return CGPT_OK;
}

int cgpt_set_stream(cgpt_ctx* ctx, void* hip_stream)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) return Fail(ctx, CGPT_ERR_UNSUPPORTED, "cgpt_set_stream: a multi-device context owns its streams");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return CGPT_OK;
}

int cgpt_scene_upload(cgpt_ctx* ctx, const cgpt_scene_desc* scene)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupSceneUpload(ctx, scene));
    if (!scene) return Fail(ctx, CGPT_ERR_INVALID, "scene is null");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    try {                                                                      // the re-layout allocates host vectors: nothing may unwind through the C ABI
        return BuildDeviceScene(ctx, *scene);
    } catch (const std::exception& e) {
        return Fail(ctx, CGPT_ERR_INVALID, "scene upload: %s", e.what());
    } catch (...) {
        return Fail(ctx, CGPT_ERR_INVALID, "scene upload: unknown exception");
    }
}

int cgpt_scene_update_materials(cgpt_ctx* ctx, const cgpt_material* materials, uint32_t n_materials)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupUpdateMaterials(ctx, materials, n_materials));
    if (!ctx->has_scene) return Fail(ctx, CGPT_ERR_NO_SCENE, "no scene uploaded");
    if (!materials || n_materials != ctx->n_materials) return Fail(ctx, CGPT_ERR_INVALID, "expected %u materials", ctx->n_materials);
    std::vector<float4> mats;
    try { mats.resize(4 * (size_t)n_materials); } catch (const std::exception& e) { return Fail(ctx, CGPT_ERR_INVALID, "out of host memory: %s", e.what()); }
    for (uint32_t i = 0; i < n_materials; ++i) PackMaterial(materials[i], mats.data() + 4 * (size_t)i);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_materials, mats.data(), mats.size() * sizeof(float4), hipMemcpyHostToDevice));
    return CGPT_OK;
}

int cgpt_camera_from_view(const float pos[3], const float view_dir[3], float fov_deg, float aspect, cgpt_camera* out)
{
    if (!pos || !view_dir || !out) return CGPT_ERR_INVALID;
    const float fov = fov_deg * 3.14159265f / 180.0f;                         // Deg2Rad, ref: MathLib.h:9-12
    float center[3];
    for (int i = 0; i < 3; ++i) center[i] = pos[i] + fov * view_dir[i];       // ref: Main.cpp:145
    for (int i = 0; i < 3; ++i) out->pos[i] = pos[i];
    out->top_left[0] = center[0] + -aspect; out->top_left[1] = center[1] + 1.0f; out->top_left[2] = center[2] + 0.0f;
    out->top_right[0] = center[0] + aspect; out->top_right[1] = center[1] + 1.0f; out->top_right[2] = center[2] + 0.0f;
    out->bottom_left[0] = center[0] + -aspect; out->bottom_left[1] = center[1] + -1.0f; out->bottom_left[2] = center[2] + 0.0f;
    return CGPT_OK;
}

}  // extern "C"

namespace cgpt {

int RenderEnqueue(cgpt_ctx* ctx, const cgpt_camera* camera, const cgpt_settings* settings, const cgpt_render_params* p)
{
    ctx->pending_kernel = 0;
    if (!camera || !settings || !p) return Fail(ctx, CGPT_ERR_INVALID, "null argument");
    if (!ctx->has_scene) return Fail(ctx, CGPT_ERR_NO_SCENE, "cgpt_render before cgpt_scene_upload");
    if (settings->max_ray_depth < 0 || settings->max_ray_depth > 254)
        return Fail(ctx, CGPT_ERR_INVALID, "max_ray_depth %d outside [0,254] (ray_depth is a uint8_t in the reference, Main.cpp:401)", settings->max_ray_depth);
    if (settings->render_mode > CGPT_MODE_ADVANCED || settings->debug_render_mode > CGPT_DEBUG_BVH_DEPTH)
        return Fail(ctx, CGPT_ERR_INVALID, "bad render_mode/debug_render_mode");
    if (settings->render_mode != CGPT_MODE_ADVANCED) {
        // TracePath (brute force) keeps its per-level operations in HBM in the persistent kernel (per lane) and in the wavefront
        // pipeline (per path), any depth; the megakernel keeps them in per-lane scratch of 32 levels
        if (p->kernel == CGPT_KERNEL_MEGAKERNEL && settings->max_ray_depth + 1 > 32)
            return Fail(ctx, CGPT_ERR_UNSUPPORTED, "brute-force / comparison modes in the megakernel support max_ray_depth <= 31 (got %d)", settings->max_ray_depth);
    }
    if ((uint64_t)p->first_sample + p->n_samples > 0xFFFFFFFFull) return Fail(ctx, CGPT_ERR_INVALID, "sample index overflow");

    Band band;
    int rc = ResolveBand(ctx, *p, band);
    if (rc != CGPT_OK) return rc;
    const uint32_t n_rows = band.n_rows, band_first = band.first, band_h = band.h, band_stride = band.stride;

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = EnsureFramebuffer(ctx, p->width, p->height, n_rows, band.key);
    if (rc != CGPT_OK) return rc;
    if (p->n_samples == 0) return CGPT_OK;

    DevRenderArgs args{};
    args.scene = ctx->scene;
    memcpy(&args.camera, camera, sizeof(DevCamera));
    args.settings.max_ray_depth = settings->max_ray_depth;
    args.settings.nee = settings->next_event_estimation_enabled;
    args.settings.cosine = settings->cosine_weighted_diffuse_reflection_enabled;
    args.settings.rr = settings->russian_roulette_enabled;
    args.settings.render_mode = settings->render_mode;
    args.settings.debug_mode = settings->debug_render_mode;
    args.width = p->width; args.height = p->height;
    args.n_rows = n_rows; args.band_first = band_first; args.band_h = band_h; args.band_stride = band_stride;
    args.first_sample = p->first_sample; args.n_samples = p->n_samples; args.seed = p->seed;
    args.accumulator = ctx->d_accumulator; args.pixels = ctx->d_pixels; args.counters = ctx->d_counters;

    const bool count = (p->flags & CGPT_RENDER_COUNTERS) != 0;
    // AUTO: all three kernels give bit-identical images, so the choice is speed alone.  MI355X, glass scene, ms per call
    // (profiles/r03/small_calls_table.txt; 16 samples and more: profiles/r02/frame_time_after.txt):
    //                 64x64  1 / 2 / 8 samples     960x540  1 / 2 / 8        1920x1080  1 / 2 / 4 / 8 / 16 / 32 / 64 / 128
    //   megakernel    0.62 / 1.32 / 5.12           1.26 / 2.53 / 9.91        1.68 / 3.17 / 6.06 / 11.9 / 23.1 / ...   one thread walks a pixel's samples: time ~ samples
    //   persistent    0.79 / 0.70 / 0.77           1.68 / 1.82 / 2.35        2.13 / 2.36 / 3.01 / 5.02 / 8.25 / 14.9 / 27.9 / 53.9
    //   wavefront     1.45 / 1.54 / 1.66           2.21 / 2.44 / 2.76        2.48 / 2.98 / 3.57 / 5.13 / 8.35 / 14.7 / 26.7 / 51.1      (256 samples: 93 vs 106 ms)
    // A call cannot finish before its longest path does (~1.2 ms in the megakernel, ~2 ms in the voted kernels on this scene), which
    // is what a one-sample call pays: the megakernel wins every one-sample call, at every frame size; with two or more samples per call
    // the voted kernels win, the persistent kernel (two launches) up to ~16 M paths, the wavefront pipeline beyond -- TracePath /
    // COMPARISON (the reference's default mode) included: 256-spp 1080p COMPARISON 88.6 ms in the pipeline against 95.3 in the
    // persistent kernel, BRUTE_FORCE 71.4 against 73.1 (profiles/r03).
    const uint64_t n_paths = (uint64_t)p->width * n_rows * p->n_samples;
    uint32_t kernel = p->kernel;
    if (kernel == CGPT_KERNEL_AUTO) {
        const bool advanced = settings->render_mode == CGPT_MODE_ADVANCED;
        if (p->n_samples == 1u && n_paths < 3000000ull && (advanced || settings->max_ray_depth + 1 <= 32)) kernel = CGPT_KERNEL_MEGAKERNEL;
        else if (n_paths < 16000000ull) kernel = CGPT_KERNEL_PERSISTENT;   // (40 M until the end of round 3: 1080p x 8 / 16 samples now 4.77 / 7.59 ms in the pipeline, 4.90 / 8.10 here)
        // Beyond that the pipeline, whatever the size of the tree.  (Rounds 2-3 sent trees of more than 800 K child pairs -- 84 MB + 63 MB of
        // leaf triangles at 1.31 M triangles, far beyond the L2 -- to the persistent kernel, then 5 % faster there.  With the ray lists ordered
        // by image band the pipeline leads: rank shares of the 1080p x 1024 / 4K x 4096 spp configurations 34.4-34.6 vs 36.6-36.8 ms and
        // 510 vs 556 ms, the whole 1080p x 1024 frame 260 vs 280 ms; profiles/r03/image_bands.md.)
        else kernel = CGPT_KERNEL_WAVEFRONT;
    }

    if (kernel == CGPT_KERNEL_MEGAKERNEL) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
        HIP_TRY(ctx, LaunchMegakernel(args, count, ctx->stream));
        ctx->kernel_launches += 1;
    } else if (kernel == CGPT_KERNEL_WAVEFRONT) {
        rc = LaunchWavefront(ctx, args, count);
        if (rc < 0) return ctx->error.empty() ? Fail(ctx, CGPT_ERR_HIP, "wavefront launch failed") : CGPT_ERR_HIP;
        ctx->kernel_launches += (uint32_t)rc;
    } else if (kernel == CGPT_KERNEL_PERSISTENT) {
        rc = LaunchPersistent(ctx, args, count);
        if (rc < 0) return ctx->error.empty() ? Fail(ctx, CGPT_ERR_HIP, "persistent kernel launch failed") : CGPT_ERR_HIP;
        ctx->kernel_launches += (uint32_t)rc;
    } else {
        return Fail(ctx, CGPT_ERR_INVALID, "unknown kernel %u", p->kernel);
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->pending_kernel = kernel; ctx->pending_args = args; ctx->pending_num_accumulated = p->first_sample + p->n_samples;
    ctx->last_debug_mode = settings->debug_render_mode;
    ctx->last_kernel = kernel;
    return CGPT_OK;
}

int RenderFinish(cgpt_ctx* ctx)
{
    const uint32_t kernel = ctx->pending_kernel;
    if (kernel == 0) return CGPT_OK;                                           // nothing was enqueued (n_samples == 0)
    ctx->pending_kernel = 0;
    const DevRenderArgs& args = ctx->pending_args;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    ctx->kernel_ms += ms;
    if (kernel == CGPT_KERNEL_MEGAKERNEL) { ctx->dominant_ms += ms; ctx->dominant_launches += 1; ctx->dominant_waves_per_simd = MegakernelWavesPerSimd(args); }
    else if (kernel == CGPT_KERNEL_PERSISTENT) {
        double tms = 0.0; uint32_t tl = 0, w = 0;
        PersistentCollectTiming(ctx->persistent_state, &tms, &tl, &w);
        ctx->dominant_ms += tms; ctx->dominant_launches += tl; ctx->dominant_waves_per_simd = w;
    } else {
        ctx->dominant_waves_per_simd = WavefrontTraceWavesPerSimd(ctx->wavefront_state);
        double tms = 0.0, r0ms = 0.0; uint32_t tl = 0, r0l = 0;
        WavefrontCollectTiming(ctx->wavefront_state, &tms, &tl, &r0ms, &r0l);
        ctx->dominant_ms += tms; ctx->dominant_launches += tl;
        ctx->dominant_round0_ms += r0ms; ctx->dominant_round0_launches += r0l;
    }
    ctx->num_accumulated = ctx->pending_num_accumulated;
    return CGPT_OK;
}

}  // namespace cgpt

extern "C" {

int cgpt_render(cgpt_ctx* ctx, const cgpt_camera* camera, const cgpt_settings* settings, const cgpt_render_params* p)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupRender(ctx, camera, settings, p));
    const int rc = RenderEnqueue(ctx, camera, settings, p);
    return rc != CGPT_OK ? rc : RenderFinish(ctx);
}

int cgpt_reset_accumulator(cgpt_ctx* ctx)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupResetAccumulator(ctx));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->num_accumulated = 0;                                                  // ref: Main.cpp:240-242
    if (ctx->d_accumulator) {
        const size_t n = (size_t)ctx->width * ctx->n_rows;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_accumulator, 0, n * sizeof(float4), ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    double zero = 0.0;
    HIP_TRY(ctx, hipMemcpy(&ctx->d_counters->total_energy, &zero, sizeof(double), hipMemcpyHostToDevice));
    return CGPT_OK;
}

int cgpt_read_accumulator(cgpt_ctx* ctx, float* dst, size_t n_floats)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupReadAccumulator(ctx, dst, n_floats));
    if (!ctx->d_accumulator) return Fail(ctx, CGPT_ERR_INVALID, "nothing rendered yet");
    const size_t n = (size_t)ctx->width * ctx->n_rows * 4;
    if (!dst || n_floats != n) return Fail(ctx, CGPT_ERR_INVALID, "expected a buffer of %zu floats", n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(dst, ctx->d_accumulator, n * sizeof(float), hipMemcpyDeviceToHost));
    return CGPT_OK;
}

int cgpt_read_pixels(cgpt_ctx* ctx, uint32_t* dst, size_t n_pixels)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupReadPixels(ctx, dst, n_pixels));
    if (!ctx->d_pixels) return Fail(ctx, CGPT_ERR_INVALID, "nothing rendered yet");
    const size_t n = (size_t)ctx->width * ctx->n_rows;
    if (!dst || n_pixels != n) return Fail(ctx, CGPT_ERR_INVALID, "expected a buffer of %zu pixels", n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(dst, ctx->d_pixels, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CGPT_OK;
}

int cgpt_write_accumulator(cgpt_ctx* ctx, const cgpt_render_params* p, const float* src, size_t n_floats, uint32_t num_accumulated)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupWriteAccumulator(ctx, p, src, n_floats, num_accumulated));
    if (!p || !src) return Fail(ctx, CGPT_ERR_INVALID, "null argument");
    Band band;
    int rc = ResolveBand(ctx, *p, band);
    if (rc != CGPT_OK) return rc;
    const size_t n = (size_t)p->width * band.n_rows;
    if (n_floats != 4 * n) return Fail(ctx, CGPT_ERR_INVALID, "expected %zu floats for %u rows of %u pixels, got %zu", 4 * n, band.n_rows, p->width, n_floats);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = EnsureFramebuffer(ctx, p->width, p->height, band.n_rows, band.key)) != CGPT_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_accumulator, src, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, LaunchPackPixels(ctx->d_accumulator, ctx->d_pixels, n, num_accumulated, ctx->stream));   // data.pixels, ref: Main.cpp:741
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->num_accumulated = num_accumulated;                                    // ref: Main.cpp:205
    return CGPT_OK;
}

int cgpt_set_tuning(cgpt_ctx* ctx, const char* name, uint32_t value)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupSetTuning(ctx, name, value));
    if (!name) return Fail(ctx, CGPT_ERR_INVALID, "null knob name");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    bool known = false;
    const int rc = PersistentSetTuning(ctx, name, value, &known);              // "pt_*" knobs
    if (known) return rc;
    return WavefrontSetTuning(ctx, name, value);
}

int cgpt_accumulator_device_ptr(cgpt_ctx* ctx, void** ptr, size_t* n_bytes)
{
    if (!ctx || !ptr || !n_bytes) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupDevicePtr(ctx, false, ptr, n_bytes));
    if (!ctx->d_accumulator) return Fail(ctx, CGPT_ERR_INVALID, "nothing rendered yet");
    *ptr = ctx->d_accumulator;
    *n_bytes = (size_t)ctx->width * ctx->n_rows * sizeof(float4);
    return CGPT_OK;
}

int cgpt_pixels_device_ptr(cgpt_ctx* ctx, void** ptr, size_t* n_bytes)
{
    if (!ctx || !ptr || !n_bytes) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupDevicePtr(ctx, true, ptr, n_bytes));
    if (!ctx->d_pixels) return Fail(ctx, CGPT_ERR_INVALID, "nothing rendered yet");
    *ptr = ctx->d_pixels;
    *n_bytes = (size_t)ctx->width * ctx->n_rows * sizeof(uint32_t);
    return CGPT_OK;
}

int cgpt_get_stats(cgpt_ctx* ctx, cgpt_stats* out)
{
    if (!ctx || !out) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupGetStats(ctx, out));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    DevCounters c;
    HIP_TRY(ctx, hipMemcpy(&c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    out->traced_rays = c.traced_rays; out->inner_steps = c.inner_steps; out->tri_tests = c.tri_tests;
    out->bvh_depth_sum = c.bvh_depth_sum; out->closest_hits = c.closest_hits; out->total_energy_received = c.total_energy;
    out->num_accumulated = ctx->num_accumulated; out->kernel_launches = ctx->kernel_launches; out->kernel_ms = ctx->kernel_ms;
    out->dominant_launches = ctx->dominant_launches; out->dominant_waves_per_simd = ctx->dominant_waves_per_simd; out->dominant_ms = ctx->dominant_ms;
    out->gather_ms = 0.0; out->gathers = 0; out->n_devices = 1; out->rccl_ranks = 0; out->last_kernel = ctx->last_kernel;
    memset(out->device_ms, 0, sizeof(out->device_ms)); out->device_ms[0] = ctx->kernel_ms;
    out->dominant_round0_ms = ctx->dominant_round0_ms; out->dominant_round0_launches = ctx->dominant_round0_launches; out->reserved_ = 0;
    return CGPT_OK;
}

int cgpt_reset_stats(cgpt_ctx* ctx)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupResetStats(ctx));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemset(ctx->d_counters, 0, sizeof(DevCounters)));
    ctx->kernel_launches = 0; ctx->kernel_ms = 0.0; ctx->dominant_launches = 0; ctx->dominant_ms = 0.0;
    ctx->dominant_round0_launches = 0; ctx->dominant_round0_ms = 0.0;
    return CGPT_OK;
}

int cgpt_intersect_rays(cgpt_ctx* ctx, const float* origins, const float* dirs, const float* tmax, uint32_t n,
                        float* out_t, uint32_t* out_obj, uint32_t* out_tri, uint32_t* out_depth)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) return GroupForwarded(ctx, cgpt_intersect_rays(GroupFirstMember(ctx), origins, dirs, tmax, n, out_t, out_obj, out_tri, out_depth));
    if (!ctx->has_scene) return Fail(ctx, CGPT_ERR_NO_SCENE, "cgpt_intersect_rays before cgpt_scene_upload");
    if (n == 0) return CGPT_OK;
    if (!origins || !dirs || !out_t || !out_obj || !out_tri || !out_depth) return Fail(ctx, CGPT_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d_o = nullptr, *d_d = nullptr, *d_tm = nullptr, *d_t = nullptr;
    uint32_t *d_obj = nullptr, *d_tri = nullptr, *d_dep = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_tm); (void)hipFree(d_t); (void)hipFree(d_obj); (void)hipFree(d_tri); (void)hipFree(d_dep); };
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return r == hipSuccess; };
    ok(hipMalloc((void**)&d_o, 12 * (size_t)n)); ok(hipMalloc((void**)&d_d, 12 * (size_t)n));
    if (tmax) ok(hipMalloc((void**)&d_tm, 4 * (size_t)n));
    ok(hipMalloc((void**)&d_t, 4 * (size_t)n)); ok(hipMalloc((void**)&d_obj, 4 * (size_t)n));
    ok(hipMalloc((void**)&d_tri, 4 * (size_t)n)); ok(hipMalloc((void**)&d_dep, 4 * (size_t)n));
    if (e == hipSuccess) {
        ok(hipMemcpyAsync(d_o, origins, 12 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        ok(hipMemcpyAsync(d_d, dirs, 12 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        if (tmax) ok(hipMemcpyAsync(d_tm, tmax, 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    }
    if (e == hipSuccess) {
        ok(LaunchIntersectRays(ctx->scene, d_o, d_d, d_tm, n, d_t, d_obj, d_tri, d_dep, ctx->d_counters, ctx->stream));
        ok(hipMemcpyAsync(out_t, d_t, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        ok(hipMemcpyAsync(out_obj, d_obj, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        ok(hipMemcpyAsync(out_tri, d_tri, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        ok(hipMemcpyAsync(out_depth, d_dep, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        ok(hipStreamSynchronize(ctx->stream));
    }
    cleanup();
    if (e != hipSuccess) return Fail(ctx, CGPT_ERR_HIP, "cgpt_intersect_rays: %s", hipGetErrorString(e));
    return CGPT_OK;
}

int cgpt_synchronize(cgpt_ctx* ctx)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (ctx->group) GROUP_CALL(ctx, GroupSynchronize(ctx));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CGPT_OK;
}

}  // extern "C"
