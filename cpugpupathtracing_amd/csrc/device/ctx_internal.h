// ctx_internal.h -- the context object behind the C ABI (shared by cgpt_abi.hip and multi_gpu.hip; not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "cpugpupt_abi.h"
#include "device_scene.h"

namespace cgpt { struct DeviceGroup; }

struct cgpt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::string error;

    // device scene
    float4* d_node_pairs = nullptr;
    float4* d_tri_leaf = nullptr;
    float4* d_tri_orig = nullptr;
    float4* d_tri_normal = nullptr;
    float4* d_materials = nullptr;
    cgpt::DevObject* d_objects = nullptr;
    float4* d_obj_trace = nullptr;
    uint32_t* d_lights = nullptr;
    cgpt::DevScene scene{};
    uint32_t n_materials = 0;
    bool has_scene = false;

    // framebuffer band
    float4* d_accumulator = nullptr;
    uint32_t* d_pixels = nullptr;
    uint32_t width = 0, height = 0, n_rows = 0;
    uint32_t band_key[5] = { 0, 0, 0, 0, 0 };     // row_begin, row_end, interleave rows/count/index of the allocated band
    uint32_t num_accumulated = 0;

    cgpt::DevCounters* d_counters = nullptr;
    uint32_t kernel_launches = 0;
    double kernel_ms = 0.0;
    uint32_t dominant_launches = 0;
    double dominant_ms = 0.0;
    uint32_t dominant_waves_per_simd = 0;
    uint32_t dominant_round0_launches = 0;
    double dominant_round0_ms = 0.0;

    // wavefront workspace (owned by wavefront_kernels.hip) and the persistent kernel's (persistent_kernel.hip)
    void* wavefront_state = nullptr;
    void* persistent_state = nullptr;

    // a render that has been enqueued and not yet finished (RenderEnqueue / RenderFinish)
    uint32_t pending_kernel = 0;
    uint32_t pending_num_accumulated = 0;
    cgpt::DevRenderArgs pending_args{};
    uint32_t last_debug_mode = 0;
    uint32_t last_kernel = 0;                     // cgpt_kernel the last render ran (AUTO resolved)

    // n_devices > 1 (or CGPT_CTX_FORCE_COLLECTIVE): this context is a group; the members are ordinary one-device contexts
    cgpt::DeviceGroup* group = nullptr;
};


namespace cgpt {
// the two halves of cgpt_render: enqueue the kernels of one context without waiting, then wait and book the timings
int RenderEnqueue(cgpt_ctx* ctx, const cgpt_camera* camera, const cgpt_settings* settings, const cgpt_render_params* p);
int RenderFinish(cgpt_ctx* ctx);
int CtxFail(cgpt_ctx* ctx, int code, const char* fmt, ...);
int CreateFail(int code, const char* fmt, ...);
hipError_t LaunchPackPixels(const float4* accumulator, uint32_t* pixels, size_t n_pixels, uint32_t num_accumulated, hipStream_t stream);   // path_kernels.hip

// multi_gpu.hip: the group behind a multi-device context
int GroupCreate(const int* device_ids, int n_devices, uint32_t flags, cgpt_ctx** out);
void GroupDestroy(cgpt_ctx* ctx);
int GroupSceneUpload(cgpt_ctx* ctx, const cgpt_scene_desc* scene);
int GroupUpdateMaterials(cgpt_ctx* ctx, const cgpt_material* materials, uint32_t n);
int GroupRender(cgpt_ctx* ctx, const cgpt_camera* camera, const cgpt_settings* settings, const cgpt_render_params* p);
int GroupResetAccumulator(cgpt_ctx* ctx);
int GroupReadAccumulator(cgpt_ctx* ctx, float* dst, size_t n_floats);
int GroupReadPixels(cgpt_ctx* ctx, uint32_t* dst, size_t n_pixels);
int GroupWriteAccumulator(cgpt_ctx* ctx, const cgpt_render_params* p, const float* src, size_t n_floats, uint32_t num_accumulated);
int GroupDevicePtr(cgpt_ctx* ctx, bool pixels, void** ptr, size_t* n_bytes);
int GroupGetStats(cgpt_ctx* ctx, cgpt_stats* out);
int GroupResetStats(cgpt_ctx* ctx);
int GroupSetTuning(cgpt_ctx* ctx, const char* name, uint32_t value);
int GroupSynchronize(cgpt_ctx* ctx);
cgpt_ctx* GroupFirstMember(cgpt_ctx* ctx);
cgpt_ctx* GroupFirstMemberOrNull(cgpt_ctx* ctx);
int GroupForwarded(cgpt_ctx* ctx, int rc);     // a call forwarded to the first member returned rc: its message becomes the group's
}  // namespace cgpt
