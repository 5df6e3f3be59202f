/*
 * cpugpupt_abi.h -- the drop-in boundary: a C ABI around the reference's Render() (MI355X / gfx950).
 *
 * The reference (Contingencyy/CPUGPUPathtracing) has no plugin or FFI interface; the path sits behind
 * the implicit boundary around `void Render()` (ref: Source/Main.cpp:691-755), which reads the
 * file-static `data` (ref: Main.cpp:200-236) and writes `data.accumulator` / `data.pixels`.  Each
 * entry point below cites the reference interface it replaces.  "ref:" = file:line under the
 * reference checkout.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Conventions: every function returns an int status (0 = CGPT_OK) and never throws; the message for
 * the last failure is available from cgpt_last_error().  One context is used from one host thread at
 * a time (Render() is not re-entrant in the reference either).  Every call blocks until its device work
 * is done -- cgpt_render returns with the accumulator updated, as Render() does; there are no
 * asynchronous variants.  The library copies everything it is handed; the caller keeps ownership of its
 * buffers.
 */
#ifndef CPUGPUPT_ABI_H
#define CPUGPUPT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGPT_ABI_VERSION 2u   /* 2: cgpt_stats grew (gather_ms .. last_kernel), CGPT_KERNEL_* / CGPT_CTX_* values added since 1 */

enum cgpt_status {
    CGPT_OK = 0,
    CGPT_ERR_INVALID = 1,      /* bad argument / inconsistent scene description */
    CGPT_ERR_HIP = 2,          /* a HIP runtime call failed (message has hipGetErrorString) */
    CGPT_ERR_NO_SCENE = 3,     /* render before cgpt_scene_upload */
    CGPT_ERR_UNSUPPORTED = 4,  /* valid in the reference's type system but EXCEPTs there too (e.g. AABB objects,
                                  non-sphere primitive lights: ref Main.cpp:383, Primitives.cpp:304) */
    CGPT_ERR_NO_DEVICE = 5     /* no usable gfx950 device: the product never falls back to a CPU path */
};

/* ---- scene description: the reference's own layouts -------------------------------------------- */

/* ref: Include/Primitives.h:9-13 */
typedef struct cgpt_vertex { float pos[3]; float normal[3]; } cgpt_vertex;
/* ref: Include/Primitives.h:46-51 (72 bytes) */
typedef struct cgpt_triangle { cgpt_vertex v0, v1, v2; } cgpt_triangle;
/* ref: Include/BVH.h:29-34 (32 bytes; right child is always left_first + 1) */
typedef struct cgpt_bvh_node {
    float aabb_min[3]; uint32_t left_first;
    float aabb_max[3]; uint32_t prim_count;
} cgpt_bvh_node;
/* ref: Source/Main.cpp:51-69 (56 bytes; is_light is the reference's bool widened to 4 bytes = its padding) */
typedef struct cgpt_material {
    float albedo[3]; float specular;
    float refractivity; float absorption[3]; float ior;
    float emissive[3]; float intensity;
    uint32_t is_light;
} cgpt_material;

enum cgpt_object_kind {        /* ref: Main.cpp:245-275 (Object = mesh-with-BVH | Primitive) */
    CGPT_OBJECT_MESH = 0,
    CGPT_OBJECT_SPHERE = 1,    /* ref: Primitives.h:36-44 */
    CGPT_OBJECT_PLANE = 2      /* ref: Primitives.h:30-34 */
};

typedef struct cgpt_object {
    uint32_t kind;             /* cgpt_object_kind */
    uint32_t mat_index;        /* ref: Main.cpp:268 */
    /* mesh: slices of the scene-wide arrays below (BVH internals, ref: BVH.h:46-52) */
    uint32_t node_offset, node_count;   /* m_nodes[0 .. m_current_node) ; node 0 is the root */
    uint32_t tri_offset, tri_count;     /* m_triangles and m_tri_indices (indices are object-local) */
    uint32_t max_depth;                 /* BVH::GetMaxDepth, ref: BVH.cpp:139-142 */
    float total_area;                   /* BVH::GetTotalArea, ref: BVH.cpp:144-147 */
    /* sphere */
    float sphere_center[3]; float sphere_radius;
    /* plane */
    float plane_normal[3]; float plane_point[3];
} cgpt_object;

/* what Render() reads from `data`: objects, materials, light_source_indices (ref: Main.cpp:209-212) */
typedef struct cgpt_scene_desc {
    const cgpt_object* objects; uint32_t n_objects;
    const cgpt_bvh_node* nodes; uint32_t n_nodes;
    const cgpt_triangle* triangles; uint32_t n_triangles;
    const uint32_t* tri_indices;               /* n_triangles entries */
    const cgpt_material* materials; uint32_t n_materials;
    const uint32_t* light_indices; uint32_t n_lights;
} cgpt_scene_desc;

/* Camera screen plane (ref: Main.cpp:162-168); cgpt_camera_from_view fills it like UpdateScreenPlane (:143-149) */
typedef struct cgpt_camera {
    float pos[3]; float top_left[3]; float top_right[3]; float bottom_left[3];
} cgpt_camera;

enum cgpt_render_mode { CGPT_MODE_COMPARISON = 0, CGPT_MODE_BRUTE_FORCE = 1, CGPT_MODE_ADVANCED = 2 };      /* ref: Main.cpp:172-178 */
enum cgpt_debug_mode { CGPT_DEBUG_NONE = 0, CGPT_DEBUG_RAY_DEPTH = 1, CGPT_DEBUG_BVH_DEPTH = 2 };           /* ref: Main.cpp:185-191 */

/* ref: Main.cpp:228-235 (Settings) + :215-216 (render_mode, debug_render_mode) */
typedef struct cgpt_settings {
    int32_t max_ray_depth;
    uint32_t next_event_estimation_enabled;
    uint32_t cosine_weighted_diffuse_reflection_enabled;
    uint32_t russian_roulette_enabled;
    uint32_t render_mode;
    uint32_t debug_render_mode;
} cgpt_settings;

/* three render paths with bit-identical results: the one-lane-per-pixel megakernel, the wavefront pipeline (trace / shade /
 * compact kernels per bounce round), and the persistent path kernel (one launch: lanes own paths, voted traversal + shade steps).
 * AUTO picks by speed alone (measured on MI355X): a one-sample call under 3 M paths -> megakernel; under 16 M
 * paths -> persistent; else the wavefront pipeline (which sizes its pools against the
 * free HBM: see cgpt_set_tuning).  All three run every render_mode (TracePathAdvanced, TracePath, COMPARISON) and debug view. */
enum cgpt_kernel { CGPT_KERNEL_AUTO = 0, CGPT_KERNEL_MEGAKERNEL = 1, CGPT_KERNEL_WAVEFRONT = 2, CGPT_KERNEL_PERSISTENT = 3 };
enum cgpt_render_flags { CGPT_RENDER_COUNTERS = 1u };   /* collect inner_steps / tri_tests / bvh_depth_sum / closest_hits */

typedef struct cgpt_render_params {
    uint32_t width, height;        /* framebuffer size (ref: Window::GetFramebufferSize, Main.cpp:698); any size, every
                                      pixel is rendered (the reference needs W%16==0 && H%16==0: SURVEY A-1) */
    uint32_t row_begin, row_end;   /* rows this context renders; [0,height) for the whole image (multi-GPU row tiling) */
    uint32_t first_sample;         /* = data.num_accumulated before the call (ref: Main.cpp:205,702) */
    uint32_t n_samples;            /* Render() calls folded into this one */
    uint32_t seed;                 /* RNG stream key; the reference's s_seed is 0x12345678 (ref: Random.h:4) */
    uint32_t kernel;               /* cgpt_kernel */
    uint32_t flags;                /* cgpt_render_flags */
    /* Interleaved row bands for load-balanced multi-GPU tiling (all zero = the contiguous rows [row_begin,row_end)):
     * with interleave_rows = h, interleave_count = R, interleave_index = r this context renders the global rows
     * (k*R + r)*h + j, k = 0,1,..., 0 <= j < h, that are < height; its band is stored compactly in that order.
     * row_begin/row_end must then be 0/height. */
    uint32_t interleave_rows, interleave_count, interleave_index;
} cgpt_render_params;

/* ref: Main.cpp:218-226 (Statistics), :207 (total_energy_received) + traversal counters for the roofline */
typedef struct cgpt_stats {
    uint64_t traced_rays;          /* IntersectScene calls, ref: Main.cpp:301 */
    uint64_t inner_steps;          /* executions of BVH.cpp:93-98 (valid with CGPT_RENDER_COUNTERS) */
    uint64_t tri_tests;            /* executions of BVH.cpp:76-77 (valid with CGPT_RENDER_COUNTERS) */
    uint64_t bvh_depth_sum;        /* sum of payload.bvh_depth, ref: BVH.cpp:118 (valid with CGPT_RENDER_COUNTERS) */
    uint64_t closest_hits;         /* mesh hits shaded, ref: Main.cpp:332 (valid with CGPT_RENDER_COUNTERS) */
    double total_energy_received;  /* ref: Main.cpp:735 */
    uint32_t num_accumulated;      /* ref: Main.cpp:205 */
    uint32_t kernel_launches;      /* render kernels launched since the last reset */
    double kernel_ms;              /* wall time of the render calls' device work, from hipEvents on the context's stream */
    uint32_t dominant_launches;    /* launches of the dominant kernel (megakernel, or the wavefront trace kernel) ... */
    uint32_t dominant_waves_per_simd; /* resident waves per SIMD of the dominant kernel (occupancy query): the setting the issue roof is measured at */
    double dominant_ms;            /* ... and their summed duration, from hipEvents on the streams they were launched on */
    /* multi-device context (one-device context: n_devices = 1, device_ms[0] = kernel_ms, the rest 0) */
    double gather_ms;              /* summed duration of the framebuffer exchanges since the last reset: grouped RCCL send/recv (or peer
                                      copies) + row reorder, hipEvents on device_ids[0]'s stream (the presenter's side of Main.cpp:935) */
    uint32_t gathers;              /* exchanges since the last reset */
    uint32_t n_devices;            /* devices (ranks) of the context */
    uint32_t rccl_ranks;           /* ranks of the RCCL communicator the exchange runs on (0: one-device context or peer-copy gather) */
    uint32_t last_kernel;          /* cgpt_kernel the last cgpt_render ran (AUTO resolved) */
    double device_ms[8];           /* kernel_ms of every device, rank order (kernel_ms above is their maximum: they run side by side) */
    /* the wavefront pipeline's round-0 trace launches (primary rays: the reference does not jitter, so the 64 rays of a wave are
       identical, SURVEY A-14) are a different population from the later rounds': their share of dominant_ms / dominant_launches */
    double dominant_round0_ms;
    uint32_t dominant_round0_launches;
    uint32_t reserved_;
} cgpt_stats;

typedef struct cgpt_ctx cgpt_ctx;

uint32_t cgpt_abi_version(void);

/* replaces ThreadPool::Init / Exit (ref: Main.cpp:773,944; ThreadPool.cpp:81-121): binds 1..8 HIP devices of one node.
 * n_devices == 1: one GPU, one stream.  n_devices > 1: ONE context for the whole frame, as the reference has one Render() --
 * every call below takes the full image; the library cuts it into bands of rows dealt round-robin over the devices, renders
 * them side by side (scene replicated, no exchange while tracing), and cgpt_read_accumulator / cgpt_read_pixels run one grouped
 * RCCL exchange of the float4 bands to device_ids[0] over xGMI (csrc/device/multi_gpu.hip).  The image is bit-identical for
 * any device count.  Such a context takes row_begin = 0, row_end = height and no interleave in cgpt_render_params. */
enum cgpt_ctx_flags {
    CGPT_CTX_FORCE_COLLECTIVE = 1u,  /* n_devices == 1 too goes through the multi-device code: tiling, RCCL exchange (with itself), reorder */
    CGPT_CTX_GATHER_PEER_COPY = 2u   /* gather with hipMemcpyPeerAsync instead of RCCL; device ids may then repeat (several ranks on
                                        one GPU: how the tiling is tested on a one-GPU box) */
};
int cgpt_ctx_create(const int* device_ids, int n_devices, uint32_t flags, cgpt_ctx** out);
int cgpt_ctx_destroy(cgpt_ctx* ctx);
/* replaces EXCEPT/LOG_ERR (ref: Common.h:9): message of the last failing call on ctx (or of the last failing
 * cgpt_ctx_create when ctx is NULL). Never NULL. */
const char* cgpt_last_error(const cgpt_ctx* ctx);
/* run on a caller-provided hipStream_t (e.g. torch's current stream); NULL restores the context's own stream */
int cgpt_set_stream(cgpt_ctx* ctx, void* hip_stream);

/* replaces the implicit use of data.objects / materials / light_source_indices (ref: Main.cpp:209-212, 303-315) */
int cgpt_scene_upload(cgpt_ctx* ctx, const cgpt_scene_desc* scene);
/* replaces Material::RenderImGui edits (ref: Main.cpp:71-91,263-265); the caller resets the accumulator as the reference does */
int cgpt_scene_update_materials(cgpt_ctx* ctx, const cgpt_material* materials, uint32_t n_materials);

/* UpdateScreenPlane (ref: Main.cpp:98-102,143-149): fov in degrees, plane at distance fov-in-radians (SURVEY A-13) */
int cgpt_camera_from_view(const float pos[3], const float view_dir[3], float fov_deg, float aspect, cgpt_camera* out);

/* replaces Render() x n_samples (ref: Main.cpp:691-755): accumulates into the context's float4 accumulator */
int cgpt_render(cgpt_ctx* ctx, const cgpt_camera* camera, const cgpt_settings* settings, const cgpt_render_params* params);
/* replaces ResetAccumulator (ref: Main.cpp:238-243) */
int cgpt_reset_accumulator(cgpt_ctx* ctx);
/* replaces reads of data.accumulator (ref: Main.cpp:204,740): (row_end-row_begin)*width*4 floats of the last render's band */
int cgpt_read_accumulator(cgpt_ctx* ctx, float* dst, size_t n_floats);
/* replaces data.pixels -> DX12::CopyToBackBuffer (ref: Main.cpp:203,741,935; packing MathLib.h:144-152) */
int cgpt_read_pixels(cgpt_ctx* ctx, uint32_t* dst, size_t n_pixels);
/* restores what cgpt_read_accumulator saved (checkpoint / resume of a long render): data.accumulator + data.num_accumulated
 * (ref: Main.cpp:204-205, the state ResetAccumulator :238-243 clears).  `band` names the framebuffer the floats belong to
 * (width, height, row_begin/row_end, interleave_*; its sample / seed / kernel fields are ignored), so it works on a fresh
 * context; src holds rows*width*4 floats in the band's own row order.  data.pixels is re-packed from the loaded sums.
 * Continue with cgpt_render(first_sample = num_accumulated): the result is bit-identical to an uninterrupted render. */
int cgpt_write_accumulator(cgpt_ctx* ctx, const cgpt_render_params* band, const float* src, size_t n_floats, uint32_t num_accumulated);
/* device pointers of the band just rendered (one-device context: for a gather by the host, e.g. torch.distributed in bench.py) or
 * of the gathered full frame on device_ids[0] (multi-device context) */
int cgpt_accumulator_device_ptr(cgpt_ctx* ctx, void** ptr, size_t* n_bytes);
int cgpt_pixels_device_ptr(cgpt_ctx* ctx, void** ptr, size_t* n_bytes);

/* replaces data.stats / total_energy_received (ref: Main.cpp:207,218-226,847-848) */
int cgpt_get_stats(cgpt_ctx* ctx, cgpt_stats* out);
int cgpt_reset_stats(cgpt_ctx* ctx);

/* IntersectScene for a batch of host rays (ref: Main.cpp:299-316): the extend kernel on its own.
 * origins/dirs: n*3 floats; tmax: n floats or NULL (1e34f, ref: Primitives.h:75); outputs n entries each:
 * t, obj_idx (~0u on miss), tri_idx, bvh_depth (ref: Primitives.h:77-82) */
int cgpt_intersect_rays(cgpt_ctx* ctx, const float* origins, const float* dirs, const float* tmax, uint32_t n,
                        float* out_t, uint32_t* out_obj, uint32_t* out_tri, uint32_t* out_depth);

/* BVH::Build with BVHBuildOption_SAHSplitIntervals on the GPU (ref: Source/BVH.cpp:11-45,204-259,299-366; Main.cpp:789,802 use
 * this option for every mesh).  The result is the reference's tree bit for bit: same 32-byte nodes in the same allocation order,
 * same m_tri_indices permutation, same m_max_depth / m_total_area -- so it can be passed to cgpt_scene_upload (cgpt_object
 * node_offset/node_count/max_depth/total_area + cgpt_scene_desc tri_indices) in place of the host build.
 * triangles: n_tris host triangles; nodes_out: room for 2*n_tris-1 nodes; tri_indices_out: n_tris entries. */
int cgpt_bvh_build(cgpt_ctx* ctx, const cgpt_triangle* triangles, uint32_t n_tris, cgpt_bvh_node* nodes_out, uint32_t* n_nodes_out,
                   uint32_t* tri_indices_out, uint32_t* max_depth_out, float* total_area_out);
/* The same for any BuildOption (ref: Include/BVH.h:7-13; Source/BVH.cpp:208-224 naive split, :225-259 SAH split intervals, :260-297
 * SAH split primitives, which never splits: SURVEY A-5) and for BVH::Rebuild (ref: BVH.cpp:47-59, the "Rebuild BVH" button :182-185):
 * initial_tri_indices = NULL starts from the identity order as Build does (:25-29); otherwise it is the tree's CURRENT m_tri_indices
 * (a permutation of 0..n_tris-1), which Rebuild does not reset -- the swap partition is order-sensitive, so the result differs from a
 * fresh Build and equals the reference's Rebuild. */
enum cgpt_bvh_build_option { CGPT_BUILD_NAIVE_SPLIT = 0, CGPT_BUILD_SAH_SPLIT_INTERVALS = 1, CGPT_BUILD_SAH_SPLIT_PRIMITIVES = 2 };
int cgpt_bvh_build_ex(cgpt_ctx* ctx, const cgpt_triangle* triangles, uint32_t n_tris, uint32_t build_option, const uint32_t* initial_tri_indices,
                      cgpt_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* tri_indices_out, uint32_t* max_depth_out, float* total_area_out);

int cgpt_synchronize(cgpt_ctx* ctx);

/* ---- tuning and measurement aids (no reference counterpart) ------------------------------------------------------- */
/* Overrides one tuning knob of the wavefront pipeline for this context (the CGPT_WF_* environment variables set the same
 * knobs process-wide; names are the variable names without the prefix, lower case: "pools", "batch", "max_batch",
 * "refill", "inner_repeat", ...; DESIGN.md 5.1 lists them).  Results never depend on a knob, only speed does. */
int cgpt_set_tuning(cgpt_ctx* ctx, const char* name, uint32_t value);
/* Measures the vector-instruction issue rate of the device -- the roof bench.py prices the trace kernel against: every
 * SIMD of every CU runs `iters` x 64 independent instructions of one kind per wave at `waves_per_simd` resident waves.
 * kind: 0 v_mul_f32, 1 v_pk_mul_f32, 2 v_pk_add_f32, 3 v_rcp_f32, 4/5/6 scalar : packed mixes (3:1 interleaved, 3:1 grouped,
 * 1:1 alternating), 7 v_cndmask_b32 (VCC), 8 v_mul_lo_u32, 9 v_cndmask_b32_e64 (SGPR pair), 10 v_cmp + v_cndmask pairs,
 * 11 v_add_u32, 12 v_min3_f32.
 * Returns wave64 instructions per second over the whole chip (and the launch duration in ms_out, may be NULL). */
int cgpt_measure_issue_rate(cgpt_ctx* ctx, uint32_t kind, uint32_t waves_per_simd, uint32_t iters, double* wave_insts_per_sec, double* ms_out);

#ifdef __cplusplus
}
#endif
#endif
