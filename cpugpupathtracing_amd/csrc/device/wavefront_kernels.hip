// wavefront_kernels.hip -- the wavefront pipeline (gfx950): generate -> per bounce [trace -> shade] -> accumulate.
//
// Why: in the megakernel a wave's traversal loop runs until its slowest lane is done (max-vs-mean ray length) and a tile
// runs until its most expensive pixel is done; PMC showed ~10 % active lanes per VALU instruction.  Here
//   * trace<>  is a PERSISTENT kernel: every wave keeps its 64 lanes filled from a global ray queue (one chunked atomic per
//     256 rays, lane slots handed out with __ballot + mbcnt), each lane runs the reference's ordered stack traversal
//     (ref: Source/BVH.cpp:61-127) on a per-wavefront LDS stack, and a lane that finishes its ray is refilled while its
//     neighbours keep going.  Extend rays and NEE shadow rays share the queue (both are closest-hit IntersectScene calls,
//     ref: Source/Main.cpp:299-316,452-453); a shadow ray's epilogue adds its pending contribution to the path's energy.
//   * shade<>  runs shade_bounce() (ref: Main.cpp:404-573) for every extend hit and compacts the surviving paths' next
//     rays and shadow rays into the next queue with __ballot/popcount (one atomic per wave).
//   * accumulate adds the finished samples to the float4 accumulator IN SAMPLE ORDER, so the image is bit-identical to
//     the megakernel's and the oracle's (ref: Main.cpp:735-746).
// Queue entry (48 B, three float4 planes indexed by queue position -> coalesced):
//   A = {o.xyz, t}   B = {d.xyz, bits(path id | shadow << 31)}   C = extend: {bits obj, tri, bvh_depth, -} (in: initial payload,
//   out: hit record) | shadow: {pending.xyz, -}.  Extend entries grow from the front of the buffer, shadow entries from the back.
// Path state (32 B per path, indexed by path id = sample_in_batch * n_pixels + pixel): {throughput.xyz, bits(depth | spec << 8)},
//   {energy.xyz, bits(rng)}.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <new>

#include "cpugpupt_abi.h"
#include "device_scene.h"
#include "rt_device.hpp"
#include "shade_device.hpp"

namespace cgpt {

using namespace dev;

hipStream_t CtxStream(cgpt_ctx* ctx);
void** CtxWavefrontSlot(cgpt_ctx* ctx);
int CtxFail(cgpt_ctx* ctx, int code, const char* fmt, ...);

extern __shared__ uint32_t lds_stack[];

static constexpr uint32_t kShadowBit = 0x80000000u;
static constexpr uint32_t kStartObject = 0xFFFFFFFFu;   // traversal code: "begin the next object of the scene"
static constexpr uint32_t kChunk = 256;                  // rays a wave takes from the queue per atomic
static constexpr uint32_t kRefillIdleLanes = 16;         // leave the traversal loop to refill once this many lanes are idle
static constexpr uint32_t kCountStride = 4;              // counts[round] = {n_extend, n_shadow, head, -}

struct WfDev {
    float4* A[2]; float4* B[2]; float4* C[2];
    float4* st_tp; float4* st_en;
    uint32_t* counts;
    uint32_t cap;           // paths in the pool; each queue buffer holds 2 * cap entries
    uint32_t n_pixels;      // padded pixel count of the band (8x8 tiles)
    uint32_t tiles_x;       // 8x8 tiles per row
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t rank_in_mask(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t wave_broadcast0(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// pixel of padded index p: 8x8 tiles in row-major tile order (a wave of consecutive p covers one tile)
__device__ __forceinline__ bool pixel_of(const DevRenderArgs& a, const WfDev& wf, uint32_t p, uint32_t& px, uint32_t& py)
{
    const uint32_t tile = p >> 6, l = p & 63u;
    px = (tile % wf.tiles_x) * 8u + (l & 7u);
    py = a.row_begin + (tile / wf.tiles_x) * 8u + (l >> 3);
    return px < a.width && py < a.row_end;
}

// ---- K1 generate: primary rays of one batch of samples (ref: Main.cpp:713-716, Camera::GetRay :133-140) ----------------
__global__ void __launch_bounds__(256) wf_generate(const DevRenderArgs args, const WfDev wf, uint32_t batch_first, uint32_t batch_n)
{
    const uint32_t n_paths = wf.n_pixels * batch_n;
    uint32_t* n_ext = &wf.counts[0];
    for (uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; base < n_paths; base += gridDim.x * blockDim.x) {
        const uint32_t pid = base + lane_id();
        bool valid = pid < n_paths;
        uint32_t px = 0, py = 0;
        if (valid) valid = pixel_of(args, wf, pid % wf.n_pixels, px, py);
        const unsigned long long m = __ballot(valid);
        uint32_t first = 0;
        if (lane_id() == 0 && m) first = atomicAdd(n_ext, (uint32_t)__popcll(m));
        first = wave_broadcast0(first);
        if (valid) {
            const uint32_t s = batch_first + pid / wf.n_pixels;
            const uint32_t rng = pcg_seed(py * args.width + px, s, args.seed);
            const Ray ray = camera_ray(args.camera, (float)px * (1.0f / (float)args.width), (float)py * (1.0f / (float)args.height));
            const uint32_t pos = first + rank_in_mask(m);
            float4 a, b, c;
            a.x = ray.o.x; a.y = ray.o.y; a.z = ray.o.z; a.w = ray.t;
            b.x = ray.d.x; b.y = ray.d.y; b.z = ray.d.z; b.w = __uint_as_float(pid);
            c.x = __uint_as_float(kNoHit); c.y = __uint_as_float(0u); c.z = __uint_as_float(0u); c.w = 0.0f;
            wf.A[0][pos] = a; wf.B[0][pos] = b; wf.C[0][pos] = c;
            float4 tp, en;
            tp.x = 1.0f; tp.y = 1.0f; tp.z = 1.0f; tp.w = __uint_as_float(0u);
            en.x = 0.0f; en.y = 0.0f; en.z = 0.0f; en.w = __uint_as_float(rng);
            wf.st_tp[pid] = tp; wf.st_en[pid] = en;
        }
    }
}

// ---- K2/K4 trace: persistent closest-hit traversal with per-lane refill ------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(256) wf_trace(const DevScene sc, const WfDev wf, uint32_t round, uint32_t buf, DevCounters* counters)
{
    uint32_t* const stack = lds_stack + threadIdx.x;
    const uint32_t stride = blockDim.x;
    const uint32_t n_ext = wf.counts[round * kCountStride + 0];
    const uint32_t n_sh = wf.counts[round * kCountStride + 1];
    const uint32_t total = n_ext + n_sh;
    uint32_t* const head = &wf.counts[round * kCountStride + 2];
    float4* const A = wf.A[buf]; const float4* const B = wf.B[buf]; float4* const C = wf.C[buf];
    const uint32_t last_pos = 2u * wf.cap - 1u;

    uint32_t w_next = 0, w_end = 0;          // wave-uniform chunk of queue indices
    bool exhausted = total == 0;

    bool has_ray = false;
    V3 o = mk(0.0f), d = mk(0.0f), inv = mk(0.0f);
    float t = 0.0f;
    uint32_t obj = kNoHit, tri = 0, depth = 0, cur_obj = 0, code = kStartObject, sp = 0, pos = 0, pidk = 0;
    Counters cnt = { 0, 0, 0, 0, 0 };

    for (;;) {
        // ---- refill idle lanes from the queue ----
        const unsigned long long need = __ballot(!has_ray);
        if (need) {
            if (w_next == w_end && !exhausted) {
                uint32_t base = 0;
                if (lane_id() == 0) base = atomicAdd(head, kChunk);
                base = wave_broadcast0(base);
                if (base >= total) exhausted = true;
                else { w_next = base; w_end = min(base + kChunk, total); }
            }
            if (w_next < w_end) {
                const uint32_t avail = w_end - w_next;
                const uint32_t rank = rank_in_mask(need);
                if (!has_ray && rank < avail) {
                    const uint32_t i = w_next + rank;
                    pos = i < n_ext ? i : last_pos - (i - n_ext);
                    const float4 a = A[pos], b = B[pos];
                    o = mk(a.x, a.y, a.z); t = a.w; d = mk(b.x, b.y, b.z); pidk = __float_as_uint(b.w);
                    inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);             // Ray ctor, ref: Primitives.h:64
                    if (pidk & kShadowBit) { obj = kNoHit; tri = 0; depth = 0; }
                    else { const float4 c = C[pos]; obj = __float_as_uint(c.x); tri = __float_as_uint(c.y); depth = __float_as_uint(c.z); }
                    cur_obj = 0; code = kStartObject; sp = 0; has_ray = true;
                    cnt.rays++;
                }
                w_next += min((uint32_t)__popcll(need), avail);
            }
        }
        if (__ballot(has_ray) == 0ull) {
            if (exhausted) break;
            continue;
        }
        const bool can_refill = !(exhausted && w_next == w_end);

        // ---- traversal until enough lanes are idle ----
        for (;;) {
            // begin the next object / finish the ray (IntersectScene's object loop, ref: Main.cpp:303-315)
            while (has_ray && code == kStartObject) {
                if (cur_obj >= sc.n_objects) {
                    if (pidk & kShadowBit) {                                  // connect epilogue, ref: Main.cpp:454-463
                        if (obj == kNoHit) {
                            const float4 pe = C[pos];
                            const uint32_t pid = pidk & ~kShadowBit;
                            float4 en = wf.st_en[pid];
                            en.x += pe.x; en.y += pe.y; en.z += pe.z;
                            wf.st_en[pid] = en;
                        }
                    } else {
                        reinterpret_cast<float*>(&A[pos])[3] = t;
                        float4 c; c.x = __uint_as_float(obj); c.y = __uint_as_float(tri); c.z = __uint_as_float(depth); c.w = 0.0f;
                        C[pos] = c;
                    }
                    has_ray = false;
                    break;
                }
                const DevObject& ob = sc.objects[cur_obj];
                if (ob.kind == 0u) { code = ob.root_code; sp = 0; }
                else {
                    bool hit;
                    if (ob.kind == 1u) hit = intersect_sphere(mk(ob.sphere_center), ob.sphere_radius_sq, o, d, t);
                    else hit = intersect_plane(mk(ob.plane_normal), mk(ob.plane_point), o, d, t);
                    if (hit) obj = cur_obj;
                    cur_obj++;
                }
            }

            const bool at_leaf = has_ray && (code & kLeafBit) != 0u;
            const bool at_inner = has_ray && !at_leaf;
            const unsigned long long inner_m = __ballot(at_inner), leaf_m = __ballot(at_leaf);
            const uint32_t n_inner = (uint32_t)__popcll(inner_m), n_leaf = (uint32_t)__popcll(leaf_m);
            const uint32_t n_busy = n_inner + n_leaf;
            if (n_busy == 0u) break;
            if (can_refill && 64u - n_busy >= kRefillIdleLanes) break;

            if (n_inner >= n_leaf) {
                if (at_inner) {                                               // one inner step, ref: BVH.cpp:93-123
                    const float4* pair = sc.node_pairs + 4u * (size_t)code;
                    const float4 lmin = pair[0], lmax = pair[1], rmin = pair[2], rmax = pair[3];
                    if (COUNT) cnt.inner++;
                    float left_dist = intersect_aabb(lmin, lmax, o, inv, t);
                    float right_dist = intersect_aabb(rmin, rmax, o, inv, t);
                    uint32_t left_code = __float_as_uint(lmin.w), right_code = __float_as_uint(rmin.w);
                    if (left_dist > right_dist) {
                        float td = left_dist; left_dist = right_dist; right_dist = td;
                        uint32_t tc = left_code; left_code = right_code; right_code = tc;
                    }
                    if (left_dist == 1e30f) {
                        if (sp == 0) { cur_obj++; code = kStartObject; }
                        else code = stack[(--sp) * stride];
                    } else {
                        depth++;
                        if (COUNT) cnt.depth++;
                        code = left_code;
                        if (right_dist != 1e30f) stack[(sp++) * stride] = right_code;
                    }
                }
            } else {
                if (at_leaf) {                                                // one triangle of the leaf, ref: BVH.cpp:74-84
                    const uint32_t i = code & ~kLeafBit;
                    const float4* rec = sc.tri_leaf + 3u * (size_t)i;
                    const float4 a = rec[0], b = rec[1], c = rec[2];
                    if (COUNT) cnt.tris++;
                    if (intersect_triangle(mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), o, d, t)) {
                        tri = __float_as_uint(c.y);
                        obj = cur_obj;                                        // ref: Main.cpp:313-314
                    }
                    if (__float_as_uint(c.z) != 0u) {                         // last triangle of the leaf: pop (ref: BVH.cpp:86-90)
                        if (sp == 0) { cur_obj++; code = kStartObject; }
                        else code = stack[(--sp) * stride];
                    } else {
                        code = kLeafBit | (i + 1u);
                    }
                }
            }
        }
    }

    wave_add_u64(&counters->traced_rays, cnt.rays);
    if (COUNT) {
        wave_add_u64(&counters->inner_steps, cnt.inner);
        wave_add_u64(&counters->tri_tests, cnt.tris);
        wave_add_u64(&counters->bvh_depth_sum, cnt.depth);
    }
}

// ---- K3 shade: one bounce per extend hit, compaction of the next queue ----------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(256) wf_shade(const DevRenderArgs args, const WfDev wf, uint32_t round, uint32_t buf)
{
    const DevScene& sc = args.scene;
    const uint32_t n_ext = wf.counts[round * kCountStride + 0];
    uint32_t* const out_ext = &wf.counts[(round + 1u) * kCountStride + 0];
    uint32_t* const out_sh = &wf.counts[(round + 1u) * kCountStride + 1];
    const float4* const A = wf.A[buf]; const float4* const B = wf.B[buf]; const float4* const C = wf.C[buf];
    float4* const oA = wf.A[buf ^ 1u]; float4* const oB = wf.B[buf ^ 1u]; float4* const oC = wf.C[buf ^ 1u];
    const uint32_t last_pos = 2u * wf.cap - 1u;
    Counters cnt = { 0, 0, 0, 0, 0 };

    for (uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; base < n_ext; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + lane_id();
        const bool valid = i < n_ext;
        bool emit_ext = false, emit_sh = false;
        Ray ray = make_ray(mk(0.0f), mk(0.0f), 0.0f), shadow = ray;
        V3 pending = mk(0.0f);
        uint32_t pid = 0;
        if (valid) {
            const float4 a = A[i], b = B[i], c = C[i];
            ray.o = mk(a.x, a.y, a.z); ray.t = a.w; ray.d = mk(b.x, b.y, b.z);
            ray.obj = __float_as_uint(c.x); ray.tri = __float_as_uint(c.y); ray.bvh_depth = __float_as_uint(c.z);
            pid = __float_as_uint(b.w);
            const float4 tp = wf.st_tp[pid], en = wf.st_en[pid];
            PathState ps;
            ps.throughput = mk(tp.x, tp.y, tp.z); ps.energy = mk(en.x, en.y, en.z);
            ps.rng = __float_as_uint(en.w);
            const uint32_t fl = __float_as_uint(tp.w);
            ps.depth = fl & 0xFFu; ps.is_specular = (fl & 0x100u) != 0u;

            const uint32_t flags = shade_bounce<COUNT>(sc, args.settings, ray, ps, shadow, pending, cnt);
            emit_ext = (flags & kBounceTerminate) == 0u;
            emit_sh = (flags & kBounceShadow) != 0u;

            float4 tpo, eno;
            tpo.x = ps.throughput.x; tpo.y = ps.throughput.y; tpo.z = ps.throughput.z;
            tpo.w = __uint_as_float((ps.depth & 0xFFu) | (ps.is_specular ? 0x100u : 0u));
            eno.x = ps.energy.x; eno.y = ps.energy.y; eno.z = ps.energy.z; eno.w = __uint_as_float(ps.rng);
            wf.st_tp[pid] = tpo; wf.st_en[pid] = eno;
        }
        // ---- active-lane compaction: __ballot + popcount, one atomic per wave and queue ----
        const unsigned long long m_ext = __ballot(emit_ext), m_sh = __ballot(emit_sh);
        uint32_t first_ext = 0, first_sh = 0;
        if (lane_id() == 0) {
            if (m_ext) first_ext = atomicAdd(out_ext, (uint32_t)__popcll(m_ext));
            if (m_sh) first_sh = atomicAdd(out_sh, (uint32_t)__popcll(m_sh));
        }
        first_ext = wave_broadcast0(first_ext); first_sh = wave_broadcast0(first_sh);
        if (emit_ext) {
            const uint32_t p = first_ext + rank_in_mask(m_ext);
            float4 a, b, c;
            a.x = ray.o.x; a.y = ray.o.y; a.z = ray.o.z; a.w = ray.t;
            b.x = ray.d.x; b.y = ray.d.y; b.z = ray.d.z; b.w = __uint_as_float(pid);
            c.x = __uint_as_float(ray.obj); c.y = __uint_as_float(ray.tri); c.z = __uint_as_float(ray.bvh_depth); c.w = 0.0f;
            oA[p] = a; oB[p] = b; oC[p] = c;
        }
        if (emit_sh) {
            const uint32_t p = last_pos - (first_sh + rank_in_mask(m_sh));
            float4 a, b, c;
            a.x = shadow.o.x; a.y = shadow.o.y; a.z = shadow.o.z; a.w = shadow.t;
            b.x = shadow.d.x; b.y = shadow.d.y; b.z = shadow.d.z; b.w = __uint_as_float(pid | kShadowBit);
            c.x = pending.x; c.y = pending.y; c.z = pending.z; c.w = 0.0f;
            oA[p] = a; oB[p] = b; oC[p] = c;
        }
    }
    if (COUNT) wave_add_u64(&args.counters->closest_hits, cnt.hits);
}

// ---- K5 accumulate + pack: samples of the batch in order (ref: Main.cpp:735-746, MathLib.h:144-152) ------------------------
__global__ void __launch_bounds__(256) wf_accumulate(const DevRenderArgs args, const WfDev wf, uint32_t batch_first, uint32_t batch_n)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    double energy_sum = 0.0;
    uint32_t px = 0, py = 0;
    if (p < wf.n_pixels && pixel_of(args, wf, p, px, py)) {
        const size_t local_index = (size_t)(py - args.row_begin) * args.width + px;
        const DevSettings& st = args.settings;
        float4 acc = args.accumulator[local_index];
        V3 last = mk(0.0f);
        for (uint32_t s = 0; s < batch_n; ++s) {
            const uint32_t pid = s * wf.n_pixels + p;
            const float4 en = wf.st_en[pid];
            PathState ps;
            ps.energy = mk(en.x, en.y, en.z);
            ps.depth = 0;
            if (st.debug_mode == 1u) ps.depth = __float_as_uint(wf.st_tp[pid].w) & 0xFFu;
            const V3 e = final_energy(st, ps);
            energy_sum += (double)(e.x + e.y + e.z) * 0.001;
            if (st.debug_mode == 0u) { acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += 1.0f; }
            else last = e;
        }
        if (st.debug_mode == 0u) {
            args.accumulator[local_index] = acc;
            const float n = (float)(batch_first + batch_n);                   // data.num_accumulated after this batch
            args.pixels[local_index] = vec4_to_uint(acc.x / n, acc.y / n, acc.z / n);
        } else {
            args.pixels[local_index] = vec4_to_uint(last.x, last.y, last.z);
        }
    }
    wave_add_f64(&args.counters->total_energy, energy_sum);
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
struct WfHost {
    WfDev dev{};
    uint32_t alloc_cap = 0;
    uint32_t alloc_rounds = 0;
    uint32_t n_cus = 0;
};

static void WfRelease(WfHost* h)
{
    for (int b = 0; b < 2; ++b) { (void)hipFree(h->dev.A[b]); (void)hipFree(h->dev.B[b]); (void)hipFree(h->dev.C[b]); h->dev.A[b] = h->dev.B[b] = h->dev.C[b] = nullptr; }
    (void)hipFree(h->dev.st_tp); (void)hipFree(h->dev.st_en); (void)hipFree(h->dev.counts);
    h->dev.st_tp = h->dev.st_en = nullptr; h->dev.counts = nullptr;
    h->alloc_cap = 0; h->alloc_rounds = 0;
}

void WavefrontFree(void* state)
{
    if (!state) return;
    WfHost* h = static_cast<WfHost*>(state);
    WfRelease(h);
    delete h;
}

static constexpr uint32_t kMaxPoolPaths = 32u << 20;     // 32 Mi paths * 224 B = 7.5 GB of queues + state
static constexpr uint32_t kMaxBatchSamples = 16;

int LaunchWavefront(cgpt_ctx* ctx, const DevRenderArgs& args_in, bool count)
{
    hipStream_t stream = CtxStream(ctx);
    void** slot = CtxWavefrontSlot(ctx);
    if (!*slot) {
        *slot = new (std::nothrow) WfHost;
        if (!*slot) { CtxFail(ctx, CGPT_ERR_INVALID, "out of host memory"); return -1; }
    }
    WfHost* h = static_cast<WfHost*>(*slot);

    const uint32_t rows = args_in.row_end - args_in.row_begin;
    const uint32_t tiles_x = (args_in.width + 7u) / 8u, tiles_y = (rows + 7u) / 8u;
    const uint64_t n_pixels64 = (uint64_t)tiles_x * tiles_y * 64u;
    if (n_pixels64 > kMaxPoolPaths) { CtxFail(ctx, CGPT_ERR_UNSUPPORTED, "band of %llu pixels exceeds the wavefront pool", (unsigned long long)n_pixels64); return -1; }
    const uint32_t n_pixels = (uint32_t)n_pixels64;
    const uint32_t batch = std::max(1u, std::min({ kMaxBatchSamples, args_in.n_samples, kMaxPoolPaths / n_pixels }));
    const uint32_t cap = n_pixels * batch;
    const uint32_t rounds = (uint32_t)args_in.settings.max_ray_depth + 2u;    // extend rounds 0..max_depth, + the trailing shadow rays

#define WF_TRY(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) { CtxFail(ctx, CGPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); return -1; } \
    } while (0)

    if (h->alloc_cap < cap || h->alloc_rounds < rounds + 1u) {
        WF_TRY(hipStreamSynchronize(stream));
        WfRelease(h);
        const size_t q = 2 * (size_t)cap * sizeof(float4);
        for (int b = 0; b < 2; ++b) {
            WF_TRY(hipMalloc((void**)&h->dev.A[b], q)); WF_TRY(hipMalloc((void**)&h->dev.B[b], q)); WF_TRY(hipMalloc((void**)&h->dev.C[b], q));
        }
        WF_TRY(hipMalloc((void**)&h->dev.st_tp, (size_t)cap * sizeof(float4)));
        WF_TRY(hipMalloc((void**)&h->dev.st_en, (size_t)cap * sizeof(float4)));
        WF_TRY(hipMalloc((void**)&h->dev.counts, (size_t)(rounds + 1u) * kCountStride * sizeof(uint32_t)));
        h->alloc_cap = cap; h->alloc_rounds = rounds + 1u;
    }
    WfDev wf = h->dev;
    wf.cap = h->alloc_cap; wf.n_pixels = n_pixels; wf.tiles_x = tiles_x;

    if (h->n_cus == 0) {
        int n_dev = 0, cus = 0;
        WF_TRY(hipGetDevice(&n_dev));
        WF_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, n_dev));
        h->n_cus = (uint32_t)cus;
    }
    const uint32_t n_cus = h->n_cus;
    const size_t lds = (size_t)args_in.scene.stack_depth * 256 * sizeof(uint32_t);
    const dim3 block(256);
    const dim3 persistent_grid(n_cus * 8u), stream_grid(n_cus * 8u);

    int launches = 0;
    DevRenderArgs args = args_in;
    for (uint32_t done = 0; done < args_in.n_samples; done += batch) {
        const uint32_t bn = std::min(batch, args_in.n_samples - done);
        const uint32_t bfirst = args_in.first_sample + done;
        WF_TRY(hipMemsetAsync(wf.counts, 0, (size_t)(rounds + 1u) * kCountStride * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(wf_generate, stream_grid, block, 0, stream, args, wf, bfirst, bn);
        ++launches;
        uint32_t buf = 0;
        for (uint32_t r = 0; r < rounds; ++r) {
            if (count) hipLaunchKernelGGL(wf_trace<true>, persistent_grid, block, lds, stream, args.scene, wf, r, buf, args.counters);
            else hipLaunchKernelGGL(wf_trace<false>, persistent_grid, block, lds, stream, args.scene, wf, r, buf, args.counters);
            ++launches;
            if (r + 1u < rounds) {
                if (count) hipLaunchKernelGGL(wf_shade<true>, stream_grid, block, 0, stream, args, wf, r, buf);
                else hipLaunchKernelGGL(wf_shade<false>, stream_grid, block, 0, stream, args, wf, r, buf);
                ++launches;
                buf ^= 1u;
            }
        }
        hipLaunchKernelGGL(wf_accumulate, dim3((n_pixels + 255u) / 256u), block, 0, stream, args, wf, bfirst, bn);
        ++launches;
        WF_TRY(hipGetLastError());
    }
#undef WF_TRY
    return launches;
}

}  // namespace cgpt
