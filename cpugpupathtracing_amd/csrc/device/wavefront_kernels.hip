// wavefront_kernels.hip -- the wavefront pipeline (gfx950): generate -> per bounce [trace -> shade] -> accumulate.
//
// Why: in the megakernel a wave's traversal loop runs until its slowest lane is done (max-vs-mean ray length) and a tile
// runs until its most expensive pixel is done; PMC showed ~10 % active lanes per VALU instruction.  Here
//   * trace<>  is a PERSISTENT kernel: every wave keeps its 64 lanes filled from the ray slots of the pool, each lane runs
//     the reference's ordered stack traversal (ref: Source/BVH.cpp:61-127) on a per-wavefront LDS stack, and a lane that
//     finishes its ray is refilled while its neighbours keep going.  Extend rays and NEE shadow rays are traced by the
//     same kernel (both are closest-hit IntersectScene calls, ref: Source/Main.cpp:299-316,452-453); a shadow ray's
//     epilogue adds its pending contribution to the path's energy.
//   * shade<>  runs shade_bounce() (ref: Main.cpp:404-573) for every live extend hit and rewrites the path's slot with its
//     next ray (and its shadow slot with the NEE connection).
//   * accumulate adds the finished samples to the float4 accumulator IN SAMPLE ORDER, so the image is bit-identical to
//     the megakernel's and the oracle's (ref: Main.cpp:735-746).
// Divergence control without global atomics: ray slots are addressed by path id (extend slot = pid, shadow slot = cap + pid)
// with a one-byte liveness flag per slot.  Waves own chunks of 256 slots (static, strided over the persistent grid), read
// the 256 flags with one dword per lane, and COMPACT the live slot ids with __ballot + mbcnt into a small per-wave LDS
// ring; idle lanes (trace) or groups of 64 (shade) are fed from the ring.  A first version compacted through global
// atomic counters (one per wave): 10.6 M waves/step serialised on one L2 word at ~88 atomics/us and cost more than the tracing.
// Slot entry (48 B, three float4 planes): A = {o.xyz, t}  B = {d.xyz, -}  C = extend: {bits obj, tri, bvh_depth, -} (in: initial
// payload, out: hit record) | shadow: {pending.xyz, -}.
// Path state (32 B per path, path id = sample_in_batch * n_pixels + pixel): {throughput.xyz, bits(depth | spec << 8)}, {energy.xyz, bits(rng)}.
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

extern __shared__ uint32_t lds_dyn[];

static constexpr uint32_t kStartObject = 0xFFFFFFFFu;   // traversal code: "begin the next object of the scene"
static constexpr uint32_t kChunk = 256;                  // slots per chunk: one flag dword per lane
static constexpr uint32_t kRing = 320;                   // per-wave LDS ring of live slot ids: up to 63 left over + 256 new
static constexpr uint32_t kRefillIdleLanes = 16;         // leave the traversal loop to refill once this many lanes are idle

struct WfDev {
    float4* A; float4* B; float4* C;   // 2 * cap slots each: [0, cap) extend, [cap, 2 cap) shadow
    float4* st_tp; float4* st_en;      // cap paths
    uint8_t* live;                     // 2 * cap flags
    uint32_t cap;                      // paths in the pool (multiple of kChunk)
    uint32_t n_pixels;                 // padded pixel count of the band (8x8 tiles)
    uint32_t tiles_x;                  // 8x8 tiles per row
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t rank_in_mask(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// pixel of padded index p: 8x8 tiles in row-major tile order (a wave of consecutive p covers one tile)
__device__ __forceinline__ bool pixel_of(const DevRenderArgs& a, const WfDev& wf, uint32_t p, uint32_t& px, uint32_t& py)
{
    const uint32_t tile = p >> 6, l = p & 63u;
    px = (tile % wf.tiles_x) * 8u + (l & 7u);
    py = a.row_begin + (tile / wf.tiles_x) * 8u + (l >> 3);
    return px < a.width && py < a.row_end;
}

// Appends the live slots of chunk `chunk` (slots chunk*256 .. +255) to the wave's LDS ring.  One flag dword per lane, four
// __ballot + mbcnt compactions.  Returns the new ring count (wave-uniform).
__device__ __forceinline__ uint32_t scan_chunk(const WfDev& wf, uint32_t chunk, uint32_t* ring, uint32_t count)
{
    const uint32_t flags = reinterpret_cast<const uint32_t*>(wf.live)[(size_t)chunk * 64u + lane_id()];
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        const bool on = ((flags >> (8u * j)) & 0xFFu) != 0u;
        const unsigned long long m = __ballot(on);
        if (on) ring[count + rank_in_mask(m)] = chunk * kChunk + lane_id() * 4u + j;
        count += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_wave_barrier();     // ring writes above are read by other lanes of this wave below
    return count;
}

// ---- K1 generate: primary rays of one batch of samples (ref: Main.cpp:713-716, Camera::GetRay :133-140) ----------------
__global__ void __launch_bounds__(256) wf_generate(const DevRenderArgs args, const WfDev wf, uint32_t batch_first, uint32_t batch_n)
{
    const uint32_t n_paths = wf.n_pixels * batch_n;
    for (uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x; pid < wf.cap; pid += gridDim.x * blockDim.x) {
        uint32_t px = 0, py = 0;
        const bool valid = pid < n_paths && pixel_of(args, wf, pid % wf.n_pixels, px, py);
        wf.live[pid] = valid ? 1u : 0u;
        if (valid) {
            const uint32_t s = batch_first + pid / wf.n_pixels;
            const uint32_t rng = pcg_seed(py * args.width + px, s, args.seed);
            const Ray ray = camera_ray(args.camera, (float)px * (1.0f / (float)args.width), (float)py * (1.0f / (float)args.height));
            float4 a, b, c;
            a.x = ray.o.x; a.y = ray.o.y; a.z = ray.o.z; a.w = ray.t;
            b.x = ray.d.x; b.y = ray.d.y; b.z = ray.d.z; b.w = 0.0f;
            c.x = __uint_as_float(kNoHit); c.y = __uint_as_float(0u); c.z = __uint_as_float(0u); c.w = 0.0f;
            wf.A[pid] = a; wf.B[pid] = b; wf.C[pid] = c;
            float4 tp, en;
            tp.x = 1.0f; tp.y = 1.0f; tp.z = 1.0f; tp.w = __uint_as_float(0u);
            en.x = 0.0f; en.y = 0.0f; en.z = 0.0f; en.w = __uint_as_float(rng);
            wf.st_tp[pid] = tp; wf.st_en[pid] = en;
        }
    }
}

// ---- K2/K4 trace: persistent closest-hit traversal with per-lane refill ------------------------------------------------
// Scans slots [slot_begin, slot_end) (multiples of kChunk).  LDS: traversal stacks (stack_depth x 256 dwords), then one
// ring of kRing dwords per wave.
template <bool COUNT>
__global__ void __launch_bounds__(256) wf_trace(const DevScene sc, const WfDev wf, uint32_t slot_begin, uint32_t slot_end, DevCounters* counters)
{
    uint32_t* const stack = lds_dyn + threadIdx.x;
    const uint32_t stride = blockDim.x;
    uint32_t* const ring = lds_dyn + sc.stack_depth * 256u + (threadIdx.x >> 6) * kRing;

    const uint32_t n_waves = gridDim.x * 4u;
    uint32_t chunk = slot_begin / kChunk + blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t chunk_end = slot_end / kChunk;
    uint32_t ring_count = 0;

    bool has_ray = false;
    V3 o = mk(0.0f), d = mk(0.0f), inv = mk(0.0f);
    float t = 0.0f;
    uint32_t obj = kNoHit, tri = 0, depth = 0, cur_obj = 0, code = kStartObject, sp = 0, slot = 0;
    Counters cnt = { 0, 0, 0, 0, 0 };

    for (;;) {
        // ---- refill idle lanes from the ring; top the ring up from this wave's next chunks ----
        const unsigned long long need = __ballot(!has_ray);
        const uint32_t n_need = (uint32_t)__popcll(need);
        while (ring_count < n_need && chunk < chunk_end) {
            ring_count = scan_chunk(wf, chunk, ring, ring_count);
            chunk += n_waves;
        }
        if (n_need && ring_count) {
            const uint32_t take = min(n_need, ring_count);
            const uint32_t rank = rank_in_mask(need);
            if (!has_ray && rank < take) {
                slot = ring[ring_count - 1u - rank];
                const float4 a = wf.A[slot], b = wf.B[slot];
                o = mk(a.x, a.y, a.z); t = a.w; d = mk(b.x, b.y, b.z);
                inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);                 // Ray ctor, ref: Primitives.h:64
                if (slot >= wf.cap) { obj = kNoHit; tri = 0; depth = 0; }     // shadow ray, ref: Main.cpp:452
                else { const float4 c = wf.C[slot]; obj = __float_as_uint(c.x); tri = __float_as_uint(c.y); depth = __float_as_uint(c.z); }
                cur_obj = 0; code = kStartObject; sp = 0; has_ray = true;
                cnt.rays++;
            }
            __builtin_amdgcn_wave_barrier();
            ring_count -= take;
        }
        if (__ballot(has_ray) == 0ull) break;                                 // ring and chunks are empty too (loop above)
        const bool can_refill = ring_count != 0u || chunk < chunk_end;

        // ---- traversal until enough lanes are idle ----
        for (;;) {
            // begin the next object / finish the ray (IntersectScene's object loop, ref: Main.cpp:303-315)
            while (has_ray && code == kStartObject) {
                if (cur_obj >= sc.n_objects) {
                    if (slot >= wf.cap) {                                     // connect epilogue, ref: Main.cpp:454-463
                        if (obj == kNoHit) {
                            const float4 pe = wf.C[slot];
                            const uint32_t pid = slot - wf.cap;
                            float4 en = wf.st_en[pid];
                            en.x += pe.x; en.y += pe.y; en.z += pe.z;
                            wf.st_en[pid] = en;
                        }
                        wf.live[slot] = 0u;
                    } else {
                        reinterpret_cast<float*>(&wf.A[slot])[3] = t;
                        float4 c; c.x = __uint_as_float(obj); c.y = __uint_as_float(tri); c.z = __uint_as_float(depth); c.w = 0.0f;
                        wf.C[slot] = c;
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

// ---- K3 shade: one bounce per live extend slot, 64 compacted slots at a time ---------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(256) wf_shade(const DevRenderArgs args, const WfDev wf)
{
    const DevScene& sc = args.scene;
    uint32_t* const ring = lds_dyn + (threadIdx.x >> 6) * kRing;
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t chunk_end = wf.cap / kChunk;
    uint32_t ring_count = 0;
    Counters cnt = { 0, 0, 0, 0, 0 };

    uint32_t chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
    for (;;) {
        while (ring_count < 64u && chunk < chunk_end) {
            ring_count = scan_chunk(wf, chunk, ring, ring_count);
            chunk += n_waves;
        }
        if (ring_count == 0u) break;
        const uint32_t take = min(64u, ring_count);
        if (lane_id() < take) {
            const uint32_t pid = ring[ring_count - 1u - lane_id()];
            const float4 a = wf.A[pid], b = wf.B[pid], c = wf.C[pid];
            Ray ray, shadow;
            ray.o = mk(a.x, a.y, a.z); ray.t = a.w; ray.d = mk(b.x, b.y, b.z);
            ray.obj = __float_as_uint(c.x); ray.tri = __float_as_uint(c.y); ray.bvh_depth = __float_as_uint(c.z);
            shadow = ray;
            V3 pending = mk(0.0f);
            const float4 tp = wf.st_tp[pid], en = wf.st_en[pid];
            PathState ps;
            ps.throughput = mk(tp.x, tp.y, tp.z); ps.energy = mk(en.x, en.y, en.z);
            ps.rng = __float_as_uint(en.w);
            const uint32_t fl = __float_as_uint(tp.w);
            ps.depth = fl & 0xFFu; ps.is_specular = (fl & 0x100u) != 0u;

            const uint32_t flags = shade_bounce<COUNT>(sc, args.settings, ray, ps, shadow, pending, cnt);

            float4 tpo, eno;
            tpo.x = ps.throughput.x; tpo.y = ps.throughput.y; tpo.z = ps.throughput.z;
            tpo.w = __uint_as_float((ps.depth & 0xFFu) | (ps.is_specular ? 0x100u : 0u));
            eno.x = ps.energy.x; eno.y = ps.energy.y; eno.z = ps.energy.z; eno.w = __uint_as_float(ps.rng);
            wf.st_tp[pid] = tpo; wf.st_en[pid] = eno;

            if ((flags & kBounceTerminate) == 0u) {                           // next extend ray, same slot
                float4 na, nb, nc;
                na.x = ray.o.x; na.y = ray.o.y; na.z = ray.o.z; na.w = ray.t;
                nb.x = ray.d.x; nb.y = ray.d.y; nb.z = ray.d.z; nb.w = 0.0f;
                nc.x = __uint_as_float(ray.obj); nc.y = __uint_as_float(ray.tri); nc.z = __uint_as_float(ray.bvh_depth); nc.w = 0.0f;
                wf.A[pid] = na; wf.B[pid] = nb; wf.C[pid] = nc;
            } else {
                wf.live[pid] = 0u;
            }
            if (flags & kBounceShadow) {                                      // NEE connection, slot cap + pid
                const uint32_t ss = wf.cap + pid;
                float4 sa, sb, scc;
                sa.x = shadow.o.x; sa.y = shadow.o.y; sa.z = shadow.o.z; sa.w = shadow.t;
                sb.x = shadow.d.x; sb.y = shadow.d.y; sb.z = shadow.d.z; sb.w = 0.0f;
                scc.x = pending.x; scc.y = pending.y; scc.z = pending.z; scc.w = 0.0f;
                wf.A[ss] = sa; wf.B[ss] = sb; wf.C[ss] = scc;
                wf.live[ss] = 1u;
            }
        }
        __builtin_amdgcn_wave_barrier();
        ring_count -= take;
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
// Batches of samples are independent until the final accumulate, and every bounce round ends in a tail where a few
// long rays keep a handful of waves busy.  kPools batches are therefore in flight at once, each with its own slot pool on
// its own HIP stream, so one batch's tail overlaps another batch's bulk; the accumulate kernels are chained with events so
// samples are still added in order.
static constexpr uint32_t kPools = 4;
static constexpr uint32_t kMaxPoolPaths = 32u << 20;     // 32 Mi paths * 130 B = 4.4 GB of slots + state per pool
static constexpr uint32_t kMaxBatchSamples = 16;

struct WfHost {
    WfDev dev[kPools] = {};
    hipStream_t streams[kPools] = {};
    hipEvent_t acc_done[kPools] = {};
    hipEvent_t begin = nullptr;
    uint32_t alloc_cap = 0;
    uint32_t n_cus = 0;
    uint32_t trace_blocks_per_cu[2] = { 0, 0 }, shade_blocks_per_cu[2] = { 0, 0 };   // [COUNT]
    size_t occupancy_lds = 0;
};

static void WfRelease(WfHost* h)
{
    for (uint32_t p = 0; p < kPools; ++p) {
        WfDev& d = h->dev[p];
        (void)hipFree(d.A); (void)hipFree(d.B); (void)hipFree(d.C);
        (void)hipFree(d.st_tp); (void)hipFree(d.st_en); (void)hipFree(d.live);
        d = WfDev{};
    }
    h->alloc_cap = 0;
}

void WavefrontFree(void* state)
{
    if (!state) return;
    WfHost* h = static_cast<WfHost*>(state);
    WfRelease(h);
    for (uint32_t p = 0; p < kPools; ++p) {
        if (h->streams[p]) (void)hipStreamDestroy(h->streams[p]);
        if (h->acc_done[p]) (void)hipEventDestroy(h->acc_done[p]);
    }
    if (h->begin) (void)hipEventDestroy(h->begin);
    delete h;
}

int LaunchWavefront(cgpt_ctx* ctx, const DevRenderArgs& args_in, bool count)
{
    hipStream_t stream = CtxStream(ctx);
    void** slot = CtxWavefrontSlot(ctx);

#define WF_TRY(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) { CtxFail(ctx, CGPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); return -1; } \
    } while (0)

    if (!*slot) {
        WfHost* fresh = new (std::nothrow) WfHost;
        if (!fresh) { CtxFail(ctx, CGPT_ERR_INVALID, "out of host memory"); return -1; }
        *slot = fresh;
        for (uint32_t p = 0; p < kPools; ++p) {
            WF_TRY(hipStreamCreateWithFlags(&fresh->streams[p], hipStreamNonBlocking));
            WF_TRY(hipEventCreateWithFlags(&fresh->acc_done[p], hipEventDisableTiming));
        }
        WF_TRY(hipEventCreateWithFlags(&fresh->begin, hipEventDisableTiming));
    }
    WfHost* h = static_cast<WfHost*>(*slot);

    const uint32_t rows = args_in.row_end - args_in.row_begin;
    const uint32_t tiles_x = (args_in.width + 7u) / 8u, tiles_y = (rows + 7u) / 8u;
    const uint64_t n_pixels64 = (uint64_t)tiles_x * tiles_y * 64u;
    if (n_pixels64 > kMaxPoolPaths) { CtxFail(ctx, CGPT_ERR_UNSUPPORTED, "band of %llu pixels exceeds the wavefront pool", (unsigned long long)n_pixels64); return -1; }
    const uint32_t n_pixels = (uint32_t)n_pixels64;
    const uint32_t batch = std::max(1u, std::min({ kMaxBatchSamples, (args_in.n_samples + kPools - 1u) / kPools, kMaxPoolPaths / n_pixels }));
    const uint32_t cap = (n_pixels * batch + kChunk - 1u) / kChunk * kChunk;
    const uint32_t rounds = (uint32_t)args_in.settings.max_ray_depth + 2u;    // extend rounds 0..max_depth, + the trailing shadow rays

    if (h->alloc_cap < cap) {
        WF_TRY(hipDeviceSynchronize());
        WfRelease(h);
        const size_t q = 2 * (size_t)cap * sizeof(float4);
        for (uint32_t p = 0; p < kPools; ++p) {
            WfDev& d = h->dev[p];
            WF_TRY(hipMalloc((void**)&d.A, q)); WF_TRY(hipMalloc((void**)&d.B, q)); WF_TRY(hipMalloc((void**)&d.C, q));
            WF_TRY(hipMalloc((void**)&d.st_tp, (size_t)cap * sizeof(float4)));
            WF_TRY(hipMalloc((void**)&d.st_en, (size_t)cap * sizeof(float4)));
            WF_TRY(hipMalloc((void**)&d.live, 2 * (size_t)cap));
        }
        h->alloc_cap = cap;
    }
    if (h->n_cus == 0) {
        int n_dev = 0, cus = 0;
        WF_TRY(hipGetDevice(&n_dev));
        WF_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, n_dev));
        h->n_cus = (uint32_t)cus;
    }
    const uint32_t n_cus = h->n_cus;
    const size_t trace_lds = ((size_t)args_in.scene.stack_depth * 256 + 4 * kRing) * sizeof(uint32_t);
    const size_t shade_lds = 4 * kRing * sizeof(uint32_t);
    // persistent grids = the resident capacity of the chip for each kernel
    if (h->occupancy_lds != trace_lds) {
        int b = 0;
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, wf_trace<false>, 256, trace_lds)); h->trace_blocks_per_cu[0] = (uint32_t)std::max(1, b);
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, wf_trace<true>, 256, trace_lds)); h->trace_blocks_per_cu[1] = (uint32_t)std::max(1, b);
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, wf_shade<false>, 256, shade_lds)); h->shade_blocks_per_cu[0] = (uint32_t)std::max(1, b);
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, wf_shade<true>, 256, shade_lds)); h->shade_blocks_per_cu[1] = (uint32_t)std::max(1, b);
        h->occupancy_lds = trace_lds;
    }
    const dim3 block(256);
    const dim3 trace_grid(n_cus * h->trace_blocks_per_cu[count ? 1 : 0]), shade_grid(n_cus * h->shade_blocks_per_cu[count ? 1 : 0]);
    const dim3 stream_grid(n_cus * 8u);

    // the pool streams start after whatever the caller queued on the context's stream
    WF_TRY(hipEventRecord(h->begin, stream));
    for (uint32_t p = 0; p < kPools; ++p) WF_TRY(hipStreamWaitEvent(h->streams[p], h->begin, 0));

    int launches = 0;
    DevRenderArgs args = args_in;
    uint32_t k = 0;
    for (uint32_t done = 0; done < args_in.n_samples; done += batch, ++k) {
        const uint32_t p = k % kPools;
        hipStream_t st = h->streams[p];
        WfDev wf = h->dev[p];
        wf.cap = cap; wf.n_pixels = n_pixels; wf.tiles_x = tiles_x;
        const uint32_t bn = std::min(batch, args_in.n_samples - done);
        const uint32_t bfirst = args_in.first_sample + done;
        // shadow flags start clear; every shadow slot is cleared again by the trace that consumes it
        if (k < kPools) WF_TRY(hipMemsetAsync(wf.live + cap, 0, cap, st));
        hipLaunchKernelGGL(wf_generate, stream_grid, block, 0, st, args, wf, bfirst, bn);
        ++launches;
        for (uint32_t r = 0; r < rounds; ++r) {
            // round 0 has no shadow rays yet; the last round has only shadow rays left
            const uint32_t s0 = r + 1u == rounds ? cap : 0u, s1 = r == 0u ? cap : 2u * cap;
            if (count) hipLaunchKernelGGL(wf_trace<true>, trace_grid, block, trace_lds, st, args.scene, wf, s0, s1, args.counters);
            else hipLaunchKernelGGL(wf_trace<false>, trace_grid, block, trace_lds, st, args.scene, wf, s0, s1, args.counters);
            ++launches;
            if (r + 1u < rounds) {
                if (count) hipLaunchKernelGGL(wf_shade<true>, shade_grid, block, shade_lds, st, args, wf);
                else hipLaunchKernelGGL(wf_shade<false>, shade_grid, block, shade_lds, st, args, wf);
                ++launches;
            }
        }
        // accumulate in sample order: batch k after batch k-1
        if (k > 0) WF_TRY(hipStreamWaitEvent(st, h->acc_done[(k - 1u) % kPools], 0));
        hipLaunchKernelGGL(wf_accumulate, dim3((n_pixels + 255u) / 256u), block, 0, st, args, wf, bfirst, bn);
        ++launches;
        WF_TRY(hipEventRecord(h->acc_done[p], st));
        WF_TRY(hipGetLastError());
    }
    // the context's stream continues after the last accumulate (which transitively follows all the others)
    if (k > 0) WF_TRY(hipStreamWaitEvent(stream, h->acc_done[(k - 1u) % kPools], 0));
#undef WF_TRY
    return launches;
}

}  // namespace cgpt
