// persistent_kernel.hip -- whole paths in ONE persistent launch (gfx950): K6 of SURVEY 2.2 done the wavefront way.
//
// The wavefront pipeline (wavefront_kernels.hip) synchronises every bounce round through HBM: per round a trace launch, a shade
// launch and two list kernels, ~150 bytes of ray / path state streamed per path, and a drain at the end of every launch in which a
// few long rays keep a handful of waves busy (a launch never takes less than ~0.2 ms, so a 1-sample frame of seven rounds costs
// 2.6 ms however few rays it has).  Here a lane owns a PATH: the same voted traversal steps (trace_steps.hpp) trace its rays,
// a fourth voted state runs shade_bounce() (ref: Main.cpp:404-573) on the hit -- same device function as the other two render
// paths, so the image is bit-identical -- and the lane goes straight on to the NEE shadow ray and the next extend ray without
// leaving the kernel.  A lane whose path ends takes the next path id from the launch's work counters (trace_steps.hpp:
// WorkFetch).  What reaches HBM is 16 bytes per path: the path's radiance, which pt_accumulate adds to the float4 accumulator in
// sample order (ref: Main.cpp:735-746).  Measured against the other two paths: DESIGN.md 5.2, profiles/r02/.
// Rays that miss everything are retired where the miss is found (no shade step), brute-force / comparison paths
// (TracePath, ref: Main.cpp:581-689) keep their per-level operations in a per-lane HBM stack and apply them innermost-first.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "cpugpupt_abi.h"
#include "device_scene.h"
#include "fast_div.h"
#include "rt_device.hpp"
#include "shade_device.hpp"
#include "trace_steps.hpp"
#include "accumulate.hpp"

namespace cgpt {

using namespace dev;

hipStream_t CtxStream(cgpt_ctx* ctx);
void** CtxPersistentSlot(cgpt_ctx* ctx);
hipEvent_t CtxStartEvent(cgpt_ctx* ctx);
void PersistentFree(void* state);
int CtxFail(cgpt_ctx* ctx, int code, const char* fmt, ...);

extern __shared__ uint32_t pt_lds[];

static constexpr uint32_t kShade = 0x40000002u;          // traversal code: the extend ray is done, shade its hit

struct PtDev {
    float4* st_en;                     // [n_paths] {energy.xyz, bits(final depth)}: the finished paths' radiance
    float4* brute;                     // [level][thread][2] BruteLevel records of the brute-force paths (BRUTE kernels only)
    uint32_t* stack_overflow;          // [level - kLdsStackLevels][thread]: the rarely used deep end of the traversal stack
    unsigned long long* phase_stats;   // COUNT kernels only: wave / lane steps per state
    uint32_t n_paths;                  // path ids 0 .. n_paths-1 of this batch
    PathGrid g;
    uint32_t* work;                    // the launch's kWorkCounters work counters, 32 bytes apart (zeroed before the launch)
    uint32_t coarse;                   // path ids a wave takes per fetch while plenty are left
    uint32_t fine_below;               // once a counter has fewer ids than this left, a wave takes only as many as it has idle lanes
    uint32_t shade_shift;              // lanes waiting to shade count 2^shift times in the vote
};

enum : uint32_t {                      // per-lane path flags
    kPfDepthMask = 0xFFu,              // TracePathAdvanced's ray_depth (ref: Main.cpp:401)
    kPfSpecular = 0x100u,              // is_specular_ray (ref: Main.cpp:402)
    kPfShadow = 0x200u,                // the ray in flight is the NEE shadow ray
    kPfDead = 0x400u,                  // the path ends once the shadow ray in flight is resolved
    kPfBrute = 0x800u                  // this path runs TracePath (brute force)
};

#ifndef CGPT_PT_WAVES_PER_SIMD
#define CGPT_PT_WAVES_PER_SIMD 1
#endif
// TAIL: the instantiation for small calls (few samples per call -- the reference's own main loop renders ONE per Render(), ref:
// Main.cpp:702,825-942), which are mostly drain: once nothing is left to fetch, a wave with few busy lanes runs their rays in the lean
// per-lane loop (trace_steps.hpp: lean_traverse) instead of voted steps, because the call ends when its longest chain does (1080p, one
// sample: 2.47 -> 2.21 ms).  Kept out of the throughput instantiations, which it costs registers and SGPR spills (profiles/r03/one_sample.md).
template <bool COUNT, bool BRUTE, bool TAIL>
__global__ void __launch_bounds__(kTraceBlock, (!COUNT && !BRUTE && !TAIL) ? CGPT_PT_WAVES_PER_SIMD : 1) pt_persistent(const DevRenderArgs args, const PtDev pt, uint32_t batch_first, const TraceTune tune)
{
    const DevScene& sc = args.scene;
    const DevSettings& st = args.settings;
    const uint32_t grid_threads = gridDim.x * kTraceBlock;
    const TravCtx ctx = trav_setup(sc, pt_lds, tune.top_records, pt.stack_overflow, grid_threads, tune.lds_tris);
    const uint32_t tid = blockIdx.x * kTraceBlock + threadIdx.x;

    // Work distribution: path ids from the launch's work counters (trace_steps.hpp: WorkFetch) -- consecutive ids, i.e. with the
    // default pixel-major ids the samples of one pixel and of its neighbours in the 8x8 tile; coarse fetches first and one id per
    // idle lane near the end
    WorkFetch work = work_begin(blockIdx.x * (kTraceBlock / 64u) + (threadIdx.x >> 6));

    Trav r;
    r.d = mk(0.0f); r.rs = make_ray_slab(r.d, r.d); r.t = 0.0f;
    r.obj = kNoHit; r.tri = 0; r.depth = 0; r.cur_obj = 0; r.code = kIdle; r.sp = 0; r.fast_levels = kLdsStackLevels;
    // the path this lane owns
    uint32_t pid = 0, rng = 0, pf = 0;
    V3 tp = mk(0.0f), en = mk(0.0f), pending = mk(0.0f);
    V3 park_o = mk(0.0f), park_d = mk(0.0f);                                  // the next extend ray, parked while the shadow ray is traced
    float park_t = 0.0f;
    uint32_t park_obj = kNoHit, park_tri = 0, park_depth = 0;                 // its payload: a ray traced again after total internal reflection keeps its hit (SURVEY A-3)
    Counters cnt = { 0, 0, 0, 0, 0 };
    uint32_t ph[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };                           // COUNT only: wave steps inner / leaf / object / shade, lanes object / shade, votes, refills, lanes leaf
#ifdef CGPT_PHASE_CYCLES
    // diagnostic build (scripts/build_variant.sh cyc -DCGPT_PHASE_CYCLES): where a wave's cycles go, by phase
    constexpr bool kCyc = !COUNT;
    unsigned long long cy[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, cy_mark = 0, cy_start = 0, cy_last_work = 0;   // refill, inner, leaf, object, shade, lean
    uint32_t cn[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, cl[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (kCyc) cy_start = __builtin_readcyclecounter();
#define PT_CYC_BEGIN() do { if (kCyc) cy_mark = __builtin_readcyclecounter(); } while (0)
#define PT_CYC_END(i, lanes) do { if (kCyc) { cy[i] += __builtin_readcyclecounter() - cy_mark; cn[i]++; cl[i] += (lanes); } } while (0)
#else
#define PT_CYC_BEGIN() do { } while (0)
#define PT_CYC_END(i, lanes) do { } while (0)
#endif

    auto finish_path = [&](V3 energy) {                                       // ref: Main.cpp:575-578: the path's radiance leaves the kernel
        float4 o4; o4.x = energy.x; o4.y = energy.y; o4.z = energy.z; o4.w = __uint_as_float(pf & kPfDepthMask);
        st_stream(&pt.st_en[pid], o4);
        r.code = kIdle;
    };

    // a finished ray is dispatched on the spot: the lanes for which object_step() returned true
    auto ray_done = [&]() {
        if (pf & kPfShadow) {                                     // NEE connection resolved, ref: Main.cpp:454-463
            if (r.obj == kNoHit) en = en + pending;
            pf &= ~kPfShadow;
            if (pf & kPfDead) {
                finish_path(en);
            } else {
                trav_start(ctx, r, park_o, park_d, park_t, park_obj, park_tri, park_depth);
                cnt.rays++;
            }
        } else if (r.obj == kNoHit && !(BRUTE && (pf & kPfBrute)) && !(st.debug_mode == 2u && (pf & kPfDepthMask) == 0u)) {
            finish_path(en);                                      // the extend ray left the scene, ref: Main.cpp:415-416
        } else {
            r.code = kShade;
        }
    };
    // one bounce of the path on the hit of its extend ray: the lanes with r.code == kShade
    auto shade_hit = [&]() {
        Ray ray;
        ray.o = trav_origin(r); ray.d = r.d; ray.t = r.t; ray.obj = r.obj; ray.tri = r.tri; ray.bvh_depth = trav_depth(r);
        if (BRUTE && (pf & kPfBrute)) {
            // TracePath level (ref: Main.cpp:581-689): record this level's operation, go on with the child ray, or fold
            // the recorded chain over the leaf's radiance, innermost level first
            uint32_t depth = pf & kPfDepthMask;
            BruteLevel lv; V3 leaf = mk(0.0f);
            bool fold = brute_bounce<COUNT>(sc, st, ray, rng, depth, lv, leaf, cnt) == kBruteLeaf;
            if (!fold) {
                float4* rec = pt.brute + ((size_t)depth * grid_threads + tid) * 2u;
                float4 r0, r1;
                r0.x = __uint_as_float(lv.kind); r0.y = lv.a.x; r0.z = lv.a.y; r0.w = lv.a.z;
                r1.x = lv.cosi; r1.y = lv.absorb.x; r1.z = lv.absorb.y; r1.w = lv.absorb.z;
                rec[0] = r0; rec[1] = r1;
                depth++;
                pf = (pf & ~kPfDepthMask) | (depth & kPfDepthMask);
                if ((int32_t)depth > st.max_ray_depth) fold = true;   // the child returns black before tracing (ref: Main.cpp:589-590)
            }
            if (fold) {
                V3 L = leaf;
                for (uint32_t k = depth; k-- > 0u;) {
                    const float4* rec = pt.brute + ((size_t)k * grid_threads + tid) * 2u;
                    const float4 r0 = rec[0], r1 = rec[1];
                    BruteLevel b;
                    b.kind = __float_as_uint(r0.x); b.a = mk(r0.y, r0.z, r0.w); b.cosi = r1.x; b.absorb = mk(r1.y, r1.z, r1.w);
                    L = brute_apply(b, L);
                }
                finish_path(L);
            } else {
                trav_start(ctx, r, ray.o, ray.d, ray.t, ray.obj, ray.tri, ray.bvh_depth);
                cnt.rays++;
            }
        } else {
            PathState ps;
            ps.throughput = tp; ps.energy = en; ps.rng = rng; ps.depth = pf & kPfDepthMask; ps.is_specular = (pf & kPfSpecular) != 0u;
            Ray shadow = ray;
            V3 pend = mk(0.0f);
            const uint32_t flags = shade_bounce<COUNT>(sc, st, ray, ps, shadow, pend, cnt);
            tp = ps.throughput; en = ps.energy; rng = ps.rng;
            pf = (ps.depth & kPfDepthMask) | (ps.is_specular ? kPfSpecular : 0u);
            const bool dead = (flags & kBounceTerminate) != 0u;
            if (flags & kBounceShadow) {                          // the shadow ray first: its contribution precedes the next bounce's
                pending = pend;
                pf |= kPfShadow | (dead ? kPfDead : 0u);
                park_o = ray.o; park_d = ray.d; park_t = ray.t; park_obj = ray.obj; park_tri = ray.tri; park_depth = ray.bvh_depth;
                trav_start(ctx, r, shadow.o, shadow.d, shadow.t, kNoHit, 0u, 0u);
                cnt.rays++;
            } else if (!dead) {
                trav_start(ctx, r, ray.o, ray.d, ray.t, ray.obj, ray.tri, ray.bvh_depth);
                cnt.rays++;
            } else {
                finish_path(en);
            }
        }
    };

    for (;;) {
        if (COUNT) ph[7]++;
        // ---- idle lanes take new paths: consecutive ids from the wave's fetched range ----
        const unsigned long long need = __builtin_amdgcn_ballot_w64(r.code == kIdle);
        uint32_t n_need = (uint32_t)__popcll(need);
        PT_CYC_BEGIN();
        if (n_need) {
            work_fetch(work, pt.work, pt.n_paths, pt.coarse, pt.fine_below, n_need);
            const uint32_t take = min(n_need, work.loc_end - work.loc_next);
            const uint32_t rank = rank_in_mask(need);
            if (r.code == kIdle && rank < take) {
                pid = work.loc_next + rank;
                Ray pr; uint32_t px = 0;
                if (primary_ray(args, pt.g, pid, batch_first, pr, rng, px)) {  // false: padding of an edge tile, the lane stays idle
                    tp = mk(1.0f); en = mk(0.0f); pf = 0u;                    // ref: Main.cpp:398-402
                    if (BRUTE && (st.render_mode == 1u || (st.render_mode == 0u && px < args.width / 2u))) pf = kPfBrute;   // ref: Main.cpp:719-729
                    trav_start(ctx, r, pr.o, pr.d, pr.t, kNoHit, 0u, 0u);
                    cnt.rays++;
                }
            }
            work.loc_next += take;
#ifdef CGPT_PHASE_CYCLES
            if (kCyc && take) cy_last_work = __builtin_readcyclecounter();
#endif
        }
        PT_CYC_END(0, n_need);
        // Done when nothing is in flight and nothing is left to fetch (nothing in flight alone is not enough: every id just handed out
        // may have been padding of an edge tile; the step loop below then falls straight through and the wave fetches on)
        if (__builtin_amdgcn_ballot_w64(r.code != kIdle) == 0ull && work.exhausted && work.loc_next == work.loc_end) break;
        const bool can_refill = !work.exhausted;

        // ---- run the most popular state's step until enough lanes are idle ----
        for (;;) {
            const uint32_t n_inner = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code < kStartObject));
            const uint32_t n_leaf = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((int32_t)r.code < 0));
            const uint32_t n_obj = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code == kStartObject));
            const uint32_t n_shade = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code == kShade));
            const uint32_t n_busy = n_inner + n_leaf + n_obj + n_shade;
            if (n_busy == 0u) break;
            if (can_refill && 64u - n_busy >= tune.refill_idle) break;
            const uint32_t w_obj = n_obj << tune.obj_shift, w_shade = n_shade << pt.shade_shift;
            if (COUNT) ph[6]++;
            // ---- the tail of the launch: a few rays left in this wave and no path to hand to the idle lanes: every lane runs its ray to
            //      the next object boundary in the lean loop (trace_steps.hpp: lean_traverse) -- the launch ends when its longest chain does
            if (TAIL && !can_refill && n_busy <= tune.tail_lanes && n_inner + n_leaf != 0u) {
                PT_CYC_BEGIN();
                if (r.code < kStartObject || (int32_t)r.code < 0) lean_traverse<COUNT, false>(ctx, r, cnt);
                PT_CYC_END(5, n_inner + n_leaf);
                continue;
            }

            if (n_inner >= n_leaf && n_inner >= w_obj && n_inner >= w_shade) {
                do {
                    if (COUNT) ph[0]++;
                    PT_CYC_BEGIN();
                    if (r.code < kStartObject) inner_step<COUNT>(ctx, r, cnt);
                    PT_CYC_END(1, n_inner);
                } while ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code < kStartObject)) >= tune.inner_repeat);
            } else if (n_leaf >= w_obj && n_leaf >= w_shade) {
                do {
                    if (COUNT) { ph[1]++; ph[8] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((int32_t)r.code < 0)); }
                    PT_CYC_BEGIN();
                    if ((int32_t)r.code < 0) leaf_step<COUNT, false>(ctx, r, cnt);
                    PT_CYC_END(2, n_leaf);
                } while ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64((int32_t)r.code < 0)) >= tune.leaf_repeat);
            } else if (w_obj >= w_shade) {
                // ---- object step; a finished ray is dispatched on the spot ----
                if (COUNT) { ph[2]++; ph[4] += n_obj; }
                PT_CYC_BEGIN();
                if (r.code == kStartObject && object_step<COUNT, false>(ctx, r, cnt)) ray_done();
                PT_CYC_END(3, n_obj);
            } else {
                // ---- shade step: one bounce of the path on the hit of its extend ray ----
                if (COUNT) { ph[3]++; ph[5] += n_shade; }
                PT_CYC_BEGIN();
                if (r.code == kShade) shade_hit();
                PT_CYC_END(4, n_shade);
            }
        }
    }

#ifdef CGPT_PHASE_CYCLES
    if (kCyc && pt.phase_stats && lane_id() == 0u) {
        const unsigned long long now = __builtin_readcyclecounter();
        atomicAdd(&pt.phase_stats[16], now - cy_start); atomicAdd(&pt.phase_stats[17], 1ull);
        atomicAdd(&pt.phase_stats[18], now - (cy_last_work ? cy_last_work : cy_start));       // cycles after the wave's last refill: its drain
        atomicMax(&pt.phase_stats[19], now - cy_start);
        for (int i = 0; i < 6; ++i) { atomicAdd(&pt.phase_stats[20 + i], cy[i]); atomicAdd(&pt.phase_stats[28 + i], (unsigned long long)cn[i]); atomicAdd(&pt.phase_stats[36 + i], (unsigned long long)cl[i]); }
    }
#endif
    wave_add_u64(&args.counters->traced_rays, cnt.rays);
    if (COUNT) {
        wave_add_u64(&args.counters->inner_steps, cnt.inner);
        wave_add_u64(&args.counters->tri_tests, cnt.tris);
        wave_add_u64(&args.counters->bvh_depth_sum, cnt.depth);
        wave_add_u64(&args.counters->closest_hits, cnt.hits);
        if (pt.phase_stats && lane_id() == 0u)
            for (int i = 0; i < 9; ++i) atomicAdd(&pt.phase_stats[i], (unsigned long long)ph[i]);
    }
}

// ---- accumulate + pack: the batch's samples in order (ref: Main.cpp:735-746, MathLib.h:144-152) ---------------------------------
__global__ void __launch_bounds__(256) pt_accumulate(const DevRenderArgs args, const float4* __restrict__ st_en, const PathGrid g, uint32_t batch_first, uint32_t batch_n)
{
    accumulate_batch(args, st_en, g, batch_first, batch_n);
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
struct PtTuning {
    uint32_t budget_gib = 24;     // HBM for the finished paths' radiance (16 bytes per path of a batch; also at most half of what is free)
    uint32_t max_paths_mi = 1536; // most paths per batch, in Mi (path ids are 32-bit)
    uint32_t refill_idle = 16, inner_repeat = 20, leaf_repeat = 4, obj_shift = 0, shade_shift = 0;
    uint32_t top_records = kLdsTopMax;
    uint32_t blocks_per_cu = 64;  // cap on resident blocks per CU (occupancy experiments)
    uint32_t streams = 2;         // batches in flight (the drain of one overlaps the start of the next)
    uint32_t path_order = 2;      // PathOrder of the path ids = the order work items are handed out (trace_steps.hpp PathGrid)
    uint32_t chunk = 0;           // 64-path tiles per coarse work-counter fetch (0 = auto)
    uint32_t lds_tris = 1;        // the small meshes' triangles (the ground quad) are read from an LDS copy
    uint32_t tail_samples = 8;    // calls of at most this many samples run the TAIL instantiation
    uint32_t tail_lanes = 16;     // small calls: with nothing left to fetch, a wave of at most this many busy lanes runs the lean per-lane loop (0 = never)
    uint32_t fine_rounds = 2;     // fine fetches (one id per idle lane) once fewer than this many ids per lane of the grid are left
};

struct PtHost {
    PtTuning tune;
    float4* st_en[2] = { nullptr, nullptr };
    size_t st_en_paths = 0;
    float4* brute = nullptr; size_t brute_floats4 = 0;
    uint32_t* overflow = nullptr; size_t overflow_words = 0;
    uint32_t* work_counters = nullptr; uint32_t n_work_counters = 0;         // one zeroed word per launch of a render
    unsigned long long* phase_stats = nullptr;
    hipStream_t streams[2] = { nullptr, nullptr };
    hipEvent_t begin = nullptr, acc_done[2] = { nullptr, nullptr };
    hipEvent_t* ev = nullptr; uint32_t ev_cap = 0, ev_used = 0;
    uint32_t n_cus = 0;
    uint32_t blocks_per_cu[2][2][2] = {};    // [COUNT][BRUTE][TAIL]
    size_t occupancy_lds = 0;
};

struct PtKnob { const char* name; uint32_t PtTuning::*field; uint32_t lo, hi; };
static const PtKnob kPtKnobs[] = {
    { "pt_budget_gib", &PtTuning::budget_gib, 1, 256 },   { "pt_max_paths_mi", &PtTuning::max_paths_mi, 1, 2047 },
    { "pt_refill", &PtTuning::refill_idle, 1, 64 },       { "pt_inner_repeat", &PtTuning::inner_repeat, 1, 65 },
    { "pt_leaf_repeat", &PtTuning::leaf_repeat, 1, 65 },  { "pt_obj_shift", &PtTuning::obj_shift, 0, 6 },
    { "pt_shade_shift", &PtTuning::shade_shift, 0, 6 },   { "pt_top_records", &PtTuning::top_records, 0, 4096 },
    { "pt_blocks", &PtTuning::blocks_per_cu, 1, 64 },     { "pt_streams", &PtTuning::streams, 1, 2 },
    { "pt_path_order", &PtTuning::path_order, 0, 2 },
    { "pt_lds_tris", &PtTuning::lds_tris, 0, 1 },            { "pt_tail_lanes", &PtTuning::tail_lanes, 0, 64 },
    { "pt_tail_samples", &PtTuning::tail_samples, 0, 4096 },
    { "pt_chunk", &PtTuning::chunk, 0, 4096 },            { "pt_fine_rounds", &PtTuning::fine_rounds, 0, 1024 },
};

static PtHost* PtGetHost(cgpt_ctx* ctx)
{
    void** slot = CtxPersistentSlot(ctx);
    if (*slot) return static_cast<PtHost*>(*slot);
    PtHost* h = new (std::nothrow) PtHost;
    if (!h) { CtxFail(ctx, CGPT_ERR_INVALID, "out of host memory"); return nullptr; }
    for (const PtKnob& k : kPtKnobs) {
        char env[64] = "CGPT_";
        size_t n = strlen(env);
        for (const char* c = k.name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
        env[n] = 0;
        const char* v = getenv(env);
        if (v && *v) h->tune.*(k.field) = (uint32_t)std::min<long>(std::max<long>(strtol(v, nullptr, 10), k.lo), k.hi);
    }
    hipError_t e = hipEventCreateWithFlags(&h->begin, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&h->streams[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->acc_done[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {                                                    // a half-built state is never left in the context
        CtxFail(ctx, CGPT_ERR_HIP, "persistent kernel streams: %s", hipGetErrorString(e));
        PersistentFree(h);
        return nullptr;
    }
    *slot = h;
    return h;
}

int PersistentSetTuning(cgpt_ctx* ctx, const char* name, uint32_t value, bool* known)
{
    *known = false;
    for (const PtKnob& k : kPtKnobs)
        if (strcmp(k.name, name) == 0) {
            *known = true;
            PtHost* h = PtGetHost(ctx);
            if (!h) return CGPT_ERR_HIP;
            if (value < k.lo || value > k.hi) return CtxFail(ctx, CGPT_ERR_INVALID, "tuning knob %s: %u outside [%u, %u]", name, value, k.lo, k.hi);
            h->tune.*(k.field) = value;
            return CGPT_OK;
        }
    return CGPT_OK;
}

void PersistentFree(void* state)
{
    if (!state) return;
    PtHost* h = static_cast<PtHost*>(state);
    (void)hipFree(h->st_en[0]); (void)hipFree(h->st_en[1]); (void)hipFree(h->brute); (void)hipFree(h->overflow); (void)hipFree(h->phase_stats); (void)hipFree(h->work_counters);
    for (int i = 0; i < 2; ++i) {
        if (h->streams[i]) (void)hipStreamDestroy(h->streams[i]);
        if (h->acc_done[i]) (void)hipEventDestroy(h->acc_done[i]);
    }
    if (h->begin) (void)hipEventDestroy(h->begin);
    for (uint32_t i = 0; i < h->ev_cap; ++i) (void)hipEventDestroy(h->ev[i]);
    free(h->ev);
    delete h;
}

void PersistentCollectTiming(void* state, double* ms, uint32_t* launches, uint32_t* waves_per_simd)
{
    *ms = 0.0; *launches = 0; *waves_per_simd = 0;
    if (!state) return;
    PtHost* h = static_cast<PtHost*>(state);
    for (uint32_t i = 0; i + 1u < h->ev_used; i += 2u) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1u]) == hipSuccess) { *ms += t; *launches += 1; }
    }
    h->ev_used = 0;
    *waves_per_simd = std::min(h->tune.blocks_per_cu, h->blocks_per_cu[0][0][0]) * (kTraceBlock / 256u);
}

int LaunchPersistent(cgpt_ctx* ctx, const DevRenderArgs& args_in, bool count)
{
    hipStream_t stream = CtxStream(ctx);
    PtHost* h = PtGetHost(ctx);
    if (!h) return -1;
#define PT_TRY(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) { CtxFail(ctx, CGPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); return -1; } \
    } while (0)

    const bool brute = args_in.settings.render_mode != 2u;
    if (h->n_cus == 0) {
        int dev = 0, cus = 0;
        PT_TRY(hipGetDevice(&dev));
        PT_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        h->n_cus = (uint32_t)cus;
    }
    const uint32_t top_records = std::min(h->tune.top_records, args_in.scene.n_top_records);
    const size_t lds = trace_lds_bytes(top_records);
    if (h->occupancy_lds != lds) {
        int b = 0;
#define PT_EACH_KERNEL(X) X(false, false, false) X(false, true, false) X(true, false, false) X(true, true, false) X(false, false, true) X(false, true, true) X(true, false, true) X(true, true, true)
        if (lds > 48u * 1024u) {                                              // more dynamic LDS than the default limit: opt in per kernel
#define PT_OPT_IN(C, B, O) PT_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pt_persistent<C, B, O>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            PT_EACH_KERNEL(PT_OPT_IN)
#undef PT_OPT_IN
        }
#define PT_OCCUPANCY(C, B, O) PT_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (pt_persistent<C, B, O>), kTraceBlock, lds)); h->blocks_per_cu[C][B][O] = (uint32_t)std::max(1, b);
        PT_EACH_KERNEL(PT_OCCUPANCY)
#undef PT_OCCUPANCY
        h->occupancy_lds = lds;
    }
    const bool tail = args_in.n_samples <= h->tune.tail_samples;              // a small call: mostly drain
    const uint32_t blocks_per_cu = std::min(h->tune.blocks_per_cu, h->blocks_per_cu[count ? 1 : 0][brute ? 1 : 0][tail ? 1 : 0]);
    // the resident capacity of the chip, or fewer blocks when there are fewer than 64 paths per wave (a small call ends sooner when
    // its paths are spread thin than when the tail of a launch waits for 4 096 waves to find out that there is nothing to do)
    const uint32_t n_tiles = ((args_in.width + 7u) / 8u) * ((args_in.n_rows + 7u) / 8u);
    const uint64_t paths_in_call = (uint64_t)n_tiles * 64u * args_in.n_samples;
    const uint32_t blocks_wanted = (uint32_t)std::min<uint64_t>(h->n_cus * blocks_per_cu, std::max<uint64_t>(1, paths_in_call / (16u * (kTraceBlock / 64u))));
    const dim3 grid(blocks_wanted), block(256), trace_block(kTraceBlock);
    uint32_t max_blocks = 1;
    for (int i = 0; i < 8; ++i) max_blocks = std::max(max_blocks, h->blocks_per_cu[i >> 2][(i >> 1) & 1][i & 1]);
    const uint32_t max_threads = h->n_cus * max_blocks * kTraceBlock;

    const uint32_t tiles_x = (args_in.width + 7u) / 8u;
    const uint64_t n_pixels64 = (uint64_t)n_tiles * 64u;
    const uint64_t max_paths = (uint64_t)h->tune.max_paths_mi << 20;
    if (n_pixels64 > max_paths) { CtxFail(ctx, CGPT_ERR_UNSUPPORTED, "band of %llu pixels exceeds the path-id range", (unsigned long long)n_pixels64); return -1; }
    const uint32_t n_pixels = (uint32_t)n_pixels64;

    // ---- samples per batch: as many as the radiance buffers hold; two buffers when there is more than one batch ----
    size_t free_b = 0, total_b = 0;
    PT_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t held = (h->st_en[0] ? h->st_en_paths * sizeof(float4) : 0) + (h->st_en[1] ? h->st_en_paths * sizeof(float4) : 0);
    const size_t budget = std::min<size_t>((size_t)h->tune.budget_gib << 30, (free_b + held) / 2);
    uint32_t batch = (uint32_t)std::min<uint64_t>(args_in.n_samples, std::min<uint64_t>(max_paths / n_pixels, budget / sizeof(float4) / n_pixels));
    if (batch == 0) { CtxFail(ctx, CGPT_ERR_HIP, "not enough free HBM for one sample of %u pixels", n_pixels); return -1; }
    uint32_t n_batches = (args_in.n_samples + batch - 1u) / batch;
    uint32_t n_streams = 1;
    if (n_batches > 1 && h->tune.streams > 1) {                               // two half-size buffers: batch k's drain overlaps batch k+1
        n_streams = 2;
        batch = std::max(1u, (uint32_t)std::min<uint64_t>(batch, budget / 2 / sizeof(float4) / n_pixels));
        n_batches = (args_in.n_samples + batch - 1u) / batch;
    }
    const size_t cap = (size_t)n_pixels * batch;
    if (h->st_en_paths < cap || (n_streams == 2 && !h->st_en[1])) {
        PT_TRY(hipDeviceSynchronize());
        (void)hipFree(h->st_en[0]); (void)hipFree(h->st_en[1]); h->st_en[0] = h->st_en[1] = nullptr; h->st_en_paths = 0;
        PT_TRY(hipMalloc((void**)&h->st_en[0], cap * sizeof(float4)));
        if (n_streams == 2) PT_TRY(hipMalloc((void**)&h->st_en[1], cap * sizeof(float4)));
        h->st_en_paths = cap;
    }
    const uint32_t deep_levels = args_in.scene.stack_depth > kLdsStackLevels ? args_in.scene.stack_depth - kLdsStackLevels : 0u;
    const size_t overflow_words = std::max<size_t>(1, (size_t)deep_levels * max_threads) * n_streams;
    if (h->overflow_words < overflow_words) {
        PT_TRY(hipDeviceSynchronize());
        (void)hipFree(h->overflow); h->overflow = nullptr;
        PT_TRY(hipMalloc((void**)&h->overflow, overflow_words * sizeof(uint32_t)));
        h->overflow_words = overflow_words;
    }
    if (brute) {
        const size_t need = (size_t)(args_in.settings.max_ray_depth + 1) * max_threads * 2u * n_streams;
        if (h->brute_floats4 < need) {
            PT_TRY(hipDeviceSynchronize());
            (void)hipFree(h->brute); h->brute = nullptr;
            PT_TRY(hipMalloc((void**)&h->brute, need * sizeof(float4)));
            h->brute_floats4 = need;
        }
    }
#ifdef CGPT_PHASE_CYCLES
    const bool want_phase_stats = getenv("CGPT_WF_PROFILE") != nullptr;
#else
    const bool want_phase_stats = count && getenv("CGPT_WF_PROFILE") != nullptr;
#endif
    if (want_phase_stats && !h->phase_stats) PT_TRY(hipMalloc((void**)&h->phase_stats, 48 * sizeof(unsigned long long)));
    if (h->phase_stats) PT_TRY(hipMemsetAsync(h->phase_stats, 0, 48 * sizeof(unsigned long long), stream));

    const uint32_t ev_needed = 2u * n_batches;
    if (h->ev_cap < ev_needed) {
        hipEvent_t* grown = static_cast<hipEvent_t*>(realloc(h->ev, (size_t)ev_needed * sizeof(hipEvent_t)));
        if (!grown) { CtxFail(ctx, CGPT_ERR_INVALID, "out of host memory"); return -1; }
        h->ev = grown;
        for (; h->ev_cap < ev_needed; ++h->ev_cap) PT_TRY(hipEventCreate(&h->ev[h->ev_cap]));
    }
    h->ev_used = 0;

    if (h->n_work_counters < n_batches) {
        PT_TRY(hipDeviceSynchronize());
        (void)hipFree(h->work_counters); h->work_counters = nullptr;
        PT_TRY(hipMalloc((void**)&h->work_counters, (size_t)n_batches * kWorkCounters * 8u * sizeof(uint32_t)));
        h->n_work_counters = n_batches;
    }
    PT_TRY(hipEventRecord(CtxStartEvent(ctx), stream));                       // one-time host setup is over: the render's device time starts here
    PT_TRY(hipMemsetAsync(h->work_counters, 0, (size_t)n_batches * kWorkCounters * 8u * sizeof(uint32_t), stream));   // before `begin`: ordered ahead of both streams
    const TraceTune tt = { h->tune.refill_idle, h->tune.inner_repeat, h->tune.leaf_repeat, 1u, h->tune.obj_shift, top_records, 0u, h->tune.lds_tris, h->tune.tail_lanes, 0u };   // shadow rays to the end here: stopping them early (wf_trace does) cost this kernel 2 % in registers

    if (n_streams == 2) {
        PT_TRY(hipEventRecord(h->begin, stream));
        for (int i = 0; i < 2; ++i) PT_TRY(hipStreamWaitEvent(h->streams[i], h->begin, 0));
    }
    int launches = 0;
    uint32_t k = 0;
    for (uint32_t done = 0; done < args_in.n_samples; done += batch, ++k) {
        const uint32_t s = n_streams == 2 ? k & 1u : 0u;
        hipStream_t st = n_streams == 2 ? h->streams[s] : stream;
        const uint32_t bn = std::min(batch, args_in.n_samples - done);
        const uint32_t bfirst = args_in.first_sample + done;
        PtDev pt{};
        pt.st_en = h->st_en[s];
        pt.brute = brute ? h->brute + (size_t)s * (h->brute_floats4 / n_streams) : nullptr;
        pt.stack_overflow = h->overflow + (size_t)s * (h->overflow_words / n_streams);
#ifdef CGPT_PHASE_CYCLES
        pt.phase_stats = h->phase_stats;
#else
        pt.phase_stats = count ? h->phase_stats : nullptr;
#endif
        pt.n_paths = n_pixels * bn;
        pt.g.n_pixels = n_pixels; pt.g.tiles_x = tiles_x; pt.g.div_tiles_x = MakeFastDiv(tiles_x); pt.g.div_n_pixels = MakeFastDiv(n_pixels);
        pt.shade_shift = h->tune.shade_shift;
        pt.g.order = h->tune.path_order; pt.g.n_samples = bn; pt.g.div_samples = MakeFastDiv(bn);
        pt.work = h->work_counters + (size_t)k * kWorkCounters * 8u;
        work_sizes(pt.n_paths, grid.x * (kTraceBlock / 64u), h->tune.fine_rounds, h->tune.chunk, pt.coarse, pt.fine_below);
        // the buffer's previous batch must have been accumulated (same stream: implicit)
        PT_TRY(hipEventRecord(h->ev[h->ev_used++], st));
#define PT_LAUNCH(C, B, O) if (count == C && brute == B && tail == O) hipLaunchKernelGGL((pt_persistent<C, B, O>), grid, trace_block, lds, st, args_in, pt, bfirst, tt);
        PT_EACH_KERNEL(PT_LAUNCH)
#undef PT_LAUNCH
        PT_TRY(hipEventRecord(h->ev[h->ev_used++], st));
        // accumulate in sample order: batch k after batch k-1
        if (n_streams == 2 && k > 0) PT_TRY(hipStreamWaitEvent(st, h->acc_done[(k - 1u) & 1u], 0));
        hipLaunchKernelGGL(pt_accumulate, dim3(std::min((n_pixels + 255u) / 256u, h->n_cus * 8u)), block, 0, st, args_in, (const float4*)pt.st_en, pt.g, bfirst, bn);
        if (n_streams == 2) PT_TRY(hipEventRecord(h->acc_done[s], st));
        PT_TRY(hipGetLastError());
        launches += 2;
    }
    if (n_streams == 2 && k > 0) PT_TRY(hipStreamWaitEvent(stream, h->acc_done[(k - 1u) & 1u], 0));
#ifdef CGPT_PHASE_CYCLES
    if (!count && h->phase_stats) {
        unsigned long long ps[48];
        PT_TRY(hipStreamSynchronize(stream));
        PT_TRY(hipMemcpy(ps, h->phase_stats, sizeof(ps), hipMemcpyDeviceToHost));
        const double tot = (double)ps[16], waves = (double)ps[17];
        static const char* names[6] = { "refill", "inner", "leaf", "object", "shade", "lean" };
        fprintf(stderr, "[pt cycles] %.0f waves, mean life %.0f kcyc, longest %.0f kcyc, mean drain after the last refill %.0f kcyc |", waves, tot / waves / 1e3, ps[19] / 1e3, ps[18] / waves / 1e3);
        double acc = 0;
        for (int i = 0; i < 6; ++i) {
            acc += ps[20 + i];
            fprintf(stderr, " %s %.3f (%llu steps, %.0f cyc/step, %.1f lanes)", names[i], ps[20 + i] / tot, ps[28 + i], ps[28 + i] ? (double)ps[20 + i] / ps[28 + i] : 0.0, ps[28 + i] ? (double)ps[36 + i] / ps[28 + i] : 0.0);
        }
        fprintf(stderr, " other %.3f\n", 1.0 - acc / tot);
    }
#endif
    if (count && h->phase_stats) {                                            // development aid: how full the steps were
        unsigned long long ps[16];
        PT_TRY(hipStreamSynchronize(stream));
        PT_TRY(hipMemcpy(ps, h->phase_stats, sizeof(ps), hipMemcpyDeviceToHost));
        DevCounters c;
        PT_TRY(hipMemcpy(&c, args_in.counters, sizeof(c), hipMemcpyDeviceToHost));
        fprintf(stderr, "[pt profile] rays %llu | inner: %llu wave steps, %.1f lanes/step | leaf: %llu, %.1f | object: %llu, %.1f | shade: %llu, %.1f | votes %llu refills %llu\n",
                c.traced_rays, ps[0], ps[0] ? (double)c.inner_steps / ps[0] : 0.0, ps[1], ps[1] ? (double)ps[8] / ps[1] : 0.0,
                ps[2], ps[2] ? (double)ps[4] / ps[2] : 0.0, ps[3], ps[3] ? (double)ps[5] / ps[3] : 0.0, ps[6], ps[7]);
    }
#undef PT_EACH_KERNEL
#undef PT_TRY
    return launches;
}

}  // namespace cgpt
