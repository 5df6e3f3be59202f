// wavefront_kernels.hip -- the wavefront pipeline (gfx950): per bounce [trace -> shade -> plan -> gather] -> accumulate.
//
// Why: in the megakernel a wave's traversal loop runs until its slowest lane is done (max-vs-mean ray length) and a tile
// runs until its most expensive pixel is done; PMC showed ~10 % active lanes per VALU instruction.  Here
//   * trace<>  is a PERSISTENT kernel: every wave keeps its 64 lanes filled from the round's ray list, each lane runs the
//     reference's ordered stack traversal (ref: Source/BVH.cpp:61-127) on a per-wavefront LDS stack, and a lane that
//     finishes its ray is refilled while its neighbours keep going.  Extend rays and NEE shadow rays are traced by the
//     same kernel (both are closest-hit IntersectScene calls, ref: Source/Main.cpp:299-316,452-453); a shadow ray's
//     epilogue adds its pending contribution to the path's energy.
//   * shade<>  runs shade_bounce() (ref: Main.cpp:404-573) for every extend hit, rewrites the path's slot with its next ray
//     (and its shadow slot with the NEE connection) and appends the surviving path ids / shadow ids to this wave's own
//     output segment with __ballot + mbcnt (active-lane compaction, no atomics).
//   * plan + gather turn the per-wave segments into one dense list per kind (exclusive scan of the segment counts, then a copy).
//   * accumulate adds the finished samples to the float4 accumulator IN SAMPLE ORDER, so the image is bit-identical to
//     the megakernel's and the oracle's (ref: Main.cpp:735-746).
// Work distribution: consumers take 64-item blocks of the dense list strided over the persistent grid (wave w: blocks w,
// w + n_waves, ...).  Every wave therefore gets the same number of rays (+-64) sampled from the whole image, which balances
// cost as well as count with no atomics at all.  Rejected on measurements: (1) compaction through global atomic counters,
// one per wave -- 10.6 M waves/step serialise on one L2 word at ~88 atomics/us and cost more than the tracing; (2) a liveness
// byte per slot scanned by strided waves -- correct but waves own unequal numbers of live rays (59 % wave residency);
// (3) the same with 64 partitioned head counters -- the atomics cost more than the imbalance they removed.
// Slot entry (48 B, three float4 planes, extend slot = path id, shadow slot = cap + path id):
//   A = {o.xyz, t_max}  B = {d.xyz, bits(depth | spec << 8)}  C = extend: hit record {bits obj, tri, bvh_depth, t} written by trace
//   (read by trace only when the same ray is traced again after total internal reflection, SURVEY A-3) | shadow: {pending.xyz, -}.
// Round 0 has no generate kernel and no slot traffic for the rays: trace and shade both recompute the primary ray from the
// path id (ref: Main.cpp:713-716, Camera::GetRay :133-140), trace stores only the 16-byte hit record, shade initialises the path state.
// TracePath (brute force, ref: Main.cpp:581-689) paths -- RENDER_MODE_BRUTE_FORCE, or the left half of the image in the reference's default
// RENDER_MODE_COMPARISON (ref: Main.cpp:215,719-725) -- run through the same rounds: shade<BRUTE> records the level's operation in
// brute[level][path] instead of updating a throughput, and folds the recorded chain over the leaf's radiance, innermost level first
// (float multiplication is not associative), when the path ends.  They have no shadow rays; flag bit 9 of B.w marks them.
// Path state (path id = sample_in_batch * n_pixels + pixel index): {throughput.xyz, bits(rng)} rewritten every bounce while the path
// lives; {energy.xyz, bits(final depth)} touched only when radiance arrives (emissive hit, unoccluded shadow ray).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cctype>
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
void** CtxWavefrontSlot(cgpt_ctx* ctx);
hipEvent_t CtxStartEvent(cgpt_ctx* ctx);
int CtxFail(cgpt_ctx* ctx, int code, const char* fmt, ...);

extern __shared__ uint32_t lds_dyn[];

static constexpr uint32_t kMaxKeys = 128;   // most runs per segment (image bands) the count / prefix tables are sized for

struct WfDev {
    float4* A; float4* B; float4* C;   // 2 * cap slots each: [0, cap) extend, [cap, 2 cap) shadow
    float4* st_tp; float4* st_en;      // cap paths
    float4* brute;                     // TracePath / COMPARISON renders only: [level][path][2] BruteLevel records (shade_device.hpp), max_ray_depth + 1 levels
    uint8_t* hit_flag;                 // cap paths: did the path's extend ray of this round hit anything (retire_misses only; written by trace)
    uint32_t* list_ext; uint32_t* list_sh;     // dense lists of path ids for the next trace / shade (cap entries each)
    uint32_t* seg_ext; uint32_t* seg_sh;       // per-wave output segments of shade (n_segs * seg_cap entries each)
    uint8_t* seg_key_ext; uint8_t* seg_key_sh; // sort keys of the segment entries (n_keys > 1 only)
    uint32_t* seg_count;               // [kind][key][segment]: extend counts, then shadow counts (n_keys = 1: [2 * n_segs])
    uint32_t* seg_prefix;              // exclusive prefix of the above, per kind, key-major: where each (key, segment) run starts in the list
    uint32_t n_keys;                   // runs per segment: 1; 8 with sort (lists binned by the octant of the ray direction); n_bands with bands
    uint32_t n_bands;                  // > 1: the next round's lists are ordered by image band (see wf_shade); 0 / 1: off
    uint32_t band_magic;               // band of path id p = min(n_bands - 1, umulhi(p, band_magic))
    uint32_t* plan;                    // {n_ext, n_sh}
    uint32_t* stack_overflow;          // [level - kLdsStackLevels][thread of the trace grid]: the rarely used deep end of the stack
    unsigned long long* phase_stats;   // COUNT kernels only: {wave steps, lane steps} of the inner / leaf / object step, votes, refills
    uint32_t cap;                      // slots per kind
    uint32_t n_paths;                  // paths of this batch (path ids 0 .. n_paths-1, all valid)
    PathGrid g;                        // path id <-> pixel of the band (trace_steps.hpp)
    uint32_t n_segs, seg_cap;
    uint32_t shade_chunk;              // consecutive 64-path blocks a shade wave takes at a time
    uint32_t trace_chunk;              // same for a trace wave
    uint32_t retire_misses;            // later rounds: trace leaves one byte per extend ray (hit or not) and shade takes only the hits (off for the debug views)
    uint32_t rot_trace[2], rot_shade;  // rotation of the wave order from one row of blocks to the next ([FIRST] for trace): see next_block()
};

// ---- K2/K4 trace: persistent closest-hit traversal with per-lane refill ------------------------------------------------
// `first_round`: the extend list is the identity over all paths and there are no shadow rays yet.
// The traversal states, their voted steps and the LDS layout are in trace_steps.hpp.
#ifndef CGPT_TRACE_WAVES_PER_SIMD
#define CGPT_TRACE_WAVES_PER_SIMD 1
#endif
#ifndef CGPT_SHADE_WAVES_PER_SIMD
#define CGPT_SHADE_WAVES_PER_SIMD 1
#endif

// FIRST (round 0) is a separate instantiation so the later rounds carry neither its code nor its registers.
template <bool COUNT, bool FIRST>
__global__ void __launch_bounds__(kTraceBlock, (!COUNT && !FIRST) ? CGPT_TRACE_WAVES_PER_SIMD : 1) wf_trace(const DevRenderArgs args, const WfDev wf, uint32_t batch_first, const TraceTune tune)
{
    constexpr bool first_round = FIRST;
    const DevScene& sc = args.scene;
    DevCounters* const counters = args.counters;
    const TravCtx ctx = trav_setup(sc, lds_dyn, tune.top_records, wf.stack_overflow, gridDim.x * kTraceBlock, tune.lds_tris);
    lds_u32* const ring = ctx.ring;

    const uint32_t n_ext = first_round ? wf.n_paths : wf.plan[0];
    const uint32_t n_sh = first_round ? 0u : wf.plan[1];
    const uint32_t blocks_ext = (n_ext + 63u) / 64u, n_blocks = blocks_ext + (n_sh + 63u) / 64u;
    const uint32_t n_waves = gridDim.x * (kTraceBlock / 64u);
    // wave-uniform: the wave's next 64-item block.  Runs of trace_chunk consecutive blocks (the same and neighbouring pixels) are dealt
    // out over the waves, so the rays a wave refills its idle lanes with come from where its other lanes' rays came from
    BlockWalk walk = first_block(blockIdx.x * (kTraceBlock / 64u) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));   // wave-uniform: scalar registers
    const uint32_t tchunk = wf.trace_chunk;
    uint32_t block = block_of(walk) * tchunk, chunk_left = tchunk;
    const uint32_t rot = wf.rot_trace[first_round ? 1 : 0];
    uint32_t ring_count = 0;

    Trav r;
    r.d = mk(0.0f); r.rs = make_ray_slab(r.d, r.d); r.t = 0.0f;
    r.obj = kNoHit; r.tri = 0; r.depth = 0; r.cur_obj = 0; r.code = kIdle; r.sp = 0; r.fast_levels = kLdsStackLevels;
    uint32_t slot_of_lane = 0;
    uint32_t wave_rays = 0;                                                   // wave-uniform: rays this wave started (later rounds)
    Counters cnt = { 0, 0, 0, 0, 0 };
    uint32_t ph_inner = 0, ph_leaf = 0, ph_leaf_lanes = 0, ph_obj = 0, ph_obj_lanes = 0, ph_votes = 0, ph_refills = 0;   // wave-uniform, COUNT only
#ifdef CGPT_PHASE_CYCLES
    // diagnostic build (scripts/build_variant.sh cyc -DCGPT_PHASE_CYCLES): where a later-round wave's cycles go, by phase
    constexpr bool kCyc = !COUNT && !FIRST;
    unsigned long long cy_refill = 0, cy_inner = 0, cy_leaf = 0, cy_obj = 0, cy_start = 0, cy_mark = 0;
    uint32_t cn_inner = 0, cn_leaf = 0, cn_obj = 0, cl_inner = 0, cl_leaf = 0, cl_obj = 0, cn_inner_lds = 0;
    if (kCyc) cy_start = __builtin_readcyclecounter();
#define CYC_BEGIN() do { if (kCyc) cy_mark = __builtin_readcyclecounter(); } while (0)
#define CYC_END(acc) do { if (kCyc) acc += __builtin_readcyclecounter() - cy_mark; } while (0)
#else
#define CYC_BEGIN() do { } while (0)
#define CYC_END(acc) do { } while (0)
#endif

    auto finish_ray = [&]() {                                                 // the ray of this lane has seen every object of the scene
        // The slot's addresses are formed here, when the ray ends: left to the optimiser they are hoisted to where the slot is assigned
        // (loop-invariant for the whole traversal) and six 64-bit addresses ride along through every step -- ten registers of the budget.
        uint32_t slot = slot_of_lane;
        asm volatile("" : "+v"(slot));
        if (slot >= wf.cap) {                                                 // connect epilogue, ref: Main.cpp:454-463
            if (r.obj == kNoHit) {
                const float4 pe = ld_stream(&wf.C[slot]);
                const uint32_t pid = slot - wf.cap;
                float4 en = ld_stream(&wf.st_en[pid]);
                en.x += pe.x; en.y += pe.y; en.z += pe.z;
                st_stream(&wf.st_en[pid], en);
            }
        } else {
            // A later-round extend ray that left the scene ends its path with nothing to add (ref: Main.cpp:415-416; the debug views
            // read the last depth, so they take the full path): ~40 % of the later rounds' rays.  Shade never sees them: it reads
            // this byte per ray and queues only the hits.
            const bool hit = r.obj != kNoHit;
            if (!first_round && wf.retire_misses) wf.hit_flag[slot] = hit ? (uint8_t)1 : (uint8_t)0;
            if (first_round || !wf.retire_misses || hit) {
                float4 c; c.x = __uint_as_float(r.obj); c.y = __uint_as_float(r.tri); c.z = __uint_as_float(trav_depth(r)); c.w = r.t;
                st_stream(&wf.C[slot], c);                                    // hit record
            }
        }
        r.code = kIdle;
    };

    for (;;) {
        if (COUNT) ph_refills++;
        CYC_BEGIN();
        // ---- refill idle lanes from the ring; top the ring up with this wave's next blocks of the dense list ----
        const unsigned long long need = __builtin_amdgcn_ballot_w64(r.code == kIdle);
        const uint32_t n_need = (uint32_t)__popcll(need);
        while (ring_count < n_need && block < n_blocks) {
            uint32_t s = 0; bool valid;
            if (block < blocks_ext) {
                const uint32_t i = block * 64u + lane_id();
                valid = i < n_ext;
                if (valid) s = first_round ? i : ld_stream(&wf.list_ext[i]);
            } else {
                const uint32_t i = (block - blocks_ext) * 64u + lane_id();
                valid = i < n_sh;
                if (valid) s = wf.cap + ld_stream(&wf.list_sh[i]);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(valid);
            if (valid) ring[ring_count + rank_in_mask(m)] = s;
            ring_count += (uint32_t)__popcll(m);
            if (--chunk_left) ++block;
            else { next_block(walk, n_waves, rot); block = block_of(walk) * tchunk; chunk_left = tchunk; }
            __builtin_amdgcn_wave_barrier();
        }
        if (n_need && ring_count) {
            const uint32_t take = min(n_need, ring_count);
            const uint32_t rank = rank_in_mask(need);
            if (r.code == kIdle && rank < take) {
                const uint32_t slot = ring[ring_count - 1u - rank];
                slot_of_lane = slot;
                bool ok = true;
                V3 o, d; float t; uint32_t obj = kNoHit, tri = 0, depth = 0;      // fresh ray (extend or shadow, ref: Primitives.h:79-81)
                if (first_round) {                                            // primary ray from the path id, nothing to load
                    uint32_t rng_unused, px_unused;
                    Ray pr;
                    ok = primary_ray(args, wf.g, slot, batch_first, pr, rng_unused, px_unused);   // false: padding of an edge tile
                    o = pr.o; d = pr.d; t = pr.t;
                } else {
                    const float4 a = ld_stream(&wf.A[slot]), b = ld_stream(&wf.B[slot]);
                    o = mk(a.x, a.y, a.z); t = a.w; d = mk(b.x, b.y, b.z);
                    if (slot < wf.cap && t != 1e34f) {                        // the same ray again after total internal reflection:
                        const float4 c = ld_stream(&wf.C[slot]);              // it keeps its previous hit as payload (SURVEY A-3)
                        obj = __float_as_uint(c.x); tri = __float_as_uint(c.y); depth = __float_as_uint(c.z);
                    }
                }
                if (ok) {
                    trav_start(ctx, r, o, d, t, obj, tri, depth);
                    if (!first_round && tune.shadow_any_hit != 0u && slot >= wf.cap) r.depth |= kAnyHitBit;
                    if (first_round) cnt.rays++;                              // later rounds: every id of the lists is a ray, counted per wave below
                }
            }
            __builtin_amdgcn_wave_barrier();
            ring_count -= take;
            if (!first_round) wave_rays += take;
        }
        // Done when nothing is in flight and nothing is left to fetch.  Nothing in flight alone is not enough: every id just handed
        // out may have been padding of an edge tile (pixel-major ids put a padded pixel's samples side by side); the step loop below
        // then falls straight through and the wave fetches on.
        CYC_END(cy_refill);
        if (__builtin_amdgcn_ballot_w64(r.code != kIdle) == 0ull && ring_count == 0u && block >= n_blocks) break;
        const bool can_refill = ring_count != 0u || block < n_blocks;

        // ---- run the most popular state's step until enough lanes are idle ----
        for (;;) {
            const uint32_t n_inner = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code < kStartObject));
            const uint32_t n_leaf = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((int32_t)r.code < 0));
            const uint32_t n_obj = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code == kStartObject));
            const uint32_t n_busy = n_inner + n_leaf + n_obj;
            if (n_busy == 0u) break;
            if (can_refill && 64u - n_busy >= tune.refill_idle) break;        // enough idle lanes: go refill them
            const uint32_t w_obj = n_obj << tune.obj_shift;
            if (COUNT) ph_votes++;
            // Round 0: the 64 lanes of a wave carry the SAME primary ray (samples of one pixel, no jitter: SURVEY A-14), so there is no
            // divergence for the voted branch-free steps to buy off; every lane walks its meshes in the lean loop (the reference's own
            // control flow, trace_steps.hpp: lean_traverse -- ~50 instructions per node instead of ~75 and no votes), then takes the
            // object step.  Same results, same counters.
            if (FIRST && tune.first_lean) {
                if (r.code < kStartObject || (int32_t)r.code < 0) lean_traverse<COUNT, !FIRST>(ctx, r, cnt);
                if (r.code == kStartObject && object_step<COUNT, !FIRST>(ctx, r, cnt)) finish_ray();
                continue;
            }

            if (n_inner >= n_leaf && n_inner >= w_obj) {
                CYC_BEGIN();
                do {
                    if (COUNT) ph_inner++;
#ifdef CGPT_PHASE_CYCLES
                    if (kCyc) { cn_inner++; cl_inner += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code < kStartObject)); cn_inner_lds += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code < ctx.n_top)); }
#endif
                    if (r.code < kStartObject) inner_step<COUNT>(ctx, r, cnt);
                } while ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code < kStartObject)) >= tune.inner_repeat);
                CYC_END(cy_inner);
            } else if (n_leaf >= w_obj) {
                CYC_BEGIN();
                do {
                    if (COUNT) { ph_leaf++; ph_leaf_lanes += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((int32_t)r.code < 0)); }
#ifdef CGPT_PHASE_CYCLES
                    if (kCyc) { cn_leaf++; cl_leaf += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((int32_t)r.code < 0)); }
#endif
                    if ((int32_t)r.code < 0) leaf_step<COUNT, !FIRST>(ctx, r, cnt);
                } while ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64((int32_t)r.code < 0)) >= tune.leaf_repeat);
                CYC_END(cy_leaf);
            } else {
                CYC_BEGIN();
                do {
                    if (COUNT) { ph_obj++; ph_obj_lanes += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code == kStartObject)); }
#ifdef CGPT_PHASE_CYCLES
                    if (kCyc) { cn_obj++; cl_obj += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code == kStartObject)); }
#endif
                    if (r.code == kStartObject && object_step<COUNT, !FIRST>(ctx, r, cnt)) finish_ray();
                } while ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r.code == kStartObject)) >= tune.obj_repeat);
                CYC_END(cy_obj);
            }
        }
    }

#ifdef CGPT_PHASE_CYCLES
    if (kCyc && wf.phase_stats && lane_id() == 0u) {
        const unsigned long long total = __builtin_readcyclecounter() - cy_start;
        const unsigned long long v[12] = { total, cy_refill, cy_inner, cy_leaf, cy_obj, cn_inner, cn_leaf, cn_obj, cl_inner, cl_leaf, cl_obj, cn_inner_lds };
        for (int i = 0; i < 12; ++i) atomicAdd(&wf.phase_stats[8 + i], v[i]);
        atomicAdd(&wf.phase_stats[20], 1ull);
    }
    if (kCyc && wf.phase_stats) {
        wave_add_u64(&wf.phase_stats[21], cnt.global_inner); wave_add_u64(&wf.phase_stats[22], cnt.both_miss);
        wave_add_u64(&wf.phase_stats[23], cnt.xy_both_miss); wave_add_u64(&wf.phase_stats[24], cnt.x_both_miss);
    }
#endif
    if (first_round) wave_add_u64(&counters->traced_rays, cnt.rays);
    else if (lane_id() == 0u && wave_rays) atomicAdd(&counters->traced_rays, (unsigned long long)wave_rays);
    if (COUNT) {
        wave_add_u64(&counters->inner_steps, cnt.inner);
        wave_add_u64(&counters->tri_tests, cnt.tris);
        wave_add_u64(&counters->bvh_depth_sum, cnt.depth);
        if (wf.phase_stats && lane_id() == 0u) {
            atomicAdd(&wf.phase_stats[0], (unsigned long long)ph_inner); atomicAdd(&wf.phase_stats[1], (unsigned long long)ph_leaf);
            atomicAdd(&wf.phase_stats[2], (unsigned long long)ph_obj); atomicAdd(&wf.phase_stats[3], (unsigned long long)ph_obj_lanes);
            atomicAdd(&wf.phase_stats[4], (unsigned long long)ph_votes); atomicAdd(&wf.phase_stats[5], (unsigned long long)ph_refills);
            atomicAdd(&wf.phase_stats[6], (unsigned long long)ph_leaf_lanes);
        }
    }
}

// ---- K3 shade: one bounce per extend hit; survivors compacted into this wave's output segment ---------------------------
// BRUTE: the render has TracePath paths (RENDER_MODE_BRUTE_FORCE / COMPARISON); a separate instantiation, so the TracePathAdvanced
// renders carry neither its code nor its registers.
template <bool COUNT, bool FIRST, bool BRUTE = false>
__global__ void __launch_bounds__(256, CGPT_SHADE_WAVES_PER_SIMD) wf_shade(const DevRenderArgs args, const WfDev wf, uint32_t batch_first)
{
    constexpr bool first_round = FIRST;
    const DevScene& sc = args.scene;
    const uint32_t n_ext = first_round ? wf.n_paths : wf.plan[0];
    const uint32_t n_blocks = (n_ext + 63u) / 64u;
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);              // = this wave's segment
    uint32_t* const out_ext = wf.seg_ext + (size_t)wave * wf.seg_cap;
    uint32_t* const out_sh = wf.seg_sh + (size_t)wave * wf.seg_cap;
    uint32_t count_ext = 0, count_sh = 0;                                     // wave-uniform
    uint32_t kc_ext[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, kc_sh[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // per-key counts (wave-uniform; n_keys == 8 only)
    const bool sorted = wf.n_keys > 1u && wf.n_bands <= 1u;
    // Image bands: a wave walks its chunks in ascending list order and the lists are band-major (round 0: path ids in image order), so
    // the entries it appends are already grouped by the band of their pixel -- contiguous runs inside its segment, with no per-entry key.
    // The wave only notes where each band's run ends; plan + gather then put all segments' runs of band 0 first, then band 1, ...
    // The rays trace works on at any one time (a contiguous piece of the list) then come from ONE strip of the image instead of from all
    // over it, and the part of the tree they walk stays in the 4 MB L2 of every XCD.
    const bool banded = wf.n_bands > 1u;
    uint32_t cur_band = 0, band_start_ext = 0, band_start_sh = 0;             // wave-uniform
    Counters cnt = { 0, 0, 0, 0, 0 };

    // The order in which a wave appends its survivors is the order of the next round's ray list.  Taking runs of `chunk`
    // consecutive blocks (the samples of one pixel and of its neighbours in round 0, and their descendants later) keeps the rays of
    // a 64-ray block of every later round from one neighbourhood of the image rather than from unrelated ones (measured +1.3 % with
    // sample-major ids; flat between 1 and 16 blocks with the pixel-major ones, profiles/r02/experiments.md).
    const uint32_t chunk = wf.shade_chunk;
    const uint32_t n_chunks = (n_blocks + chunk - 1u) / chunk;
    BlockWalk walk = first_block(wave);
    uint32_t ci = wave, block = ci * chunk, block_end = min((ci + 1u) * chunk, n_blocks);
    bool more = ci < n_chunks;                                                // wave-uniform: the list has a block left for this wave
    auto advance = [&]() {
        if (++block >= block_end) {
            next_block(walk, n_waves, wf.rot_shade);
            ci = block_of(walk);
            more = ci < n_chunks;
            block = ci * chunk; block_end = min((ci + 1u) * chunk, n_blocks);
        }
    };
    // Later rounds with retire_misses: the wave reads one hit byte per ray of its blocks and queues the path ids of the hits (in list
    // order: the queue is first in, first out, so what the wave appends to its segments is in the order it would have been anyway);
    // every pass below then has 64 hits to shade instead of the ~38 a block of the list holds.
    __shared__ uint32_t s_queue[4][128];
    uint32_t* const queue = s_queue[threadIdx.x >> 6];
    uint32_t queued = 0;                                                      // wave-uniform
    const bool hits_only = !first_round && wf.retire_misses != 0u;

    for (;;) {
        bool active = false;
        bool emit_ext = false, emit_sh = false;
        uint32_t pid = 0, key_ext = 0, key_sh = 0;
        if (!hits_only) {
            if (!more) break;
            const uint32_t i = block * 64u + lane_id();
            active = i < n_ext;
            if (active) pid = first_round ? i : ld_stream(&wf.list_ext[i]);
            advance();
        } else {
            while (queued < 64u && more) {
                const uint32_t i = block * 64u + lane_id();
                uint32_t p = 0; bool hit = false;
                if (i < n_ext) { p = ld_stream(&wf.list_ext[i]); hit = wf.hit_flag[p] != 0; }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (hit) queue[queued + rank_in_mask(m)] = p;
                queued += (uint32_t)__popcll(m);
                advance();
            }
            if (queued == 0u) break;
            __builtin_amdgcn_wave_barrier();
            const uint32_t take = min(queued, 64u), rest = queued - take;
            active = lane_id() < take;
            if (active) pid = queue[lane_id()];
            const uint32_t moved = queue[64u + lane_id()];
            __builtin_amdgcn_wave_barrier();
            if (lane_id() < rest) queue[lane_id()] = moved;
            __builtin_amdgcn_wave_barrier();
            queued = rest;
        }
        if (active) {
            const float4 c = ld_stream(&wf.C[pid]);                           // hit record written by trace
            Ray ray, shadow;
            PathState ps;
            bool is_pixel = true;
            bool brute_path = false;                                          // this path runs TracePath
            if (first_round) {                                                // primary ray and fresh path state from the path id
                uint32_t px = 0;
                is_pixel = primary_ray(args, wf.g, pid, batch_first, ray, ps.rng, px);
                ps.throughput = mk(1.0f); ps.energy = mk(0.0f); ps.depth = 0; ps.is_specular = false;
                if (BRUTE) brute_path = args.settings.render_mode == 1u || (args.settings.render_mode == 0u && px < args.width / 2u);   // ref: Main.cpp:719-729
            } else {
                const float4 a = ld_stream(&wf.A[pid]), b = ld_stream(&wf.B[pid]);
                ray.o = mk(a.x, a.y, a.z); ray.d = mk(b.x, b.y, b.z);
                const float4 tp = ld_stream(&wf.st_tp[pid]);
                ps.throughput = mk(tp.x, tp.y, tp.z); ps.rng = __float_as_uint(tp.w);
                ps.energy = mk(0.0f);                                         // the bounce's own addition; folded into st_en below
                const uint32_t fl = __float_as_uint(b.w);
                ps.depth = fl & 0xFFu; ps.is_specular = (fl & 0x100u) != 0u;
                if (BRUTE) brute_path = (fl & 0x200u) != 0u;
            }
            ray.t = c.w; ray.obj = __float_as_uint(c.x); ray.tri = __float_as_uint(c.y); ray.bvh_depth = __float_as_uint(c.z);
            shadow = ray;
            V3 pending = mk(0.0f);

            uint32_t flags = kBounceTerminate;
            if (BRUTE && brute_path) {
                // One TracePath level (ref: Main.cpp:581-689): record this level's operation and go on with the child ray, or fold the
                // recorded chain over the leaf's radiance, innermost level first.  The level just made is applied from registers.
                if (is_pixel) {
                    BruteLevel lv; V3 leaf = mk(0.0f);
                    bool fold = brute_bounce<COUNT>(sc, args.settings, ray, ps.rng, ps.depth, lv, leaf, cnt) == kBruteLeaf;
                    uint32_t stored = ps.depth;                               // levels 0 .. stored-1 are in memory
                    if (!fold) {
                        ps.depth++;
                        if ((int32_t)ps.depth > args.settings.max_ray_depth) {   // the child returns black before tracing (ref: Main.cpp:589-590)
                            fold = true;
                            leaf = brute_apply(lv, mk(0.0f));
                        } else {
                            float4* rec = wf.brute + ((size_t)stored * wf.cap + pid) * 2u;
                            float4 r0, r1;
                            r0.x = __uint_as_float(lv.kind); r0.y = lv.a.x; r0.z = lv.a.y; r0.w = lv.a.z;
                            r1.x = lv.cosi; r1.y = lv.absorb.x; r1.z = lv.absorb.y; r1.w = lv.absorb.z;
                            st_stream(&rec[0], r0); st_stream(&rec[1], r1);
                        }
                    }
                    if (fold) {
                        V3 L = leaf;
                        for (uint32_t k = stored; k-- > 0u;) {
                            const float4* rec = wf.brute + ((size_t)k * wf.cap + pid) * 2u;
                            const float4 r0 = ld_stream(&rec[0]), r1 = ld_stream(&rec[1]);
                            BruteLevel b;
                            b.kind = __float_as_uint(r0.x); b.a = mk(r0.y, r0.z, r0.w); b.cosi = r1.x; b.absorb = mk(r1.y, r1.z, r1.w);
                            L = brute_apply(b, L);
                        }
                        ps.energy = L;
                        flags = kBounceTerminate | kBounceBruteDone;
                    } else {
                        flags = 0u;
                    }
                }
            } else if (is_pixel) flags = shade_bounce<COUNT>(sc, args.settings, ray, ps, shadow, pending, cnt);
            emit_ext = (flags & kBounceTerminate) == 0u;
            emit_sh = (flags & kBounceShadow) != 0u;

            // radiance added by this bounce (emissive hit / BVH-depth view): energy_old + x, the reference's single addition
            const bool final_depth_needed = args.settings.debug_mode == 1u && !emit_ext;      // ray-depth view reads the last depth
            if (is_pixel && (first_round || (flags & (kBounceEnergy | kBounceBruteDone)) || final_depth_needed)) {
                float4 en;
                if (first_round || (BRUTE && (flags & kBounceBruteDone))) { en.x = 0.0f; en.y = 0.0f; en.z = 0.0f; en.w = 0.0f; }
                else en = ld_stream(&wf.st_en[pid]);
                if (BRUTE && (flags & kBounceBruteDone)) { en.x = ps.energy.x; en.y = ps.energy.y; en.z = ps.energy.z; }   // TracePath's return value, as it is
                if (flags & kBounceEnergy) { en.x += ps.energy.x; en.y += ps.energy.y; en.z += ps.energy.z; }
                en.w = __uint_as_float(ps.depth & 0xFFu);
                st_stream(&wf.st_en[pid], en);
            }
            if (emit_ext) {                                                   // the path lives on: state + next extend ray, same slot
                float4 tpo;
                tpo.x = ps.throughput.x; tpo.y = ps.throughput.y; tpo.z = ps.throughput.z; tpo.w = __uint_as_float(ps.rng);
                st_stream(&wf.st_tp[pid], tpo);
                float4 na, nb;
                na.x = ray.o.x; na.y = ray.o.y; na.z = ray.o.z; na.w = ray.t;    // 1e34 for a fresh ray, the hit t for a re-traced one
                nb.x = ray.d.x; nb.y = ray.d.y; nb.z = ray.d.z;
                nb.w = __uint_as_float((ps.depth & 0xFFu) | (ps.is_specular ? 0x100u : 0u) | ((BRUTE && brute_path) ? 0x200u : 0u));
                st_stream(&wf.A[pid], na); st_stream(&wf.B[pid], nb);         // C keeps the hit record (payload of a re-traced ray)
                key_ext = (ray.d.x < 0.0f ? 1u : 0u) | (ray.d.y < 0.0f ? 2u : 0u) | (ray.d.z < 0.0f ? 4u : 0u);
            }
            if (emit_sh) {                                                    // NEE connection, slot cap + pid
                const uint32_t ss = wf.cap + pid;
                float4 sa, sb, scc;
                sa.x = shadow.o.x; sa.y = shadow.o.y; sa.z = shadow.o.z; sa.w = shadow.t;
                sb.x = shadow.d.x; sb.y = shadow.d.y; sb.z = shadow.d.z; sb.w = 0.0f;
                scc.x = pending.x; scc.y = pending.y; scc.z = pending.z; scc.w = 0.0f;
                st_stream(&wf.A[ss], sa); st_stream(&wf.B[ss], sb); st_stream(&wf.C[ss], scc);
                key_sh = (shadow.d.x < 0.0f ? 1u : 0u) | (shadow.d.y < 0.0f ? 2u : 0u) | (shadow.d.z < 0.0f ? 4u : 0u);
            }
        }
        // active-lane compaction into the wave's own segments: __ballot + mbcnt, no atomics
        const unsigned long long m_ext = __builtin_amdgcn_ballot_w64(emit_ext), m_sh = __builtin_amdgcn_ballot_w64(emit_sh);
        if (banded) {                                                         // close the runs of the bands this pass has left behind
            const uint32_t band = min(wf.n_bands - 1u, __umulhi(pid, wf.band_magic));
            const bool emits = emit_ext | emit_sh;
            while (__builtin_amdgcn_ballot_w64(emits && band > cur_band) != 0ull) {
                const unsigned long long behind = __builtin_amdgcn_ballot_w64(band <= cur_band);
                const uint32_t end_ext = count_ext + (uint32_t)__popcll(m_ext & behind), end_sh = count_sh + (uint32_t)__popcll(m_sh & behind);
                if (lane_id() == 0u) {
                    wf.seg_count[cur_band * wf.n_segs + wave] = end_ext - band_start_ext;
                    wf.seg_count[(wf.n_bands + cur_band) * wf.n_segs + wave] = end_sh - band_start_sh;
                }
                band_start_ext = end_ext; band_start_sh = end_sh;
                ++cur_band;
            }
        }
        if (emit_ext) st_stream(&out_ext[count_ext + rank_in_mask(m_ext)], pid);
        if (emit_sh) st_stream(&out_sh[count_sh + rank_in_mask(m_sh)], pid);
        if (sorted) {                                                         // the direction octants, for the binning in wf_gather
            if (emit_ext) wf.seg_key_ext[(size_t)wave * wf.seg_cap + count_ext + rank_in_mask(m_ext)] = (uint8_t)key_ext;
            if (emit_sh) wf.seg_key_sh[(size_t)wave * wf.seg_cap + count_sh + rank_in_mask(m_sh)] = (uint8_t)key_sh;
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) {
                kc_ext[k] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(emit_ext && key_ext == k));
                kc_sh[k] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(emit_sh && key_sh == k));
            }
        }
        count_ext += (uint32_t)__popcll(m_ext);
        count_sh += (uint32_t)__popcll(m_sh);
    }
    if (banded) {                                                             // the last band this wave reached takes the rest; later bands are empty
        if (lane_id() == 0u)
            for (uint32_t b = cur_band; b < wf.n_bands; ++b) {
                wf.seg_count[b * wf.n_segs + wave] = b == cur_band ? count_ext - band_start_ext : 0u;
                wf.seg_count[(wf.n_bands + b) * wf.n_segs + wave] = b == cur_band ? count_sh - band_start_sh : 0u;
            }
    } else if (lane_id() == 0) {
        if (!sorted) { wf.seg_count[wave] = count_ext; wf.seg_count[wf.n_segs + wave] = count_sh; }
        else {
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) { wf.seg_count[k * wf.n_segs + wave] = kc_ext[k]; wf.seg_count[(8u + k) * wf.n_segs + wave] = kc_sh[k]; }
        }
    }
    if (COUNT) wave_add_u64(&args.counters->closest_hits, cnt.hits);
}

// ---- plan: exclusive scan of the segment counts -------------------------------------------------------------------------------
// One 256-thread block per (kind, key): the exclusive prefix of that key's n_segs counts (a few thousand) goes to seg_prefix, the key's
// total to plan[2 + kind * n_keys + key]; wf_gather adds the totals of the keys before (and sums them into plan[kind] for the next
// round's kernels).  One key: plan[kind] directly.  Small blocks with 1 KB of LDS start in the wave slots a resident persistent kernel
// of another batch leaves free (a 1024-thread block had to wait for a whole CU to drain: 0.4 ms average in the 8-pool profile); one
// block scanning all keys in turn took 100 us with 8 keys, on the critical path of its batch.
__global__ void __launch_bounds__(256) wf_plan(const WfDev wf)
{
    __shared__ uint32_t partial[256];
    const uint32_t n = wf.n_segs;
    const uint32_t kk = blockIdx.x;                                           // kind * n_keys + key
    const uint32_t* cnt = wf.seg_count + (size_t)kk * n;
    uint32_t* pre = wf.seg_prefix + (size_t)kk * n;
    const uint32_t per = (n + 255u) / 256u;
    const uint32_t begin = min(threadIdx.x * per, n), end = min(begin + per, n);
    uint32_t sum = 0;
    for (uint32_t i = begin; i < end; ++i) sum += cnt[i];
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 256u; off <<= 1) {                           // Hillis-Steele inclusive scan
        const uint32_t v = threadIdx.x >= off ? partial[threadIdx.x - off] : 0u;
        __syncthreads();
        partial[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = partial[threadIdx.x] - sum;                                // exclusive prefix of this thread's range
    for (uint32_t i = begin; i < end; ++i) { pre[i] = run; run += cnt[i]; }
    if (threadIdx.x == 255u) {
        wf.plan[2u + kk] = partial[255];
        if (wf.n_keys == 1u) wf.plan[kk] = partial[255];
    }
}

// ---- gather: segments -> dense lists -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wf_gather(const WfDev wf)
{
    if (wf.n_keys == 1u) {
        for (uint32_t s = blockIdx.x; s < 2u * wf.n_segs; s += gridDim.x) {
            const bool sh = s >= wf.n_segs;
            const uint32_t seg = sh ? s - wf.n_segs : s;
            const uint32_t n = wf.seg_count[s], base = wf.seg_prefix[s];
            const uint32_t* src = (sh ? wf.seg_sh : wf.seg_ext) + (size_t)seg * wf.seg_cap;
            uint32_t* dst = (sh ? wf.list_sh : wf.list_ext) + base;
            // 16 bytes per lane: the segment starts on a 16-byte boundary, its place in the list on a 4-byte one (global dwordx4
            // accesses only need dword alignment).  Round 0 moves ~1 GB per batch here while the other batch's shade streams its state.
            typedef uint32_t u4a4 __attribute__((ext_vector_type(4), aligned(4)));
            const uint32_t n4 = n & ~3u;
            for (uint32_t i = threadIdx.x * 4u; i < n4; i += blockDim.x * 4u) *reinterpret_cast<u4a4*>(dst + i) = *reinterpret_cast<const u4a4*>(src + i);
            if (threadIdx.x < n - n4) dst[n4 + threadIdx.x] = src[n4 + threadIdx.x];
        }
        return;
    }
    if (wf.n_bands > 1u) {                                                    // band runs: contiguous in the segment, each copied to its place in the list
        // One block per segment.  The run table of the segment (where each band's run starts in the segment and in the list) is built
        // once in LDS -- all counts, prefixes and band totals fetched in parallel, then one short serial pass -- and the four waves copy
        // runs side by side; read one after the other from HBM, 32 bands were 32 dependent round trips per segment.
        __shared__ uint32_t s_n[kMaxKeys], s_dst[kMaxKeys], s_src[kMaxKeys];
        typedef uint32_t u4a4 __attribute__((ext_vector_type(4), aligned(4)));
        const uint32_t lane = threadIdx.x & 63u, wave_in_block = threadIdx.x >> 6;
        for (uint32_t s = blockIdx.x; s < 2u * wf.n_segs; s += gridDim.x) {
            const bool sh = s >= wf.n_segs;
            const uint32_t seg = sh ? s - wf.n_segs : s;
            const uint32_t* cnt = wf.seg_count + (sh ? wf.n_bands * wf.n_segs : 0u);
            const uint32_t* pre = wf.seg_prefix + (sh ? wf.n_bands * wf.n_segs : 0u);
            const uint32_t* key_total = wf.plan + 2u + (sh ? wf.n_bands : 0u);
            __syncthreads();                                                  // the previous segment's table is no longer read
            if (threadIdx.x < wf.n_bands) {
                s_n[threadIdx.x] = cnt[threadIdx.x * wf.n_segs + seg];
                s_dst[threadIdx.x] = pre[threadIdx.x * wf.n_segs + seg];     // + the totals of the bands before, below
                s_src[threadIdx.x] = key_total[threadIdx.x];
            }
            __syncthreads();
            if (threadIdx.x == 0u) {
                uint32_t off = 0, base = 0;                                   // base: entries of the bands before this one, all segments
                for (uint32_t b = 0; b < wf.n_bands; ++b) {
                    const uint32_t n = s_n[b], total = s_src[b];
                    s_dst[b] += base; s_src[b] = off;
                    off += n; base += total;
                }
                if (seg == 0u) wf.plan[sh ? 1 : 0] = base;                    // the list's length, for the next round's kernels
            }
            __syncthreads();
            const uint32_t* src = (sh ? wf.seg_sh : wf.seg_ext) + (size_t)seg * wf.seg_cap;
            uint32_t* const list = sh ? wf.list_sh : wf.list_ext;
            for (uint32_t b = wave_in_block; b < wf.n_bands; b += 4u) {
                const uint32_t n = s_n[b], n4 = n & ~3u;
                uint32_t* dst = list + s_dst[b];
                const uint32_t* run = src + s_src[b];                         // 16 bytes per lane, dword-aligned at both ends
                for (uint32_t i = lane * 4u; i < n4; i += 256u) *reinterpret_cast<u4a4*>(dst + i) = *reinterpret_cast<const u4a4*>(run + i);
                if (lane < n - n4) dst[n4 + lane] = run[n4 + lane];
            }
        }
        return;
    }
    // binned by key (a counting sort whose counts shade already took): one wave per segment, entries in order, each to the next
    // free place of its (key, segment) run -- stable, so rays of one key keep the image-neighbourhood order of their segment
    const uint32_t n_waves = gridDim.x * 4u;
    for (uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6); s < 2u * wf.n_segs; s += n_waves) {
        const bool sh = s >= wf.n_segs;
        const uint32_t seg = sh ? s - wf.n_segs : s;
        const uint32_t* cnt = wf.seg_count + (sh ? 8u * wf.n_segs : 0u);
        const uint32_t* pre = wf.seg_prefix + (sh ? 8u * wf.n_segs : 0u);
        uint32_t n = 0, next[8], base = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) { n += cnt[k * wf.n_segs + seg]; next[k] = base + pre[k * wf.n_segs + seg]; base += wf.plan[2u + (sh ? 8u : 0u) + k]; }
        if (seg == 0u && lane_id() == 0u) wf.plan[sh ? 1 : 0] = base;
        const uint32_t* src = (sh ? wf.seg_sh : wf.seg_ext) + (size_t)seg * wf.seg_cap;
        const uint8_t* keys = (sh ? wf.seg_key_sh : wf.seg_key_ext) + (size_t)seg * wf.seg_cap;
        uint32_t* dst = sh ? wf.list_sh : wf.list_ext;
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t i = base + lane_id();
            const bool valid = i < n;
            const uint32_t pid = valid ? src[i] : 0u, key = valid ? keys[i] : 8u;
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(key == k);
                if (key == k) dst[next[k] + rank_in_mask(m)] = pid;
                next[k] += (uint32_t)__popcll(m);
            }
        }
    }
}

// ---- K5 accumulate + pack: samples of the batch in order (ref: Main.cpp:735-746, MathLib.h:144-152) ------------------------
__global__ void __launch_bounds__(256) wf_accumulate(const DevRenderArgs args, const WfDev wf, uint32_t batch_first, uint32_t batch_n)
{
    accumulate_batch(args, wf.st_en, wf.g, batch_first, batch_n);
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
// Batches of samples are independent until the final accumulate, and every bounce round ends in a tail where a few
// long rays keep a handful of waves busy.  Several batches are therefore in flight at once, each with its own pool on
// its own HIP stream, so one batch's tail overlaps another batch's bulk; the accumulate kernels are chained with events so
// samples are still added in order.
static constexpr uint32_t kMaxPools = 8;

struct WfTuning {               // defaults measured on MI355X (profiles/r01); overridable for sweeps via CGPT_WF_* env vars
    uint32_t pools = 8;         // most sample batches in flight (the memory budget usually allows fewer)
    uint32_t batch = 0;         // samples per batch; 0 = auto (LaunchWavefront: "samples per batch")
    uint32_t max_batch = 512;   // auto: largest batch (a 1080p / 8 band at 1024 spp: 128 -> 54.3 ms, 256 -> 49.4, 512 -> 43.7: fewer, longer rounds)
    uint32_t pool_paths_mi = 512;   // auto: most paths per pool, in Mi (path ids and slot indices are 32-bit: 2 * paths < 2^32)
    uint32_t budget_gib = 96;   // HBM the pools may take (also at most half of what is free)
    uint32_t refill_idle = 16;  // trace leaves its traversal loop to refill once this many lanes are idle
    uint32_t leaf_repeat = 4;         // same for leaf triangles (measured plateau: inner 16-20, leaf 4-8)
    uint32_t inner_repeat = 20;       // trace keeps taking inner steps without re-voting while this many lanes are at inner nodes
    uint32_t obj_repeat = 1;          // same for the object step (1: until no lane is at an object boundary)
    uint32_t obj_shift = 0;           // lanes at an object boundary count 2^shift times in the vote
    uint32_t top_records = kLdsTopMax;   // records of the top of the tree mirrored in LDS
    uint32_t max_trace_blocks = 64;   // cap on trace blocks per CU (occupancy experiments)
    uint32_t shade_chunk = 4;         // consecutive blocks per shade work item
    uint32_t shade_chunk_banded = 32; // the same when the lists are ordered by image band: a longer piece of one band per work item (C3 86.8-87.3 -> 85.8-86.0 ms,
                                      // C4 rank share 33.7-34.4 -> 33.4-33.6 ms; 8: 86.4-87.0, 16: 86.4-86.9, 64: 86.1, 128: 86.8-87.3, 256: 91.8-92.0).  Plain lists
                                      // lose with it (C3 rank share of 8, 33 M-path batches: 13.07 -> 13.30 ms)
    uint32_t trace_chunk = 1;         // consecutive blocks per trace work item
    uint32_t shadow_any_hit = 1;      // shadow rays stop at their first hit (not in the counting kernels)
    uint32_t lds_tris = 1;            // the small meshes' triangles (the ground quad) are read from an LDS copy
    uint32_t first_lean = 1;          // round 0 (identical rays per wave) walks in the lean per-lane loop instead of voted steps
    uint32_t trace_events = 1;        // time every trace launch with its own hipEvent pair (cgpt_stats.dominant_ms)
    uint32_t path_order = 2;          // PathOrder of the path ids (trace_steps.hpp PathGrid): 2 pixel-major, 1 tile-major, 0 sample-major
    uint32_t retire_misses = 1;       // shade skips the state loads of later-round rays that hit nothing
    uint32_t bands_min_paths = 32u << 20;  // batches of fewer paths keep plain lists (1080p, 8 samples per call: 4.81 ms plain, 5.21 banded; 33 M-path batches: level)
    uint32_t sort = 0;                // 1: bin every round's ray lists by direction octant (measured in profiles/r02/k7_sort.md)
    uint32_t bands = 32;              // > 1: every round's ray lists ordered by image band (wf_shade: "Image bands"); at most kMaxKeys.  C3: 1 band 89.1-89.4 ms,
                                      // 8: 87.3-87.6, 16: 88.0-88.2, 32: 87.1-87.5, 64: 87.8-88.4 (profiles/r03/image_bands.md)
};

static uint32_t Gcd(uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; }
// smallest rot in [0, n_waves) with gcd(n_waves + rot, n_tiles) == 1 (device: next_block)
static uint32_t CoprimeRotation(uint32_t n_waves, uint32_t n_tiles)
{
    if (getenv("CGPT_WF_NO_ROTATION")) return 0u;
    for (uint32_t rot = 0; rot < n_waves && rot < 4096u; ++rot)
        if (Gcd(n_waves + rot, std::max(1u, n_tiles)) == 1u) return rot;
    return 0u;
}

static uint32_t EnvU32(const char* name, uint32_t fallback, uint32_t lo, uint32_t hi)
{
    const char* v = getenv(name);
    if (!v || !*v) return fallback;
    const long x = strtol(v, nullptr, 10);
    return (uint32_t)std::min<long>(std::max<long>(x, lo), hi);
}

struct WfHost {
    WfTuning tune;
    WfDev dev[kMaxPools] = {};
    hipStream_t streams[kMaxPools] = {};
    hipEvent_t acc_done[kMaxPools] = {};
    hipEvent_t begin = nullptr;
    uint32_t alloc_cap = 0, alloc_segs = 0, alloc_seg_cap = 0, alloc_pools = 0, alloc_overflow = 0, alloc_brute_levels = 0;
    bool alloc_sort = false;
    uint32_t n_cus = 0;
    uint32_t trace_blocks_per_cu[2][2] = {}, shade_blocks_per_cu[2][2] = {};   // trace: [COUNT][FIRST]; shade: [COUNT][BRUTE]
    size_t occupancy_lds = 0;
    unsigned long long* phase_stats = nullptr;   // CGPT_WF_PROFILE=1: step counts of the COUNT trace kernels, printed after the render
    // hipEvent pairs around every trace launch of the last render (roofline accounting: the dominant kernel's own duration)
    hipEvent_t* trace_ev = nullptr; uint32_t trace_ev_cap = 0, trace_ev_used = 0, trace_rounds = 0;
};

static void WfRelease(WfHost* h)
{
    for (uint32_t p = 0; p < kMaxPools; ++p) {
        WfDev& d = h->dev[p];
        (void)hipFree(d.A); (void)hipFree(d.B); (void)hipFree(d.C);
        (void)hipFree(d.st_tp); (void)hipFree(d.st_en); (void)hipFree(d.hit_flag); (void)hipFree(d.brute);
        (void)hipFree(d.list_ext); (void)hipFree(d.list_sh); (void)hipFree(d.seg_ext); (void)hipFree(d.seg_sh);
        (void)hipFree(d.seg_count); (void)hipFree(d.seg_prefix); (void)hipFree(d.plan); (void)hipFree(d.stack_overflow);
        (void)hipFree(d.seg_key_ext); (void)hipFree(d.seg_key_sh);
        d = WfDev{};
    }
    h->alloc_cap = 0; h->alloc_segs = 0; h->alloc_seg_cap = 0; h->alloc_pools = 0; h->alloc_overflow = 0; h->alloc_sort = false; h->alloc_brute_levels = 0;
}

void WavefrontFree(void* state)
{
    if (!state) return;
    WfHost* h = static_cast<WfHost*>(state);
    WfRelease(h);
    for (uint32_t p = 0; p < kMaxPools; ++p) {
        if (h->streams[p]) (void)hipStreamDestroy(h->streams[p]);
        if (h->acc_done[p]) (void)hipEventDestroy(h->acc_done[p]);
    }
    if (h->begin) (void)hipEventDestroy(h->begin);
    (void)hipFree(h->phase_stats);
    for (uint32_t i = 0; i < h->trace_ev_cap; ++i) (void)hipEventDestroy(h->trace_ev[i]);
    free(h->trace_ev);
    delete h;
}

// resident waves per SIMD of the later-round trace kernel (one 256-thread block = one wave on each of the CU's four SIMDs)
uint32_t WavefrontTraceWavesPerSimd(void* state)
{
    if (!state) return 0;
    const WfHost* h = static_cast<const WfHost*>(state);
    return std::min(h->tune.max_trace_blocks, h->trace_blocks_per_cu[0][0]) * (kTraceBlock / 256u);
}

// Sum of the trace launches' durations of the last render (and the round-0 launches' share); call after the render's device work has completed.
void WavefrontCollectTiming(void* state, double* trace_ms, uint32_t* trace_launches, double* round0_ms, uint32_t* round0_launches)
{
    *trace_ms = 0.0; *trace_launches = 0; *round0_ms = 0.0; *round0_launches = 0;
    if (!state) return;
    WfHost* h = static_cast<WfHost*>(state);
    for (uint32_t i = 0; i + 1u < h->trace_ev_used; i += 2u) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, h->trace_ev[i], h->trace_ev[i + 1u]) != hipSuccess) continue;
        *trace_ms += ms; *trace_launches += 1;
        if (h->trace_rounds && (i / 2u) % h->trace_rounds == 0u) { *round0_ms += ms; *round0_launches += 1; }   // launches are recorded batch by batch, round by round
    }
    h->trace_ev_used = 0;
}

// One knob table for the CGPT_WF_* environment variables (process-wide defaults, read when the context first needs its
// wavefront state) and cgpt_set_tuning (per context, any time between renders).
struct KnobDesc { const char* name; uint32_t WfTuning::*field; uint32_t lo, hi; };
static const KnobDesc kKnobs[] = {
    { "pools", &WfTuning::pools, 1, kMaxPools },           { "batch", &WfTuning::batch, 0, 4096 },
    { "max_batch", &WfTuning::max_batch, 1, 4096 },         { "pool_paths_mi", &WfTuning::pool_paths_mi, 1, 1024 },
    { "budget_gib", &WfTuning::budget_gib, 1, 256 },       { "refill", &WfTuning::refill_idle, 1, 64 },
    { "leaf_repeat", &WfTuning::leaf_repeat, 1, 65 },      { "inner_repeat", &WfTuning::inner_repeat, 1, 65 },
    { "obj_repeat", &WfTuning::obj_repeat, 1, 65 },        { "obj_shift", &WfTuning::obj_shift, 0, 6 },
    { "top_records", &WfTuning::top_records, 0, 4096 },     { "trace_blocks", &WfTuning::max_trace_blocks, 1, 64 },
    { "shade_chunk", &WfTuning::shade_chunk, 1, 256 },     { "shade_chunk_banded", &WfTuning::shade_chunk_banded, 1, 256 },     { "trace_chunk", &WfTuning::trace_chunk, 1, 256 },
    { "shadow_any_hit", &WfTuning::shadow_any_hit, 0, 1 },     { "trace_events", &WfTuning::trace_events, 0, 1 },
    { "sort", &WfTuning::sort, 0, 1 },                     { "path_order", &WfTuning::path_order, 0, 2 },                     { "retire_misses", &WfTuning::retire_misses, 0, 1 },
    { "lds_tris", &WfTuning::lds_tris, 0, 1 },             { "first_lean", &WfTuning::first_lean, 0, 1 },                     { "bands", &WfTuning::bands, 1, kMaxKeys },             { "bands_min_paths", &WfTuning::bands_min_paths, 0, 0x7FFFFFFF },
};

static WfHost* WfGetHost(cgpt_ctx* ctx)
{
    void** slot = CtxWavefrontSlot(ctx);
    if (*slot) return static_cast<WfHost*>(*slot);
    WfHost* fresh = new (std::nothrow) WfHost;
    if (!fresh) { CtxFail(ctx, CGPT_ERR_INVALID, "out of host memory"); return nullptr; }
    for (const KnobDesc& k : kKnobs) {
        char env[64] = "CGPT_WF_";
        size_t n = strlen(env);
        for (const char* c = k.name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
        env[n] = 0;
        fresh->tune.*(k.field) = EnvU32(env, fresh->tune.*(k.field), k.lo, k.hi);
    }
    hipError_t e = hipEventCreateWithFlags(&fresh->begin, hipEventDisableTiming);
    for (uint32_t p = 0; p < kMaxPools && e == hipSuccess; ++p) {
        e = hipStreamCreateWithFlags(&fresh->streams[p], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&fresh->acc_done[p], hipEventDisableTiming);
    }
    if (e != hipSuccess) {                                                    // a half-built state is never left in the context
        CtxFail(ctx, CGPT_ERR_HIP, "wavefront streams / events: %s", hipGetErrorString(e));
        WavefrontFree(fresh);
        return nullptr;
    }
    *slot = fresh;                                                            // fully initialised: owned by the context from here on (WavefrontFree)
    return fresh;
}

int WavefrontSetTuning(cgpt_ctx* ctx, const char* name, uint32_t value)
{
    WfHost* h = WfGetHost(ctx);
    if (!h) return CGPT_ERR_HIP;
    for (const KnobDesc& k : kKnobs)
        if (strcmp(k.name, name) == 0) {
            if (value < k.lo || value > k.hi) return CtxFail(ctx, CGPT_ERR_INVALID, "tuning knob %s: %u outside [%u, %u]", name, value, k.lo, k.hi);
            h->tune.*(k.field) = value;
            return CGPT_OK;
        }
    return CtxFail(ctx, CGPT_ERR_INVALID, "unknown tuning knob '%s'", name);
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

    if (!WfGetHost(ctx)) return -1;
    WfHost* h = static_cast<WfHost*>(*slot);
#ifdef CGPT_PHASE_CYCLES
    const bool want_phase_stats = getenv("CGPT_WF_PROFILE") != nullptr;
#else
    const bool want_phase_stats = count && getenv("CGPT_WF_PROFILE") != nullptr;
#endif
    if (want_phase_stats && !h->phase_stats) WF_TRY(hipMalloc((void**)&h->phase_stats, 32 * sizeof(unsigned long long)));
    if (h->phase_stats) WF_TRY(hipMemsetAsync(h->phase_stats, 0, 32 * sizeof(unsigned long long), stream));
    const uint32_t rows = args_in.n_rows;
    const uint32_t tiles_x = (args_in.width + 7u) / 8u, tiles_y = (rows + 7u) / 8u;
    const uint64_t n_pixels64 = (uint64_t)tiles_x * tiles_y * 64u;             // padded to whole 8x8 tiles
    const uint32_t pool_paths = h->tune.pool_paths_mi << 20;
    if (n_pixels64 > pool_paths) { CtxFail(ctx, CGPT_ERR_UNSUPPORTED, "band of %llu pixels exceeds the wavefront pool", (unsigned long long)n_pixels64); return -1; }
    const uint32_t n_pixels = (uint32_t)n_pixels64;
    const uint32_t rounds = (uint32_t)args_in.settings.max_ray_depth + 2u;    // extend rounds 0..max_depth, + the trailing shadow rays

    if (h->n_cus == 0) {
        int n_dev = 0, cus = 0;
        WF_TRY(hipGetDevice(&n_dev));
        WF_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, n_dev));
        h->n_cus = (uint32_t)cus;
    }
    const uint32_t n_cus = h->n_cus;
    const uint32_t top_records = std::min(h->tune.top_records, args_in.scene.n_top_records);
    const size_t trace_lds = trace_lds_bytes(top_records);
    // persistent grids = the resident capacity of the chip for each kernel
    if (h->occupancy_lds != trace_lds) {
        int b = 0;
        if (trace_lds > 48u * 1024u) {                                        // more dynamic LDS than the default limit: opt in per kernel
            WF_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trace<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trace_lds));
            WF_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trace<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trace_lds));
            WF_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trace<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trace_lds));
            WF_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trace<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trace_lds));
        }
        // shade: the round-0 and later-round instantiations share one grid size (one output segment per wave)
        int b2 = 0;
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (wf_trace<false, false>), kTraceBlock, trace_lds)); h->trace_blocks_per_cu[0][0] = (uint32_t)std::max(1, b);
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (wf_trace<false, true>), kTraceBlock, trace_lds)); h->trace_blocks_per_cu[0][1] = (uint32_t)std::max(1, b);
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (wf_trace<true, false>), kTraceBlock, trace_lds)); h->trace_blocks_per_cu[1][0] = (uint32_t)std::max(1, b);
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (wf_trace<true, true>), kTraceBlock, trace_lds)); h->trace_blocks_per_cu[1][1] = (uint32_t)std::max(1, b);
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (wf_shade<false, false>), 256, 0));
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b2, (wf_shade<false, true>), 256, 0)); h->shade_blocks_per_cu[0][0] = (uint32_t)std::max(1, std::min(b, b2));
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (wf_shade<true, false>), 256, 0));
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b2, (wf_shade<true, true>), 256, 0)); h->shade_blocks_per_cu[1][0] = (uint32_t)std::max(1, std::min(b, b2));
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (wf_shade<false, false, true>), 256, 0));
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b2, (wf_shade<false, true, true>), 256, 0)); h->shade_blocks_per_cu[0][1] = (uint32_t)std::max(1, std::min(b, b2));
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (wf_shade<true, false, true>), 256, 0));
        WF_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b2, (wf_shade<true, true, true>), 256, 0)); h->shade_blocks_per_cu[1][1] = (uint32_t)std::max(1, std::min(b, b2));
        h->occupancy_lds = trace_lds;
    }
    const dim3 block(256);
    const dim3 trace_grid_first(n_cus * std::min(h->tune.max_trace_blocks, h->trace_blocks_per_cu[count ? 1 : 0][1]));
    const dim3 trace_grid_later(n_cus * std::min(h->tune.max_trace_blocks, h->trace_blocks_per_cu[count ? 1 : 0][0]));
    const bool brute = args_in.settings.render_mode != 2u;                    // the render has TracePath paths (ref: Main.cpp:719-729)
    const uint32_t brute_levels = brute ? (uint32_t)args_in.settings.max_ray_depth + 1u : 0u;
    const dim3 shade_grid(n_cus * h->shade_blocks_per_cu[count ? 1 : 0][brute ? 1 : 0]);
    // one output segment per shade wave, sized for the most 64-item blocks a wave can be handed
    const uint32_t n_segs = n_cus * std::max({ h->shade_blocks_per_cu[0][0], h->shade_blocks_per_cu[1][0], h->shade_blocks_per_cu[0][1], h->shade_blocks_per_cu[1][1] }) * 4u;
    const uint32_t min_shade_waves = n_cus * std::min({ h->shade_blocks_per_cu[0][0], h->shade_blocks_per_cu[1][0], h->shade_blocks_per_cu[0][1], h->shade_blocks_per_cu[1][1] }) * 4u;

    // deep end of the traversal stacks: one dword per level beyond the LDS part and per thread of the largest trace grid
    const uint32_t max_trace_threads = n_cus * std::max({ h->trace_blocks_per_cu[0][0], h->trace_blocks_per_cu[0][1], h->trace_blocks_per_cu[1][0], h->trace_blocks_per_cu[1][1] }) * kTraceBlock;
    const uint32_t deep_levels = args_in.scene.stack_depth > kLdsStackLevels ? args_in.scene.stack_depth - kLdsStackLevels : 0u;
    const uint32_t overflow_words = std::max(1u, deep_levels * max_trace_threads);

    uint32_t shade_chunk = h->tune.shade_chunk;                              // set with the batch size below (banded lists take longer chunks)
    // ---- samples per batch and batches in flight ----
    // Big batches win: every bounce round is one pass of the persistent kernels over its ray list, the late rounds of a batch
    // are short, and a short list leaves the waves draining most of their life (measured at 1080p / 256 spp: 16 spp per
    // batch x 8 pools 118 ms, 64 x 4 108 ms, 128 x 2 104 ms; one batch of 256 with nothing to overlap its tails 129 ms).  288 GB
    // of HBM is what makes that possible: a pool is ~150 B per path, 128 spp of a 1080p frame is 265 M paths = 40 GB per pool.
    // So: the largest power-of-two batch up to max_batch that fits the pool limit and leaves at least two batches (two
    // pools overlap each other's tails), within a memory budget of half the free HBM (at most budget_gib).
    const size_t kBytesPerPath = 160 + 32 * (size_t)brute_levels;             // slots 96, state 32, lists 8, segments ~8-16; TracePath levels 32 each
    size_t free_b = 0, total_b = 0;
    WF_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t held = (size_t)h->alloc_pools * h->alloc_cap * (160 + 32 * (size_t)h->alloc_brute_levels);
    const size_t budget = std::min<size_t>((size_t)h->tune.budget_gib << 30, (free_b + held) / 2);
    uint32_t batch = h->tune.batch;
    if (batch == 0) {
        batch = 1;
        while (batch < h->tune.max_batch && (uint64_t)n_pixels * batch * 2u <= (uint64_t)pool_paths) batch *= 2u;
    }
    batch = std::max(1u, std::min({ batch, std::max(1u, args_in.n_samples / 2u), args_in.n_samples, pool_paths / n_pixels }));
    uint32_t cap = 0, n_pools = 0, seg_cap = 0, n_batches = 0;
    for (int attempt = 0;; ++attempt) {
        for (;;) {
            cap = n_pixels * batch;
            n_batches = (args_in.n_samples + batch - 1u) / batch;
            const uint32_t afford = (uint32_t)std::min<size_t>(kMaxPools, budget / ((size_t)cap * kBytesPerPath));
            n_pools = std::max(1u, std::min({ h->tune.pools, n_batches, afford }));
            if (batch == 1u || (afford >= 1u && n_pools >= std::min({ 2u, n_batches, h->tune.pools }))) break;
            batch /= 2u;                                                      // smaller batches: room for a second pool
        }
        const bool banded = !h->tune.sort && h->tune.bands > 1u && cap >= h->tune.bands_min_paths;   // (the last, shorter batch of a render may still fall below: it keeps this chunk)
        shade_chunk = banded ? h->tune.shade_chunk_banded : h->tune.shade_chunk;
        seg_cap = ((((cap + 63u) / 64u + shade_chunk - 1u) / shade_chunk + min_shade_waves - 1u) / min_shade_waves) * shade_chunk * 64u;   // whole chunks per wave
        if (h->alloc_overflow >= overflow_words && h->alloc_cap >= cap && h->alloc_segs >= n_segs && h->alloc_seg_cap >= seg_cap && h->alloc_pools >= n_pools && (!h->tune.sort || h->alloc_sort) && h->alloc_brute_levels >= brute_levels) break;
        WF_TRY(hipDeviceSynchronize());
        WfRelease(h);
        const size_t q = 2 * (size_t)cap * sizeof(float4);
        hipError_t err = hipSuccess;
        auto get = [&](void** ptr, size_t bytes) { if (err == hipSuccess) err = hipMalloc(ptr, bytes); };
        for (uint32_t p = 0; p < n_pools; ++p) {
            WfDev& d = h->dev[p];
            get((void**)&d.A, q); get((void**)&d.B, q); get((void**)&d.C, q);
            get((void**)&d.st_tp, (size_t)cap * sizeof(float4));
            get((void**)&d.st_en, (size_t)cap * sizeof(float4));
            get((void**)&d.hit_flag, (size_t)cap);
            if (brute_levels) get((void**)&d.brute, (size_t)brute_levels * cap * 2u * sizeof(float4));
            get((void**)&d.list_ext, (size_t)cap * sizeof(uint32_t));
            get((void**)&d.list_sh, (size_t)cap * sizeof(uint32_t));
            get((void**)&d.seg_ext, (size_t)n_segs * seg_cap * sizeof(uint32_t));
            get((void**)&d.seg_sh, (size_t)n_segs * seg_cap * sizeof(uint32_t));
            get((void**)&d.seg_count, 2 * (size_t)kMaxKeys * n_segs * sizeof(uint32_t));
            get((void**)&d.seg_prefix, 2 * (size_t)kMaxKeys * n_segs * sizeof(uint32_t));
            if (h->tune.sort) { get((void**)&d.seg_key_ext, (size_t)n_segs * seg_cap); get((void**)&d.seg_key_sh, (size_t)n_segs * seg_cap); }
            get((void**)&d.plan, (2 + 2 * (size_t)kMaxKeys) * sizeof(uint32_t));
            get((void**)&d.stack_overflow, (size_t)overflow_words * sizeof(uint32_t));
        }
        if (err == hipSuccess) {
            h->alloc_cap = cap; h->alloc_segs = n_segs; h->alloc_seg_cap = seg_cap; h->alloc_pools = n_pools; h->alloc_overflow = overflow_words; h->alloc_sort = h->tune.sort != 0u; h->alloc_brute_levels = brute_levels;
            break;
        }
        (void)hipGetLastError();                                              // out of memory: give everything back and ask for half
        WfRelease(h);
        if (batch == 1u || attempt >= 8) { CtxFail(ctx, CGPT_ERR_HIP, "wavefront pools: %s", hipGetErrorString(err)); return -1; }
        batch /= 2u;
    }

    // event pairs for the trace launches of this render
    const uint32_t ev_needed = 2u * n_batches * rounds;
    if (h->trace_ev_cap < ev_needed) {
        hipEvent_t* grown = static_cast<hipEvent_t*>(realloc(h->trace_ev, (size_t)ev_needed * sizeof(hipEvent_t)));
        if (!grown) { CtxFail(ctx, CGPT_ERR_INVALID, "out of host memory"); return -1; }
        h->trace_ev = grown;
        for (; h->trace_ev_cap < ev_needed; ++h->trace_ev_cap) WF_TRY(hipEventCreate(&h->trace_ev[h->trace_ev_cap]));
    }
    h->trace_ev_used = 0; h->trace_rounds = rounds;

    // one-time host setup is over: the render's device time starts here (cgpt_stats.kernel_ms).  The pool streams start after
    // whatever the caller queued on the context's stream.
    WF_TRY(hipEventRecord(CtxStartEvent(ctx), stream));
    WF_TRY(hipEventRecord(h->begin, stream));
    for (uint32_t p = 0; p < n_pools; ++p) WF_TRY(hipStreamWaitEvent(h->streams[p], h->begin, 0));

    int launches = 0;
    DevRenderArgs args = args_in;
    const TraceTune tt = { h->tune.refill_idle, h->tune.inner_repeat, h->tune.leaf_repeat, h->tune.obj_repeat, h->tune.obj_shift, top_records, h->tune.shadow_any_hit, h->tune.lds_tris, 0u, h->tune.first_lean };
    uint32_t k = 0;
    for (uint32_t done = 0; done < args_in.n_samples; done += batch, ++k) {
        const uint32_t p = k % n_pools;
        hipStream_t st = h->streams[p];
        const uint32_t bn = std::min(batch, args_in.n_samples - done);
        const uint32_t bfirst = args_in.first_sample + done;
        WfDev wf = h->dev[p];
        wf.cap = h->alloc_cap; wf.g.n_pixels = n_pixels; wf.n_paths = n_pixels * bn;
#ifdef CGPT_PHASE_CYCLES
        wf.phase_stats = h->phase_stats;
#else
        wf.phase_stats = count ? h->phase_stats : nullptr;
#endif
        wf.rot_trace[0] = CoprimeRotation(trace_grid_later.x * (kTraceBlock / 64u), std::max(1u, tiles_x * tiles_y / h->tune.trace_chunk));
        wf.rot_trace[1] = CoprimeRotation(trace_grid_first.x * (kTraceBlock / 64u), std::max(1u, tiles_x * tiles_y / h->tune.trace_chunk));
        wf.rot_shade = CoprimeRotation(shade_grid.x * 4u, std::max(1u, tiles_x * tiles_y / shade_chunk));
        wf.shade_chunk = shade_chunk; wf.trace_chunk = h->tune.trace_chunk;
        wf.g.tiles_x = tiles_x; wf.g.div_tiles_x = MakeFastDiv(tiles_x); wf.g.div_n_pixels = MakeFastDiv(n_pixels);
        wf.g.n_samples = bn; wf.g.div_samples = MakeFastDiv(bn); wf.g.order = h->tune.path_order; wf.n_segs = h->alloc_segs; wf.seg_cap = h->alloc_seg_cap;
        // segments of waves that a smaller shade grid does not launch must read as empty
        wf.n_keys = h->tune.sort && wf.seg_key_ext ? 8u : 1u;
        wf.n_bands = wf.n_keys == 1u && h->tune.bands > 1u && wf.n_paths >= h->tune.bands_min_paths ? h->tune.bands : 1u;   // short lists: nothing to order, and every band is a run per segment to plan and copy
        if (wf.n_bands > 1u) {
            wf.n_keys = wf.n_bands;
            wf.band_magic = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (((uint64_t)wf.n_bands << 32) + wf.n_paths - 1u) / wf.n_paths);   // ceil(2^32 * bands / paths)
        }
        wf.retire_misses = h->tune.retire_misses && args_in.settings.debug_mode == 0u ? 1u : 0u;
        if (k < n_pools) WF_TRY(hipMemsetAsync(wf.seg_count, 0, 2 * (size_t)kMaxKeys * wf.n_segs * sizeof(uint32_t), st));
        for (uint32_t r = 0; r < rounds; ++r) {
            const bool first = r == 0u;
            if (h->tune.trace_events) WF_TRY(hipEventRecord(h->trace_ev[h->trace_ev_used++], st));
            const dim3 trace_grid = first ? trace_grid_first : trace_grid_later;
            if (count && first) hipLaunchKernelGGL((wf_trace<true, true>), trace_grid, dim3(kTraceBlock), trace_lds, st, args, wf, bfirst, tt);
            else if (count) hipLaunchKernelGGL((wf_trace<true, false>), trace_grid, dim3(kTraceBlock), trace_lds, st, args, wf, bfirst, tt);
            else if (first) hipLaunchKernelGGL((wf_trace<false, true>), trace_grid, dim3(kTraceBlock), trace_lds, st, args, wf, bfirst, tt);
            else hipLaunchKernelGGL((wf_trace<false, false>), trace_grid, dim3(kTraceBlock), trace_lds, st, args, wf, bfirst, tt);
            if (h->tune.trace_events) WF_TRY(hipEventRecord(h->trace_ev[h->trace_ev_used++], st));
            ++launches;
            if (r + 1u < rounds) {
                if (brute) {
                    if (count && first) hipLaunchKernelGGL((wf_shade<true, true, true>), shade_grid, block, 0, st, args, wf, bfirst);
                    else if (count) hipLaunchKernelGGL((wf_shade<true, false, true>), shade_grid, block, 0, st, args, wf, bfirst);
                    else if (first) hipLaunchKernelGGL((wf_shade<false, true, true>), shade_grid, block, 0, st, args, wf, bfirst);
                    else hipLaunchKernelGGL((wf_shade<false, false, true>), shade_grid, block, 0, st, args, wf, bfirst);
                }
                else if (count && first) hipLaunchKernelGGL((wf_shade<true, true>), shade_grid, block, 0, st, args, wf, bfirst);
                else if (count) hipLaunchKernelGGL((wf_shade<true, false>), shade_grid, block, 0, st, args, wf, bfirst);
                else if (first) hipLaunchKernelGGL((wf_shade<false, true>), shade_grid, block, 0, st, args, wf, bfirst);
                else hipLaunchKernelGGL((wf_shade<false, false>), shade_grid, block, 0, st, args, wf, bfirst);
                hipLaunchKernelGGL(wf_plan, dim3(2u * wf.n_keys), dim3(256), 0, st, wf);
                hipLaunchKernelGGL(wf_gather, dim3(std::min(2u * wf.n_segs, n_cus * 16u)), block, 0, st, wf);
                launches += 3;
            }
        }
        // accumulate in sample order: batch k after batch k-1
        if (k > 0) WF_TRY(hipStreamWaitEvent(st, h->acc_done[(k - 1u) % n_pools], 0));
        hipLaunchKernelGGL(wf_accumulate, dim3(std::min((n_pixels + 255u) / 256u, n_cus * 8u)), block, 0, st, args, wf, bfirst, bn);
        ++launches;
        WF_TRY(hipEventRecord(h->acc_done[p], st));
        WF_TRY(hipGetLastError());
    }
    // the context's stream continues after the last accumulate (which transitively follows all the others)
    if (k > 0) WF_TRY(hipStreamWaitEvent(stream, h->acc_done[(k - 1u) % n_pools], 0));
#ifdef CGPT_PHASE_CYCLES
    if (!count && h->phase_stats) {                                           // diagnostic build: cycles of the later-round trace waves by phase
        unsigned long long ps[32];
        WF_TRY(hipStreamSynchronize(stream));
        WF_TRY(hipMemcpy(ps, h->phase_stats, sizeof(ps), hipMemcpyDeviceToHost));
        const double tot = (double)ps[8];
        fprintf(stderr, "[wf cycles] later-round trace: %llu waves, %.0f Mcyc/wave | refill %.3f inner %.3f leaf %.3f object %.3f other %.3f | cycles per wave-step: inner %.0f (%.1f lanes, %.2f from LDS) leaf %.0f (%.1f lanes) object %.0f (%.1f lanes)\n",
                ps[20], ps[20] ? tot / ps[20] / 1e6 : 0.0, ps[9] / tot, ps[10] / tot, ps[11] / tot, ps[12] / tot, 1.0 - (ps[9] + ps[10] + ps[11] + ps[12]) / tot,
                ps[13] ? (double)ps[10] / ps[13] : 0.0, ps[13] ? (double)ps[16] / ps[13] : 0.0, ps[16] ? (double)ps[19] / ps[16] : 0.0,
                ps[14] ? (double)ps[11] / ps[14] : 0.0, ps[14] ? (double)ps[17] / ps[14] : 0.0,
                ps[15] ? (double)ps[12] / ps[15] : 0.0, ps[15] ? (double)ps[18] / ps[15] : 0.0);
        fprintf(stderr, "[wf cycles] inner lane-steps served from global memory: %llu; both children missed %.3f; both missed on the x,y slabs alone %.3f; on the x slab alone %.3f\n",
                ps[21], ps[21] ? (double)ps[22] / ps[21] : 0.0, ps[21] ? (double)ps[23] / ps[21] : 0.0, ps[21] ? (double)ps[24] / ps[21] : 0.0);
    }
#endif
    if (count && h->phase_stats) {                                            // development aid: how full the steps were
        unsigned long long ps[8];
        WF_TRY(hipStreamSynchronize(stream));
        WF_TRY(hipMemcpy(ps, h->phase_stats, sizeof(ps), hipMemcpyDeviceToHost));
        DevCounters c;
        WF_TRY(hipMemcpy(&c, args_in.counters, sizeof(c), hipMemcpyDeviceToHost));
        fprintf(stderr, "[wf profile] rays %llu | inner: %llu wave steps, %.1f lanes/step | leaf: %llu wave steps, %.1f lanes/step | object: %llu wave steps, %.1f lanes/step | votes %llu refills %llu\n",
                c.traced_rays, ps[0], ps[0] ? (double)c.inner_steps / ps[0] : 0.0, ps[1], ps[1] ? (double)ps[6] / ps[1] : 0.0,
                ps[2], ps[2] ? (double)ps[3] / ps[2] : 0.0, ps[4], ps[5]);
    }
#undef WF_TRY
    return launches;
}

}  // namespace cgpt
