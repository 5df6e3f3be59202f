// trace_steps.hpp -- the voted traversal steps shared by the persistent kernels (gfx950): wf_trace (wavefront pipeline) and
// pt_persistent (whole paths in one launch).
//
// A lane's ray is in one of the traversal states encoded in its code -- at an inner node (code = child-pair record), at a leaf
// triangle (bit 31), at an object boundary (kStartObject: begin scene object cur_obj, or finish the ray when there is none left).
// Every scheduling iteration the wave votes for the state most lanes are in and runs that state's step, repeating it while
// enough lanes stay in the state.  The steps are written without divergent branches: both children of a node are tested with
// packed f32 math, the next code / stack pointer / object index are selects, the far child is stored to the free LDS slot above
// the stack top unconditionally (it only counts when the stack pointer moves), and the entry below the stack pointer is read
// next to the node, so a pop is a select.  The rare cases -- an axis-parallel ray in the wave (the NaN-exact slab test, SURVEY
// A-18) or a stack deeper than the LDS part -- take a general step with the same results.
// ref: Source/BVH.cpp:61-127 (Traverse), Source/Main.cpp:299-316 (IntersectScene).
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.h"
#include "fast_div.h"
#include "rt_device.hpp"

namespace cgpt {
namespace dev {

static constexpr uint32_t kStartObject = 0x40000000u;   // traversal code: "begin the next object of the scene" (record codes are < 2^26)
static constexpr uint32_t kIdle = 0x40000001u;           // traversal code of a lane without a ray
#ifndef CGPT_LDS_STACK_LEVELS
#define CGPT_LDS_STACK_LEVELS 16
#endif
static constexpr uint32_t kLdsStackLevels = CGPT_LDS_STACK_LEVELS;   // traversal stack levels kept in LDS; deeper entries overflow to HBM
#ifndef CGPT_TRACE_BLOCK
#define CGPT_TRACE_BLOCK 256
#endif
static constexpr uint32_t kTraceBlock = CGPT_TRACE_BLOCK;   // threads per block of the trace kernels (256: five blocks per CU, each with its own
                                                         // copy of the tree top; 1024: one block per CU sharing one large copy -- profiles/r02/experiments.md)
static constexpr uint32_t kRing = 128;                   // per-wave LDS ring of work items: up to 63 left over + one 64-item block
static constexpr uint32_t kLdsObjects = 31;              // scene objects whose trace records are mirrored in LDS (+ one end marker)
static constexpr uint32_t kKindEnd = 3u;                 // object kind of the end marker (0 mesh, 1 sphere, 2 plane: cgpt_object_kind)
static constexpr uint32_t kTopStride = 20;               // dwords per record in the LDS copy of the top of the tree: 80 bytes, so that
                                                         // consecutive records start 20 banks apart and 16-byte reads of random
                                                         // records spread over all 32 banks (64-byte records would use 8 of them)
static constexpr uint32_t kLdsTopMax = 127;              // 7 full levels of one tree; with stacks, rings and object table 29.6 KB per block: 5 blocks per CU
                                                         // (measured on MI355X: 127 records +2.3 %, 166 the same with 12 stack levels, 255 -2 %: 4 blocks per CU)

struct TraceTune { uint32_t refill_idle, inner_repeat, leaf_repeat, obj_repeat, obj_shift, top_records, shadow_any_hit, lds_tris, tail_lanes, first_lean; };

// lane index from the execution-mask count, recomputed at every use (two instructions): as threadIdx.x & 63 it is loop-invariant and the
// optimiser keeps it in a register through the whole kernel
__device__ __forceinline__ uint32_t lane_id()
{
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ uint32_t rank_in_mask(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Work distribution of the persistent kernels: the 64-item blocks of a list are dealt out in rows of n_waves blocks, wave w
// taking position (w + row * rot) mod n_waves of every row.  With rot = 0 that is a plain stride of n_waves -- and when the
// stride shares a large factor with the number of 8x8 tiles of the band (3840 tiles, 6144 waves: gcd 768) a wave gets the same
// five screen tiles of every sample, all sky or all mesh: measured +15 % render time on that band.  The host picks the
// smallest rot that makes n_waves + rot coprime to the tile count, so a wave's blocks walk over the whole band.
struct BlockWalk { uint32_t row_base, pos; };
__device__ __forceinline__ BlockWalk first_block(uint32_t wave) { BlockWalk b; b.row_base = 0u; b.pos = wave; return b; }
__device__ __forceinline__ uint32_t block_of(BlockWalk b) { return b.row_base + b.pos; }
__device__ __forceinline__ void next_block(BlockWalk& b, uint32_t n_waves, uint32_t rot)
{
    b.row_base += n_waves;
    b.pos += rot;
    if (b.pos >= n_waves) b.pos -= n_waves;
}

// Work distribution of the persistent path kernel (measured in wf_trace too and not adopted there: profiles/r02/experiments.md):
// items (path ids) are handed out by kWorkCounters
// atomic counters, each owning a contiguous slice of the index range (a wave starts at counter wave % kWorkCounters and moves on
// when a slice is used up).  A fetch takes `coarse` items while plenty are left and, near the end of a slice, exactly as many as
// the wave has idle lanes.  Why not the static deal above: it balances a launch in which every wave gets thousands of 64-item
// blocks, but a launch always ENDS in a drain -- the waves that drew the long rays keep going while the rest of the chip idles --
// and a launch with few items per wave (a later bounce round, a one-sample frame: 32 400 tiles over 4 096 waves) is mostly
// drain.  With fine fetches near the end, the last items are spread over all waves instead of sitting 64 deep in a few.  One
// returning atomic per fetch stays far below the rate one L2 word serves (~88 per microsecond; eight words).
static constexpr uint32_t kWorkCounters = 8;             // counters of one launch, 32 bytes apart
struct WorkFetch {                                        // wave-uniform
    uint32_t counter, tried, loc_next, loc_end;           // [loc_next, loc_end): items fetched and not yet handed to a lane
    bool fine, exhausted;
};
__device__ __forceinline__ WorkFetch work_begin(uint32_t wave)
{
    WorkFetch w; w.counter = wave % kWorkCounters; w.tried = 0; w.loc_next = 0; w.loc_end = 0; w.fine = false; w.exhausted = false; return w;
}
// refills [loc_next, loc_end) when it is empty (wave-uniform control flow; lane 0 does the atomic)
__device__ __forceinline__ void work_fetch(WorkFetch& w, uint32_t* counters, uint32_t n_items, uint32_t coarse, uint32_t fine_below, uint32_t n_need)
{
    const uint32_t slice = (n_items + kWorkCounters - 1u) / kWorkCounters;
    while (!w.exhausted && w.loc_next == w.loc_end) {
        const uint32_t begin = min(w.counter * slice, n_items), end = min(begin + slice, n_items);
        const uint32_t want = w.fine ? n_need : max(coarse, n_need);
        uint32_t v = 0;
        if ((threadIdx.x & 63u) == 0u) v = atomicAdd(&counters[w.counter * 8u], want);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v < end - begin) {
            w.loc_next = begin + v; w.loc_end = min(w.loc_next + want, end);
            w.fine = (end - w.loc_end) < fine_below;
            w.tried = 0;
        } else {                                                              // this slice is used up: the next counter
            w.counter = w.counter + 1u == kWorkCounters ? 0u : w.counter + 1u;
            w.fine = false;
            w.exhausted = ++w.tried == kWorkCounters;
        }
    }
}
// host side: fetch sizes for a launch of n_items over n_waves waves (coarse: ~32 fetches per wave, whole 64-item blocks, at most 8
// of them -- with tile-major ids 256-512 ids measured best, 4 032 ids 4 % slower, 16 384 ids half the speed; fine fetches for the last fine_rounds items per lane of the grid, spread over the counters)
inline void work_sizes(uint32_t n_items, uint32_t n_waves, uint32_t fine_rounds, uint32_t chunk_override, uint32_t& coarse, uint32_t& fine_below)
{
    const uint32_t per_fetch = n_items / (n_waves * 32u);
    coarse = chunk_override ? chunk_override * 64u : (per_fetch >= 512u ? 512u : (per_fetch >= 64u ? per_fetch / 64u * 64u : (per_fetch > 16u ? per_fetch : 16u)));
    const uint64_t fb = (uint64_t)n_waves * 64u * fine_rounds / kWorkCounters;
    fine_below = (uint32_t)(fb > 0x7FFFFFFFull ? 0x7FFFFFFFull : (fb ? fb : 1ull));
}

// Streaming data (slots, path state, lists) goes through non-temporal loads / stores: the BVH and the triangles are what should
// stay in the 4 MB per-XCD L2.
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p)
{
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    float4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r;
}
__device__ __forceinline__ void st_stream(float4* p, float4 v)
{
    nt_f4 x; x.x = v.x; x.y = v.y; x.z = v.z; x.w = v.w;
    __builtin_nontemporal_store(x, reinterpret_cast<nt_f4*>(p));
}
__device__ __forceinline__ uint32_t ld_stream(const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }

// ---- path id <-> (sample, pixel) ----------------------------------------------------------------------------------------------
// Pixel indices enumerate 8x8 screen tiles in row-major tile order, row-major inside a tile (64 consecutive indices = one tile);
// tiles on the right / bottom edge are padded, the padded indices are not pixels.  Path ids of a batch of n_samples samples:
//   pixel-major (default): id = pixel * n_samples + sample                -- the 64 lanes of a wave are samples of the same pixel (or of
//                          64 / n_samples neighbouring ones): the reference does not jitter (SURVEY A-14), so their primary rays are
//                          identical and the wave walks the tree in lockstep; after the first bounce the rays still start from one
//                          surface point.  Active lanes per traversal step 34 -> 38 (inner), 35 -> 40 (leaf); 256 spp frame: wavefront
//                          107.4 -> 98.3 ms, persistent kernel 120.9 -> 106.1 ms (profiles/r02/experiments.md)
//   tile-major:            id = (tile * n_samples + sample) * 64 + lane   -- a wave is one 8x8 tile, the samples of a tile are
//                          neighbouring waves (persistent kernel 137.3 -> 119.6 ms against sample-major)
//   sample-major:          id = sample * n_pixels + pixel
enum PathOrder : uint32_t { kSampleMajor = 0, kTileMajor = 1, kPixelMajor = 2 };
struct PathGrid {
    uint32_t n_pixels;                 // pixel indices of the band, padded to whole 8x8 tiles
    uint32_t tiles_x;                  // 8x8 tiles per row
    FastDiv div_tiles_x, div_n_pixels;
    uint32_t n_samples;                // samples of this batch
    FastDiv div_samples;
    uint32_t order;                    // PathOrder
};
__device__ __forceinline__ void path_split(const PathGrid& g, uint32_t pid, uint32_t& sample, uint32_t& pixel)
{
    if (g.order == kPixelMajor) {
        pixel = fast_div(pid, g.div_samples);
        sample = pid - pixel * g.n_samples;
    } else if (g.order == kTileMajor) {
        const uint32_t b = pid >> 6, tile = fast_div(b, g.div_samples);
        sample = b - tile * g.n_samples;
        pixel = (tile << 6) | (pid & 63u);
    } else {
        sample = fast_div(pid, g.div_n_pixels);
        pixel = pid - sample * g.n_pixels;
    }
}
__device__ __forceinline__ uint32_t path_id(const PathGrid& g, uint32_t sample, uint32_t pixel)
{
    if (g.order == kPixelMajor) return pixel * g.n_samples + sample;
    return g.order == kTileMajor ? ((((pixel >> 6) * g.n_samples + sample) << 6) | (pixel & 63u)) : sample * g.n_pixels + pixel;
}
__device__ __forceinline__ bool pixel_of_index(const DevRenderArgs& a, const PathGrid& g, uint32_t p, uint32_t& px, uint32_t& py, uint32_t& local_row)
{
    const uint32_t tile = p >> 6, l = p & 63u;
    const uint32_t ty = fast_div(tile, g.div_tiles_x);
    const uint32_t tx = tile - ty * g.tiles_x;
    px = tx * 8u + (l & 7u);
    local_row = ty * 8u + (l >> 3);
    py = GlobalRow(local_row, a.band_first, a.band_h, a.band_stride);
    return px < a.width && local_row < a.n_rows;
}
// primary ray + RNG stream of path `pid` (ref: Main.cpp:713-716, Camera::GetRay :133-140); false for the padded indices
__device__ __forceinline__ bool primary_ray(const DevRenderArgs& args, const PathGrid& g, uint32_t pid, uint32_t batch_first, Ray& ray, uint32_t& rng, uint32_t& px_out)
{
    uint32_t s, p;
    path_split(g, pid, s, p);
    uint32_t px, py, local_row;
    if (!pixel_of_index(args, g, p, px, py, local_row)) return false;
    px_out = px;
    rng = pcg_seed(py * args.width + px, batch_first + s, args.seed);
    ray = camera_ray(args.camera, (float)px * (1.0f / (float)args.width), (float)py * (1.0f / (float)args.height));
    return true;
}

// ---- per-block LDS layout + the traversal state of a lane ---------------------------------------------------------------------
// LDS: traversal stacks (kLdsStackLevels x kTraceBlock dwords, stack[level][thread]: conflict-free), one ring of kRing dwords per
// wave, IntersectScene's object list (ref: Main.cpp:303-315; 8 dwords per object + an end marker, so that moving on to the next
// object is an LDS read inside the step that finishes the previous one), and the top of the trees (records [0, n_top),
// breadth-first: device_scene.h "record order" -- every ray walks it, and reading it here takes those fetches off the texture
// path).
// LDS pointers carry their address space in the type: handed around as plain pointers, the optimiser merged the LDS and the HBM
// fetch of a child-pair record into one generic flat_load (waits on both counters, +14 % trace time).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) f4v lds_f4v;
typedef __attribute__((address_space(3))) u2v lds_u2v;

struct TravCtx {
    const DevScene* sc;
    lds_u32* stack;           // this thread's column of the LDS stacks
    lds_u32* lds_base;        // start of the block's dynamic LDS (= stack column of thread 0)
    lds_u32* ring;            // this wave's ring
    const lds_u32* objtab;
    const lds_u32* top_cache;
    const lds_u32* tri_cache;  // LDS copy of the small meshes' leaf-triangle records [0, n_lds_tris)
    uint32_t n_lds_tris;
    uint32_t* deep_block;     // this block's columns of the HBM overflow levels (wave-uniform: the thread's column is found from its LDS stack
    uint32_t deep_stride;     //  address when a deep entry is touched, which is rare -- no per-thread pointer is kept in registers)
    uint32_t n_top;
    uint32_t first_code;      // code a fresh ray starts with (root of object 0 if that is a mesh)
    bool tab;                 // the object table is in LDS (n_objects <= kLdsObjects)
};
__host__ __device__ inline size_t trace_lds_bytes(uint32_t top_records)
{
    return ((size_t)kLdsStackLevels * kTraceBlock + (kTraceBlock / 64) * kRing + (kLdsObjects + 1) * 8 + kLdsTrisMax * 12 + (size_t)top_records * kTopStride) * sizeof(uint32_t);
}

// fills the block's LDS tables and returns the per-thread context (all threads of the block call it; ends in a barrier)
__device__ __forceinline__ TravCtx trav_setup(const DevScene& sc, uint32_t* lds_generic, uint32_t top_records, uint32_t* overflow_base, uint32_t grid_threads, uint32_t lds_tris = 1u)
{
    TravCtx c;
    c.sc = &sc;
    lds_u32* const lds = (lds_u32*)lds_generic;
    c.stack = lds + threadIdx.x;
    c.lds_base = lds;
    c.ring = lds + kLdsStackLevels * kTraceBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * kRing;   // wave-uniform: a scalar register
    lds_u32* const objtab = lds + kLdsStackLevels * kTraceBlock + (kTraceBlock / 64u) * kRing;
    c.tab = sc.n_objects <= kLdsObjects;                                      // otherwise the object step reads HBM and nothing is folded
    // Word 7 of an entry (free in every kind) = the traversal code a ray arriving at this object continues with: the mesh's root, or
    // kStartObject (analytic primitive / end of the list: the object step takes over) -- one dword read per voted step, no select.
    if (c.tab) {
        const uint32_t n_words = sc.n_objects * 8u;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(sc.obj_trace);
        for (uint32_t i = threadIdx.x; i < n_words + 8u; i += kTraceBlock) {
            uint32_t v = i < n_words ? src[i] : (i == n_words ? kKindEnd : 0u);
            if ((i & 7u) == 7u) v = (i < n_words && src[i - 7u] == 0u) ? src[i - 6u] : kStartObject;
            objtab[i] = v;
        }
    } else if (threadIdx.x == 0u) objtab[7] = kStartObject;                   // the one word the fetch sequences read when the table is not in LDS
    lds_u32* const tri_cache = objtab + (kLdsObjects + 1u) * 8u;
    c.n_lds_tris = lds_tris ? min(sc.n_small_tris, kLdsTrisMax) : 0u;
    // LDS copy of a triangle record: dwords 0..7 as in HBM, the tail {e2.z, tri_idx, last} moved from dwords 9..11 to 8..10, where it is
    // 16-byte aligned and one ds_read_b96 (the HBM record keeps its tail at byte 36 for the 12-byte global load)
    for (uint32_t i = threadIdx.x; i < c.n_lds_tris * 12u; i += kTraceBlock) {
        const uint32_t w = i % 12u;
        tri_cache[i] = w == 11u ? 0u : reinterpret_cast<const uint32_t*>(sc.tri_leaf)[w >= 8u ? i + 1u : i];
    }
    c.tri_cache = tri_cache;
    lds_u32* const top_cache = tri_cache + kLdsTrisMax * 12u;
    c.n_top = min(top_records, sc.n_top_records);
    for (uint32_t i = threadIdx.x; i < c.n_top * 16u; i += kTraceBlock)
#ifdef CGPT_NODE_SOA
        top_cache[(i >> 4) * kTopStride + (i & 15u)] = reinterpret_cast<const uint32_t*>(sc.node_pairs)[(size_t)(i & 15u) * sc.n_pair_records + (i >> 4)];
#else
        top_cache[(i >> 4) * kTopStride + (i & 15u)] = reinterpret_cast<const uint32_t*>(sc.node_pairs)[i];
#endif
    __syncthreads();
    c.objtab = objtab; c.top_cache = top_cache;
    c.first_code = kStartObject;
    if (c.tab && objtab[0] == 0u) c.first_code = objtab[1];
    c.deep_block = overflow_base + blockIdx.x * kTraceBlock;
    c.deep_stride = grid_threads;
    return c;
}

struct Trav {                 // one lane's ray in flight
    V3 d;
    RaySlab rs;               // origin and 1/d as the slab test wants them: o = {oxy.x, oxy.y, ozi.x}
    float t;
    uint32_t obj, tri, depth; // closest hit so far (ref: Primitives.h:77-82 payload)
    uint32_t cur_obj, code, sp;
    uint32_t fast_levels;     // stack depths at which this ray may take the branch-free inner step: kLdsStackLevels, or 0 for an axis-parallel
                              // direction (NaN-exact slab test) -- one compare per step decides between the two forms of the step
    // Bit 31 of `depth` = "any hit": a shadow ray, whose caller only asks whether ANYTHING was hit (ref: Main.cpp:454-463), so the ray may
    // stop at its first hit; the counting kernels walk on to the end, as the reference does, to keep its step counts.  (The flag shares
    // the register of a count that never comes near 2^31; the steps only ever add to it.)
};
static constexpr uint32_t kAnyHitBit = 0x80000000u;
__device__ __forceinline__ bool trav_any_hit(const Trav& r) { return (int32_t)r.depth < 0; }
__device__ __forceinline__ uint32_t trav_depth(const Trav& r) { return r.depth & ~kAnyHitBit; }
__device__ __forceinline__ V3 trav_origin(const Trav& r) { return mk(r.rs.oxy.x, r.rs.oxy.y, r.rs.ozi.x); }

// a fresh IntersectScene call for this lane (Ray ctor, ref: Primitives.h:64)
__device__ __forceinline__ void trav_start(const TravCtx& c, Trav& r, V3 o, V3 d, float t, uint32_t obj, uint32_t tri, uint32_t depth)
{
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.d = d; r.t = t; r.obj = obj; r.tri = tri; r.depth = depth;
    r.fast_levels = has_infinite_component(inv) ? 0u : kLdsStackLevels;
    r.rs = make_ray_slab(o, inv);
    r.cur_obj = 0; r.code = c.first_code; r.sp = 0;
}

__device__ __forceinline__ void load_pair_lds(const lds_u32* top_cache, uint32_t code, NodePair& n)
{
    static_assert(kTopStride == 20u, "record stride as shifts");
    const lds_u32* rec = top_cache + __umul24(code, kTopStride);             // code < n_top: a full-rate 24-bit multiply (v_mul_lo_u32 is not)
    n.q0 = *reinterpret_cast<const lds_f4v*>(rec);
    n.q1 = *reinterpret_cast<const lds_f4v*>(rec + 4);
    n.q2 = *reinterpret_cast<const lds_f4v*>(rec + 8);
    const u2v codes = *reinterpret_cast<const lds_u2v*>(rec + 14);
    n.lcode = codes.x; n.rcode = codes.y;
}

__device__ __forceinline__ LeafTri load_leaf_tri_lds(const lds_u32* tri_cache, uint32_t index)
{
    const lds_u32* rec = tri_cache + __umul24(index, 12u);                    // index * 12 dwords (index < n_lds_tris)
    const f4v a = *reinterpret_cast<const lds_f4v*>(rec), b = *reinterpret_cast<const lds_f4v*>(rec + 4);
    const f4v cc = *reinterpret_cast<const lds_f4v*>(rec + 8);               // {e2.z, tri_idx, last, -}: trav_setup
    LeafTri t;
    t.v0 = mk(a.x, a.y, a.z); t.e1 = mk(a.w, b.x, b.y); t.e2 = mk(b.z, b.w, cc.x);
    t.tri_idx = __float_as_uint(cc.y); t.last = __float_as_uint(cc.z) != 0u;
    return t;
}

// ---- one fetch sequence per voted step (experiment build -DCGPT_FETCH_SEQ; measured slower, not the default) -----------------
// A voted step needs a record (child pair or leaf triangle: from the LDS copy for some lanes, from HBM for the others), the stack
// entry below the lane's stack pointer and the object-table entry after the lane's object.  Left to the compiler the two halves of
// the record fetch share their destination registers, and its wait-count bookkeeping is per register, not per lane: it waits for the
// HBM loads before it lets the LDS reads of the OTHER lanes issue, and reads the stack entry after both -- global latency + two LDS
// round trips in series, every step.  Here all of them are issued back to back under their lane masks and waited for once.
// Measured (profiles/r03/ab_fetch_sequence.txt): later-round trace 70.6-71.6 ms with the compiler's serial fetch, 72.9-73.6 ms with
// this one, with or without skipping the empty half.  A step is not bound by its own latency (five waves per SIMD cover it); what the
// sequence costs is the work the compiler used to put in the loads' shadow, which now sits behind the wait.
typedef uint32_t u3v __attribute__((ext_vector_type(3)));
typedef uint64_t exec_mask_t;
template <typename P> __device__ __forceinline__ uint32_t lds_addr(P* p) { return (uint32_t)(uintptr_t)p; }

struct StepAux { uint32_t top, next; };                                       // stack entry below sp; the code object cur_obj + 1 begins with (objtab word 7)

__device__ __forceinline__ void fetch_pair_seq(const TravCtx& c, const Trav& r, NodePair& n, StepAux& aux)
{
    const uint32_t goff = r.code << 6;                                        // byte offset of the 64-byte record (r.code < 2^26)
    const uint32_t laddr = lds_addr(c.top_cache) + goff + (r.code << 4);      // code * 80 bytes (kTopStride dwords)
    static_assert(kTopStride == 20u, "LDS record stride as shifts");
    aux.top = lds_addr(c.stack) + (r.sp - (r.sp != 0u ? 1u : 0u)) * (kTraceBlock * 4u);   // address in, entry out (same register)
    aux.next = lds_addr(c.objtab) + 28u + (c.tab ? (r.cur_obj + 1u) * 32u : 0u);
    exec_mask_t save;
    u2v cd;
    asm volatile(
        "ds_read_b32 %[top], %[top]\n\t"
        "ds_read_b32 %[nxt], %[nxt]\n\t"
        "v_cmp_gt_u32_e32 vcc, %[ntop], %[code]\n\t"                          // lanes whose record is in the LDS copy of the tree top
        "s_and_saveexec_b64 %[save], vcc\n\t"
        "s_cbranch_execz .Lpair_no_lds_%=\n\t"                                // (a memory instruction with no lane still makes its round trip)
        "ds_read_b128 %[q0], %[laddr]\n\t"
        "ds_read_b128 %[q1], %[laddr] offset:16\n\t"
        "ds_read_b128 %[q2], %[laddr] offset:32\n\t"
        "ds_read_b64 %[cd], %[laddr] offset:56\n"
        ".Lpair_no_lds_%=:\n\t"
        "s_andn2_b64 exec, %[save], vcc\n\t"                                  // the other lanes: HBM (L2)
        "s_cbranch_execz .Lpair_no_hbm_%=\n\t"
        "global_load_dwordx4 %[q0], %[goff], %[base]\n\t"
        "global_load_dwordx4 %[q1], %[goff], %[base] offset:16\n\t"
        "global_load_dwordx4 %[q2], %[goff], %[base] offset:32\n\t"
        "global_load_dwordx2 %[cd], %[goff], %[base] offset:56\n"
        ".Lpair_no_hbm_%=:\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)"
        : [q0] "=&v"(n.q0), [q1] "=&v"(n.q1), [q2] "=&v"(n.q2), [cd] "=&v"(cd), [top] "+v"(aux.top), [nxt] "+v"(aux.next), [save] "=&s"(save)
        : [ntop] "s"(c.n_top), [code] "v"(r.code), [laddr] "v"(laddr), [goff] "v"(goff), [base] "s"(c.sc->node_pairs)
        : "vcc", "memory");
    n.lcode = cd.x; n.rcode = cd.y;
}

__device__ __forceinline__ void fetch_leaf_seq(const TravCtx& c, const Trav& r, LeafTri& t, StepAux& aux)
{
    const uint32_t index = r.code & ~kLeafBit;
    const uint32_t goff = (index << 5) + (index << 4);                        // index * 48 bytes (index < 2^26)
    const uint32_t laddr = lds_addr(c.tri_cache) + goff;
    aux.top = lds_addr(c.stack) + (r.sp - (r.sp != 0u ? 1u : 0u)) * (kTraceBlock * 4u);
    aux.next = lds_addr(c.objtab) + 28u + (c.tab ? (r.cur_obj + 1u) * 32u : 0u);
    exec_mask_t save;
    f4v a, b; u3v tail;
    asm volatile(
        "ds_read_b32 %[top], %[top]\n\t"
        "ds_read_b32 %[nxt], %[nxt]\n\t"
        "v_cmp_gt_u32_e32 vcc, %[nlds], %[index]\n\t"                         // a small mesh's triangle (the ground quad): LDS copy
        "s_and_saveexec_b64 %[save], vcc\n\t"
        "s_cbranch_execz .Lleaf_no_lds_%=\n\t"
        "ds_read_b128 %[a], %[laddr]\n\t"
        "ds_read_b128 %[b], %[laddr] offset:16\n\t"
        "ds_read_b96 %[tail], %[laddr] offset:32\n"
        ".Lleaf_no_lds_%=:\n\t"
        "s_andn2_b64 exec, %[save], vcc\n\t"
        "s_cbranch_execz .Lleaf_no_hbm_%=\n\t"
        "global_load_dwordx4 %[a], %[goff], %[base]\n\t"
        "global_load_dwordx4 %[b], %[goff], %[base] offset:16\n\t"
        "global_load_dwordx3 %[tail], %[goff], %[base] offset:36\n"
        ".Lleaf_no_hbm_%=:\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)"
        : [a] "=&v"(a), [b] "=&v"(b), [tail] "=&v"(tail), [top] "+v"(aux.top), [nxt] "+v"(aux.next), [save] "=&s"(save)
        : [nlds] "s"(c.n_lds_tris), [index] "v"(index), [laddr] "v"(laddr), [goff] "v"(goff), [base] "s"(c.sc->tri_leaf)
        : "vcc", "memory");
    t.v0 = mk(a.x, a.y, a.z); t.e1 = mk(a.w, b.x, b.y); t.e2 = mk(b.z, b.w, __uint_as_float(tail.x));
    t.tri_idx = tail.y; t.last = tail.z != 0u;
}
__device__ __forceinline__ uint32_t next_code_of(const TravCtx&, const StepAux& aux) { return aux.next; }

// Traversal code a lane continues with when the object it is in ends: the root of object cur_obj + 1 if that is a mesh,
// otherwise kStartObject (analytic primitive or end of the list: the object step takes over).
__device__ __forceinline__ uint32_t next_object_code(const TravCtx& c, uint32_t cur_obj)
{
    if (!c.tab) return kStartObject;
    return c.objtab[(cur_obj + 1u) * 8u + 7u];                               // word 7: trav_setup; entry n_objects is the end marker
}

__device__ __forceinline__ uint32_t* deep_column(const TravCtx& c)             // this thread's column: c.stack = LDS base + 4 * threadIdx.x
{
    uint32_t a = lds_addr(c.stack);
    asm volatile("" : "+v"(a));                                               // computed where it is used, from the one per-thread address the steps keep anyway
    return c.deep_block + ((a - lds_addr(c.lds_base)) >> 2);                  // (hoisted out of the step loop the column costs two registers)
}
__device__ __forceinline__ void stack_push_any(const TravCtx& c, uint32_t level, uint32_t value)      // general forms (rare)
{
    if (level < kLdsStackLevels) c.stack[level * kTraceBlock] = value;
    else __builtin_nontemporal_store(value, &deep_column(c)[(size_t)(level - kLdsStackLevels) * c.deep_stride]);
}
__device__ __forceinline__ uint32_t stack_peek_any(const TravCtx& c, uint32_t count)                   // entry count-1 of a stack holding `count` entries
{
    uint32_t v = 0;
    if (count > kLdsStackLevels) v = __builtin_nontemporal_load(&deep_column(c)[(size_t)(count - 1u - kLdsStackLevels) * c.deep_stride]);
    else if (count > 0u) v = c.stack[(count - 1u) * kTraceBlock];
    return v;
}

// ---- inner step: both children, near one first (ref: BVH.cpp:93-123); for the lanes with r.code < kStartObject ---------------
template <bool COUNT>
__device__ __forceinline__ void inner_step(const TravCtx& c, Trav& r, Counters& cnt)
{
    const DevScene& sc = *c.sc;
    NodePair n;
    const bool general = __builtin_amdgcn_ballot_w64(r.sp >= r.fast_levels) != 0ull;   // wave-uniform, rare: a deep stack or an axis-parallel ray
#if defined(CGPT_NODE_SOA) || !defined(CGPT_FETCH_SEQ)
    if (r.code < c.n_top) load_pair_lds(c.top_cache, r.code, n);
#ifdef CGPT_NODE_SOA
    else load_pair_soa(sc.node_pairs, sc.n_pair_records, r.code, n);
#else
    else load_pair(sc.node_pairs, r.code, n);
#endif
#else
    StepAux aux;
    if (!general) fetch_pair_seq(c, r, n, aux);                               // record + stack entry + object table entry, one wait
    else if (r.code < c.n_top) load_pair_lds(c.top_cache, r.code, n);
    else load_pair(sc.node_pairs, r.code, n);
#endif
    if (COUNT) cnt.inner++;
    float left_dist, right_dist;
    if (!general) {
        // the entry below the stack pointer, read next to the node (LDS is faster): a pop is then a select
#if defined(CGPT_NODE_SOA) || !defined(CGPT_FETCH_SEQ)
        const uint32_t top = c.stack[(r.sp - (r.sp != 0u ? 1u : 0u)) * kTraceBlock];   // unused when sp == 0
        const uint32_t next_code = next_object_code(c, r.cur_obj);            // used when this object ends here
#else
        const uint32_t top = aux.top;                                         // unused when sp == 0
        const uint32_t next_code = next_code_of(c, aux);                      // used when this object ends here
#endif
        slab_pair(n, r.rs, r.t, false, left_dist, right_dist);
        const bool swap = left_dist > right_dist;                             // ref: BVH.cpp:101-105
        const uint32_t near_code = swap ? n.rcode : n.lcode, far_code = swap ? n.lcode : n.rcode;
        const float near_dist = swap ? right_dist : left_dist, far_dist = swap ? left_dist : right_dist;
        const bool miss = near_dist == 1e30f;                                 // ref: BVH.cpp:108-114
        const bool empty = r.sp == 0u;
#ifdef CGPT_PHASE_CYCLES
        if (r.code >= c.n_top) {                                              // diagnostic: how often would a partial record have been enough?
            const SlabProducts sp_ = slab_products(n, r.rs);
            auto xy_hit = [&](float t1x, float t2x, float t1y, float t2y) {
                const float hx = fmaxf(t1x, t2x), lx = fminf(t1x, t2x), hy = fmaxf(t1y, t2y), ly = fminf(t1y, t2y);
                const float tmax = fminf(hx, hy), tmin = fmaxf(lx, ly);
                return tmax >= tmin && tmin < r.t && tmax > 0.0f;
            };
            auto x_hit = [&](float t1x, float t2x) { const float hx = fmaxf(t1x, t2x), lx = fminf(t1x, t2x); return lx < r.t && hx > 0.0f; };
            cnt.global_inner++;
            cnt.both_miss += miss ? 1u : 0u;
            cnt.xy_both_miss += (!xy_hit(sp_.t1x.x, sp_.t2x.x, sp_.t1y.x, sp_.t2y.x) && !xy_hit(sp_.t1x.y, sp_.t2x.y, sp_.t1y.y, sp_.t2y.y)) ? 1u : 0u;
            cnt.x_both_miss += (!x_hit(sp_.t1x.x, sp_.t2x.x) && !x_hit(sp_.t1x.y, sp_.t2x.y)) ? 1u : 0u;
        }
#endif
        c.stack[r.sp * kTraceBlock] = far_code;                                      // the free slot above the top: counts only if sp moves up
        r.code = miss ? (empty ? next_code : top) : near_code;
        r.cur_obj += (miss & empty) ? 1u : 0u;
        r.depth += miss ? 0u : 1u;                                            // ref: BVH.cpp:118
        if (COUNT) cnt.depth += miss ? 0u : 1u;
        r.sp = miss ? (empty ? 0u : r.sp - 1u) : r.sp + ((far_dist != 1e30f) ? 1u : 0u);
    } else {
        slab_pair(n, r.rs, r.t, __builtin_amdgcn_ballot_w64(r.fast_levels == 0u) != 0ull, left_dist, right_dist);
        uint32_t left_code = n.lcode, right_code = n.rcode;
        if (left_dist > right_dist) {
            float td = left_dist; left_dist = right_dist; right_dist = td;
            uint32_t tc = left_code; left_code = right_code; right_code = tc;
        }
        if (left_dist == 1e30f) {
            if (r.sp == 0u) { r.cur_obj++; r.code = kStartObject; }
            else { r.code = stack_peek_any(c, r.sp); --r.sp; }
        } else {
            r.depth++;
            if (COUNT) cnt.depth++;
            r.code = left_code;
            if (right_dist != 1e30f) { stack_push_any(c, r.sp, right_code); ++r.sp; }
        }
    }
}

// ---- leaf step: one triangle of the leaf (ref: BVH.cpp:74-90); for the lanes with bit 31 of r.code set ------------------------
template <bool COUNT, bool ANY_HIT = true>                                    // ANY_HIT false: the caller's rays never carry the any-hit flag (round 0)
__device__ __forceinline__ void leaf_step(const TravCtx& c, Trav& r, Counters& cnt)
{
    LeafTri lt;
    const uint32_t leaf_index = r.code & ~kLeafBit;
    uint32_t top, next_code;                                                  // entry below the stack pointer, code after this object: read next to the triangle
#ifndef CGPT_FETCH_SEQ
    if (leaf_index < c.n_lds_tris) lt = load_leaf_tri_lds(c.tri_cache, leaf_index);   // a small mesh's triangle (the ground quad): LDS copy
    else lt = load_leaf_tri(c.sc->tri_leaf, leaf_index);
    if (__builtin_amdgcn_ballot_w64(r.sp > kLdsStackLevels) == 0ull) top = c.stack[(r.sp - (r.sp != 0u ? 1u : 0u)) * kTraceBlock];
    else top = stack_peek_any(c, r.sp);
    next_code = next_object_code(c, r.cur_obj);
#else
    if (__builtin_amdgcn_ballot_w64(r.sp > kLdsStackLevels) == 0ull) {
        StepAux aux;
        fetch_leaf_seq(c, r, lt, aux);
        top = aux.top; next_code = next_code_of(c, aux);
    } else {
        if (leaf_index < c.n_lds_tris) lt = load_leaf_tri_lds(c.tri_cache, leaf_index);
        else lt = load_leaf_tri(c.sc->tri_leaf, leaf_index);
        top = stack_peek_any(c, r.sp);
        next_code = next_object_code(c, r.cur_obj);
    }
#endif
    if (COUNT) cnt.tris++;
    float t_hit;
    const bool hit = intersect_triangle_flags(lt.v0, lt.e1, lt.e2, trav_origin(r), r.d, r.t, t_hit);
    r.t = hit ? t_hit : r.t;
    r.tri = hit ? lt.tri_idx : r.tri;
    r.obj = hit ? r.cur_obj : r.obj;                                          // ref: Main.cpp:313-314
    const bool last = lt.last;                                                // last triangle of the leaf: pop (ref: BVH.cpp:86-90)
    const bool empty = r.sp == 0u;
    r.code = last ? (empty ? next_code : top) : r.code + 1u;
    r.cur_obj += (last & empty) ? 1u : 0u;
    r.sp = (last & !empty) ? r.sp - 1u : r.sp;
    if (!COUNT && ANY_HIT && trav_any_hit(r) && hit) { r.code = kStartObject; r.cur_obj = c.sc->n_objects; r.sp = 0u; }   // occluded: straight to the end of the object list
}

// ---- lean traversal: this lane's ray through its meshes to the next object boundary, in a tight divergent loop ----------------
// The voted steps above buy lane occupancy with latency: a lane waits until its state is voted, and every step runs the branch-free
// form of both outcomes.  At the END of a launch that is the wrong trade -- a handful of long rays are left (a one-sample 1080p frame
// of the glass scene has paths of 1 600 dependent fetches where the mean is 25), nothing can refill the idle lanes, and the launch ends
// when the longest chain does.  This loop is the reference's own control flow (ref: BVH.cpp:68-125) per lane, early returns and all:
// one record fetch and ~50 instructions per step instead of a vote, and the results are the voted steps' bit for bit (same slab test,
// same triangle arithmetic -- intersect_triangle's early returns leave the same t as the flag form, rt_device.hpp).
// Returns with r.code == kStartObject (the lane's ray is at an analytic object or at the end of the object list).
template <bool COUNT, bool ANY_HIT = true>
__device__ __forceinline__ void lean_traverse(const TravCtx& c, Trav& r, Counters& cnt)
{
    const DevScene& sc = *c.sc;
    const V3 o = trav_origin(r);
    while (r.code != kStartObject) {
        if ((int32_t)r.code < 0) {                                            // a leaf: its triangles in order (ref: BVH.cpp:74-90)
            uint32_t i = r.code & ~kLeafBit;
            bool occluded = false;
            for (;;) {
                LeafTri lt;
                if (i < c.n_lds_tris) lt = load_leaf_tri_lds(c.tri_cache, i); else lt = load_leaf_tri(sc.tri_leaf, i);
                if (COUNT) cnt.tris++;
                if (intersect_triangle(lt.v0, lt.e1, lt.e2, o, r.d, r.t)) {
                    r.tri = lt.tri_idx; r.obj = r.cur_obj;                    // ref: Main.cpp:313-314
                    if (!COUNT && ANY_HIT && trav_any_hit(r)) { occluded = true; break; }
                }
                if (lt.last) break;
                ++i;
            }
            if (occluded) { r.code = kStartObject; r.cur_obj = sc.n_objects; r.sp = 0u; break; }
            if (r.sp == 0u) { r.code = next_object_code(c, r.cur_obj); r.cur_obj++; }
            else { r.code = stack_peek_any(c, r.sp); --r.sp; }
            continue;
        }
        NodePair n;
        if (r.code < c.n_top) load_pair_lds(c.top_cache, r.code, n);
#ifdef CGPT_NODE_SOA
        else load_pair_soa(sc.node_pairs, sc.n_pair_records, r.code, n);
#else
        else load_pair(sc.node_pairs, r.code, n);
#endif
        if (COUNT) cnt.inner++;
        float left_dist, right_dist;
        slab_pair(n, r.rs, r.t, r.fast_levels == 0u, left_dist, right_dist);
        uint32_t left_code = n.lcode, right_code = n.rcode;
        if (left_dist > right_dist) {                                         // ref: BVH.cpp:101-105
            const float td = left_dist; left_dist = right_dist; right_dist = td;
            const uint32_t tc = left_code; left_code = right_code; right_code = tc;
        }
        if (left_dist == 1e30f) {                                             // ref: BVH.cpp:108-114
            if (r.sp == 0u) { r.code = next_object_code(c, r.cur_obj); r.cur_obj++; }
            else { r.code = stack_peek_any(c, r.sp); --r.sp; }
        } else {                                                              // ref: BVH.cpp:115-123
            r.depth++;
            if (COUNT) cnt.depth++;
            r.code = left_code;
            if (right_dist != 1e30f) { stack_push_any(c, r.sp, right_code); ++r.sp; }
        }
    }
}

// ---- object step: the analytic primitives from cur_obj on, then begin the next mesh or finish the ray
//      (IntersectScene's loop, ref: Main.cpp:303-315); for the lanes with r.code == kStartObject.
// Returns true when the scene's object list is exhausted for this lane: the ray is done and the caller runs its epilogue
// (r.code is left at kStartObject).  Otherwise the lane continues inside a mesh.
template <bool COUNT, bool ANY_HIT = true>
__device__ __forceinline__ bool object_step(const TravCtx& c, Trav& r, Counters& cnt)
{
    const DevScene& sc = *c.sc;
    const V3 o = trav_origin(r);
    for (;;) {
        float4 q0, q1;
        if (c.tab) {
            const f4v l0 = *reinterpret_cast<const lds_f4v*>(c.objtab + r.cur_obj * 8u), l1 = *reinterpret_cast<const lds_f4v*>(c.objtab + r.cur_obj * 8u + 4u);
            q0.x = l0.x; q0.y = l0.y; q0.z = l0.z; q0.w = l0.w; q1.x = l1.x; q1.y = l1.y; q1.z = l1.z; q1.w = l1.w;
        } else if (r.cur_obj < sc.n_objects) {
            q0 = sc.obj_trace[2u * r.cur_obj]; q1 = sc.obj_trace[2u * r.cur_obj + 1u];
        } else {
            q0.x = __uint_as_float(kKindEnd); q0.y = q0.z = q0.w = 0.0f; q1 = q0;
        }
        const uint32_t kind = __float_as_uint(q0.x);
        if (kind == kKindEnd) return true;                                    // no object left: the ray is done
        if (kind == 0u) { r.code = __float_as_uint(q0.y); r.sp = 0u; return false; }
        bool hit;
        if (kind == 1u) hit = intersect_sphere(mk(q0.y, q0.z, q0.w), q1.x, o, r.d, r.t);
        else hit = intersect_plane(mk(q0.y, q0.z, q0.w), mk(q1.x, q1.y, q1.z), o, r.d, r.t);
        if (hit) r.obj = r.cur_obj;
        if (!COUNT && ANY_HIT && trav_any_hit(r) && hit) return true;
        r.cur_obj++;
    }
}

}  // namespace dev
}  // namespace cgpt
