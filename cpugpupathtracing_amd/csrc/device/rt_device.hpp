// rt_device.hpp -- device functions of the hot path (gfx950): math, per-lane PCG, intersectors, LDS-stack traversal,
// sampling/optics.  Each function cites the reference code it computes the same result as; operand order is kept and
// the translation unit is compiled with -ffp-contract=off (no FMA), IEEE divide/sqrt, so results are bit-identical to
// the CPU reference except for expf (Beer's law), which is value-only (SURVEY section 7 "Hard parts").
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.h"

namespace cgpt {
namespace dev {

// ---- float3 in registers (ref: Include/MathLib.h:57-102) ---------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 mk(float s) { return mk(s, s, s); }
__device__ __forceinline__ V3 mk(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float length(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a) { float rcp = 1.0f / length(a); return a * rcp; }   // ref: MathLib.h:93
__device__ __forceinline__ V3 lerp(V3 a, V3 b, float s) { return mk(a.x + (b.x - a.x) * s, a.y + (b.y - a.y) * s, a.z + (b.z - a.z) * s); }
// std::min(a,b) = (b<a)?b:a ; std::max(a,b) = (a<b)?b:a ; std::clamp(v,lo,hi) = (v<lo)?lo:(hi<v)?hi:v
__device__ __forceinline__ float min_std(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float max_std(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float clamp_std(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

static constexpr float kPi = 3.14159265f;          // ref: MathLib.h:5
static constexpr float kInvPi = 1.0f / kPi;        // ref: MathLib.h:7
static constexpr float kNudge = 0.001f;            // ref: Main.cpp:49 RAY_REFLECT_NUDGE_MULTIPLIER

// ref: MathLib.h:144-152 (truncation, no gamma; negative clamped to 0: SURVEY A-15)
__device__ __forceinline__ uint32_t vec4_to_uint(float x, float y, float z)
{
    float fr = 255.0f * min_std(1.0f, x), fg = 255.0f * min_std(1.0f, y), fb = 255.0f * min_std(1.0f, z);
    uint32_t r = (uint32_t)(int32_t)(fr < 0.0f ? 0.0f : fr) & 0xFFu;
    uint32_t g = (uint32_t)(int32_t)(fg < 0.0f ? 0.0f : fg) & 0xFFu;
    uint32_t b = (uint32_t)(int32_t)(fb < 0.0f ? 0.0f : fb) & 0xFFu;
    return (255u << 24) + (b << 16) + (g << 8) + r;
}

// ---- per-lane RNG (replaces the racy global xorshift, ref: Include/Random.h:4-51; SURVEY A-2, Appendix C) ---------
__device__ __forceinline__ uint32_t wang_hash(uint32_t seed)   // ref: Random.h:6-13
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u; seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
// stream key = (global pixel index, sample index, seed): identical for any row tiling / GPU count
__device__ __forceinline__ uint32_t pcg_seed(uint32_t pixel_index, uint32_t sample_index, uint32_t seed)
{
    uint32_t h = wang_hash(seed);
    h = wang_hash(h ^ sample_index);
    h = wang_hash(h ^ pixel_index);
    return h;
}
__device__ __forceinline__ uint32_t pcg_next(uint32_t& s)       // PCG-RXS-M-XS-32
{
    uint32_t old = s;
    s = old * 747796405u + 2891336453u;
    uint32_t w = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
    return (w >> 22u) ^ w;
}
__device__ __forceinline__ float random_float(uint32_t& s) { return (float)pcg_next(s) * 2.3283064365387e-10f; }   // ref: Random.h:31-34
__device__ __forceinline__ uint32_t random_range(uint32_t& s, uint32_t mn, uint32_t mx)                             // ref: Random.h:41-46
{
    if (mx - mn == 0) return mn;
    return mn + (pcg_next(s) % ((mx + 1) - mn));
}

// ---- ray (ref: Include/Primitives.h:59-83) ------------------------------------------------------------------------
struct Ray {
    V3 o, d;
    float t;
    uint32_t obj, tri, bvh_depth;
};
__device__ __forceinline__ Ray make_ray(V3 o, V3 d, float t) { Ray r; r.o = o; r.d = d; r.t = t; r.obj = kNoHit; r.tri = 0; r.bvh_depth = 0; return r; }

struct Counters { uint32_t rays, inner, tris, depth, hits; uint32_t both_miss = 0, xy_both_miss = 0, x_both_miss = 0, global_inner = 0; };   // the last four: diagnostic builds only

// ---- intersectors (ref: Source/Primitives.cpp:6-130) --------------------------------------------------------------
// Moeller-Trumbore with the reference's absolute determinant epsilon (SURVEY A-9); e1/e2 precomputed at upload.
__device__ __forceinline__ bool intersect_triangle(V3 v0, V3 e1, V3 e2, V3 o, V3 d, float& ray_t)
{
    V3 H = cross(d, e2);
    float a = dot(e1, H);
    if (fabsf(a) < 0.001f) return false;
    float f = 1.0f / a;
    V3 S = o - v0;
    float u = f * dot(S, H);
    if (u < 0.0f || u > 1.0f) return false;
    V3 Q = cross(S, e1);
    float v = f * dot(d, Q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = f * dot(e2, Q);
    if (t > 0.0f && t < ray_t) { ray_t = t; return true; }
    return false;
}

// The same test without early returns: every rejection of the reference becomes a flag, all of them are ANDed at the end
// (the tests have no side effects, so the result is the same for every input, NaNs included).  For the wavefront trace
// kernel, where the lanes of a wave hold unrelated rays and an early return saves nothing unless all of them take it.
__device__ __forceinline__ bool intersect_triangle_flags(V3 v0, V3 e1, V3 e2, V3 o, V3 d, float ray_t, float& t_out)
{
    const V3 H = cross(d, e2);
    const float a = dot(e1, H);
    bool ok = !(fabsf(a) < 0.001f);
    const float f = 1.0f / a;
    const V3 S = o - v0;
    const float u = f * dot(S, H);
    ok = ok & !((u < 0.0f) | (u > 1.0f));
    const V3 Q = cross(S, e1);
    const float v = f * dot(d, Q);
    ok = ok & !((v < 0.0f) | (u + v > 1.0f));
    const float t = f * dot(e2, Q);
    ok = ok & ((t > 0.0f) & (t < ray_t));
    t_out = t;
    return ok;
}

__device__ __forceinline__ bool intersect_plane(V3 normal, V3 point, V3 o, V3 d, float& ray_t)   // ref: Primitives.cpp:49-69
{
    float denom = dot(d, normal);
    if ((double)fabsf(denom) > 1e-6) {   // the reference compares against a double literal
        V3 p0 = point - o;
        float t = dot(p0, normal) / denom;
        if (t > 0.0f && t < ray_t) { ray_t = t; return true; }
    }
    return false;
}

__device__ __forceinline__ bool intersect_sphere(V3 center, float radius_sq, V3 o, V3 d, float& ray_t)   // ref: Primitives.cpp:71-114
{
    V3 L = center - o;
    float tca = dot(L, d);
    if (tca < 0.0f) return false;                      // origin inside & centre behind -> miss (SURVEY A-12)
    float d2 = dot(L, L) - tca * tca;
    if (d2 > radius_sq) return false;
    float thc = sqrtf(radius_sq - d2);
    float t0 = tca - thc, t1 = tca + thc;
    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 < 0.0f) { t0 = t1; if (t0 < 0.0f) return false; }
    if (t0 < ray_t) { ray_t = t0; return true; }
    return false;
}

// ---- slab test (ref: Primitives.cpp:116-130): see slab_pair() below ------------------------------------------------------
__device__ __forceinline__ bool has_infinite_component(V3 inv)
{
    return __builtin_isinf(inv.x) || __builtin_isinf(inv.y) || __builtin_isinf(inv.z);
}
// ---- both children of an inner node at once (node_pairs layout: device_scene.h) ----------------------------------------
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

struct NodePair { f4v q0, q1, q2; uint32_t lcode, rcode; };                 // 56 useful bytes of the 64-byte record
#ifdef CGPT_NODE_SOA
// Experiment build (north_star suggests SoA node planes "for coalesced HBM reads"; measured in profiles/r02/node_layout_ab.md):
// component plane p of record r at dword p * n_records + r, so a lane's record is 14 separate 4-byte fetches from 14 lines.
__device__ __forceinline__ void load_pair_soa(const float4* node_pairs, uint32_t n_records, uint32_t code, NodePair& n)
{
    const float* p = reinterpret_cast<const float*>(node_pairs) + code;
    const size_t s = n_records;
    n.q0 = f4v{ p[0], p[s], p[2 * s], p[3 * s] };
    n.q1 = f4v{ p[4 * s], p[5 * s], p[6 * s], p[7 * s] };
    n.q2 = f4v{ p[8 * s], p[9 * s], p[10 * s], p[11 * s] };
    n.lcode = __float_as_uint(p[14 * s]); n.rcode = __float_as_uint(p[15 * s]);
}
#endif
__device__ __forceinline__ void load_pair(const float4* node_pairs, uint32_t code, NodePair& n)
{
    // byte offset in 32 bits (2^26 records = 4 GB): global_load with a scalar base and a 32-bit VGPR offset, no 64-bit address math
    const char* rec = reinterpret_cast<const char*>(node_pairs) + (size_t)(code << 6);
    n.q0 = *reinterpret_cast<const f4v*>(rec);
    n.q1 = *reinterpret_cast<const f4v*>(rec + 16);
    n.q2 = *reinterpret_cast<const f4v*>(rec + 32);
    const u2v codes = *reinterpret_cast<const u2v*>(rec + 56);               // 8 bytes, 8-byte aligned: stays a dwordx2 load
    n.lcode = codes.x; n.rcode = codes.y;
}
// Ray constants of the slab test as three even VGPR pairs: {o.x, o.y}, {inv.x, inv.y}, {o.z, inv.z}.
struct RaySlab { f2v oxy, ixy, ozi; };
__device__ __forceinline__ RaySlab make_ray_slab(V3 o, V3 inv)
{
    RaySlab r; r.oxy.x = o.x; r.oxy.y = o.y; r.ixy.x = inv.x; r.ixy.y = inv.y; r.ozi.x = o.z; r.ozi.y = inv.z; return r;
}
// (bounds - o) * inv for both children at once (ref: Primitives.cpp:118-120, BVH.cpp:97-98): each {left, right} pair of
// the record is one even VGPR pair, so the six subtractions and six multiplications of the two slab tests are six
// v_pk_add_f32 and six v_pk_mul_f32 (IEEE, the same roundings as the scalar forms).  op_sel picks the ray component out of
// its pair for both halves, so no splatted copies of the ray are kept in registers.
struct SlabProducts { f2v t1x, t1y, t1z, t2x, t2y, t2z; };                  // .x = left child, .y = right child
__device__ __forceinline__ f2v pk_sub_mul_lo_lo(f2v p, f2v o_pair, f2v i_pair)   // (p - o_pair.x) * i_pair.x
{
    f2v d, r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(o_pair));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(d), "v"(i_pair));
    return r;
}
__device__ __forceinline__ f2v pk_sub_mul_hi_hi(f2v p, f2v o_pair, f2v i_pair)   // (p - o_pair.y) * i_pair.y
{
    f2v d, r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(o_pair));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(d), "v"(i_pair));
    return r;
}
__device__ __forceinline__ f2v pk_sub_mul_lo_hi(f2v p, f2v oi_pair)              // (p - oi_pair.x) * oi_pair.y
{
    f2v d, r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(oi_pair));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(d), "v"(oi_pair));
    return r;
}
__device__ __forceinline__ f2v sc_sub_mul(f2v p, float o, float i)          // the scalar form: four full-rate instructions
{
    f2v r; r.x = (p.x - o) * i; r.y = (p.y - o) * i; return r;
}
__device__ __forceinline__ SlabProducts slab_products(const NodePair& n, const RaySlab& r)
{
    SlabProducts s;
#ifdef CGPT_SLAB_SCALAR
    s.t1x = sc_sub_mul(n.q0.xy, r.oxy.x, r.ixy.x); s.t1y = sc_sub_mul(n.q0.zw, r.oxy.y, r.ixy.y); s.t1z = sc_sub_mul(n.q1.xy, r.ozi.x, r.ozi.y);
    s.t2x = sc_sub_mul(n.q1.zw, r.oxy.x, r.ixy.x); s.t2y = sc_sub_mul(n.q2.xy, r.oxy.y, r.ixy.y); s.t2z = sc_sub_mul(n.q2.zw, r.ozi.x, r.ozi.y);
    return s;
#endif
    s.t1x = pk_sub_mul_lo_lo(n.q0.xy, r.oxy, r.ixy); s.t1y = pk_sub_mul_hi_hi(n.q0.zw, r.oxy, r.ixy); s.t1z = pk_sub_mul_lo_hi(n.q1.xy, r.ozi);
    s.t2x = pk_sub_mul_lo_lo(n.q1.zw, r.oxy, r.ixy); s.t2y = pk_sub_mul_hi_hi(n.q2.xy, r.oxy, r.ixy); s.t2z = pk_sub_mul_lo_hi(n.q2.zw, r.ozi);
    return s;
}
// Slab distance from the products, for rays without an axis-parallel direction: no 0 * inf = NaN can occur, and for
// non-NaN operands v_min / v_max / v_min3 / v_max3 return the same values as the reference's compare-and-select forms
// except for the sign of a zero result, which none of the comparisons below (or the callers' ==, >) can observe.
__device__ __forceinline__ float slab_dist_finite(float t1x, float t1y, float t1z, float t2x, float t2y, float t2z, float ray_t)
{
    float hx, hy, hz, lx, ly, lz, tmax, tmin;
    asm("v_max_f32 %0, %1, %2" : "=v"(hx) : "v"(t1x), "v"(t2x));
    asm("v_max_f32 %0, %1, %2" : "=v"(hy) : "v"(t1y), "v"(t2y));
    asm("v_max_f32 %0, %1, %2" : "=v"(hz) : "v"(t1z), "v"(t2z));
    asm("v_min_f32 %0, %1, %2" : "=v"(lx) : "v"(t1x), "v"(t2x));
    asm("v_min_f32 %0, %1, %2" : "=v"(ly) : "v"(t1y), "v"(t2y));
    asm("v_min_f32 %0, %1, %2" : "=v"(lz) : "v"(t1z), "v"(t2z));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(hx), "v"(hy), "v"(hz));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"(lx), "v"(ly), "v"(lz));
    return (tmax >= tmin && tmin < ray_t && tmax > 0.0f) ? tmin : 1e30f;
}
// the SSE version's NaN behaviour (ref: Primitives.cpp:121-128; SURVEY A-18), for waves holding an axis-parallel ray
__device__ __forceinline__ float slab_dist_exact(float t1x, float t1y, float t1z, float t2x, float t2y, float t2z, float ray_t)
{
    float vmaxx = t1x > t2x ? t1x : t2x, vminx = t1x < t2x ? t1x : t2x;
    float vmaxy = t1y > t2y ? t1y : t2y, vminy = t1y < t2y ? t1y : t2y;
    float vmaxz = t1z > t2z ? t1z : t2z, vminz = t1z < t2z ? t1z : t2z;
    float tmax = min_std(vmaxx, min_std(vmaxy, vmaxz));
    float tmin = max_std(vminx, max_std(vminy, vminz));
    if (tmax >= tmin && tmin < ray_t && tmax > 0.0f) return tmin;
    return 1e30f;
}
__device__ __forceinline__ void slab_pair(const NodePair& n, const RaySlab& r, float ray_t, bool exact, float& left_dist, float& right_dist)
{
    const SlabProducts s = slab_products(n, r);
    if (!exact) {
        left_dist = slab_dist_finite(s.t1x.x, s.t1y.x, s.t1z.x, s.t2x.x, s.t2y.x, s.t2z.x, ray_t);
        right_dist = slab_dist_finite(s.t1x.y, s.t1y.y, s.t1z.y, s.t2x.y, s.t2y.y, s.t2z.y, ray_t);
    } else {
        left_dist = slab_dist_exact(s.t1x.x, s.t1y.x, s.t1z.x, s.t2x.x, s.t2y.x, s.t2z.x, ray_t);
        right_dist = slab_dist_exact(s.t1x.y, s.t1y.y, s.t1z.y, s.t2x.y, s.t2y.y, s.t2z.y, ray_t);
    }
}

// leaf triangle record (tri_leaf layout: device_scene.h): 44 useful bytes as 16 + 16 + 12
struct LeafTri { V3 v0, e1, e2; uint32_t tri_idx; bool last; };
struct LeafTail { float e2z; uint32_t tri_idx, last; };
__device__ __forceinline__ LeafTri load_leaf_tri(const float4* tri_leaf, uint32_t index)
{
    const char* rec = reinterpret_cast<const char*>(tri_leaf) + (size_t)((index << 5) + (index << 4));   // index * 48 as two shifts (v_mul_lo_u32 is quarter rate); 32-bit byte offset (index < 2^26)
    const f4v a = *reinterpret_cast<const f4v*>(rec), b = *reinterpret_cast<const f4v*>(rec + 16);
    const LeafTail c = *reinterpret_cast<const LeafTail*>(rec + 36);
    LeafTri t;
    t.v0 = mk(a.x, a.y, a.z); t.e1 = mk(a.w, b.x, b.y); t.e2 = mk(b.z, b.w, c.e2z);
    t.tri_idx = c.tri_idx; t.last = c.last != 0u;
    return t;
}

// ---- BVH traversal with a per-wavefront LDS stack (ref: Source/BVH.cpp:61-127) -----------------------------------
// Ordered (near child first) traversal; the far child is pushed only when hit, so the stack never holds more than one
// entry per tree level.  Stack layout: stack[level * blockDim.x + threadIdx.x] -> the 64 lanes of a wave hit 64
// consecutive banks (conflict-free ds_read/ds_write_b32).
template <bool COUNT>
__device__ __forceinline__ bool traverse_mesh(const DevScene& sc, uint32_t root_code, V3 o, V3 d, V3 inv, float& ray_t,
                                              uint32_t& tri_idx, uint32_t& bvh_depth, uint32_t* __restrict__ stack,
                                              uint32_t stack_stride, Counters& cnt)
{
    bool result = false;
    uint32_t code = root_code;
    uint32_t sp = 0;
    const bool exact_slab = has_infinite_component(inv);                     // axis-parallel ray: NaN-exact slab test (SURVEY A-18)
    const RaySlab rs = make_ray_slab(o, inv);
    for (;;) {
        if (code & kLeafBit) {
            uint32_t i = code & ~kLeafBit;
            for (;;) {
                const LeafTri lt = load_leaf_tri(sc.tri_leaf, i);
                if (COUNT) cnt.tris++;
                if (intersect_triangle(lt.v0, lt.e1, lt.e2, o, d, ray_t)) {
                    tri_idx = lt.tri_idx;
                    result = true;
                }
                if (lt.last) break;
                ++i;
            }
            if (sp == 0) break;
            code = stack[(--sp) * stack_stride];
            continue;
        }
        NodePair n;
#ifdef CGPT_NODE_SOA
        load_pair_soa(sc.node_pairs, sc.n_pair_records, code, n);
#else
        load_pair(sc.node_pairs, code, n);
#endif
        if (COUNT) cnt.inner++;
        float left_dist, right_dist;
        slab_pair(n, rs, ray_t, __builtin_amdgcn_ballot_w64(exact_slab) != 0ull, left_dist, right_dist);   // wave-uniform: NaN-exact form only if somebody needs it
        uint32_t left_code = n.lcode, right_code = n.rcode;
        if (left_dist > right_dist) {                                     // ref: BVH.cpp:101-105
            float td = left_dist; left_dist = right_dist; right_dist = td;
            uint32_t tc = left_code; left_code = right_code; right_code = tc;
        }
        if (left_dist == 1e30f) {                                         // ref: BVH.cpp:108-114
            if (sp == 0) break;
            code = stack[(--sp) * stack_stride];
        } else {                                                          // ref: BVH.cpp:115-123
            bvh_depth++;
            if (COUNT) cnt.depth++;
            code = left_code;
            if (right_dist != 1e30f) stack[(sp++) * stack_stride] = right_code;
        }
    }
    return result;
}

// IntersectScene (ref: Source/Main.cpp:299-316): closest hit over all objects in order; strict t < ray.t everywhere.
template <bool COUNT>
__device__ __forceinline__ void intersect_scene(const DevScene& sc, Ray& ray, uint32_t* __restrict__ stack, uint32_t stack_stride, Counters& cnt)
{
    cnt.rays++;
    const V3 inv = mk(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);   // Ray ctor, ref: Primitives.h:64
    for (uint32_t obj_idx = 0; obj_idx < sc.n_objects; ++obj_idx) {
        const DevObject& obj = sc.objects[obj_idx];
        bool hit;
        if (obj.kind == 0u) hit = traverse_mesh<COUNT>(sc, obj.root_code, ray.o, ray.d, inv, ray.t, ray.tri, ray.bvh_depth, stack, stack_stride, cnt);
        else if (obj.kind == 1u) hit = intersect_sphere(mk(obj.sphere_center), obj.sphere_radius_sq, ray.o, ray.d, ray.t);
        else hit = intersect_plane(mk(obj.plane_normal), mk(obj.plane_point), ray.o, ray.d, ray.t);
        if (hit) ray.obj = obj_idx;
    }
}

// ---- sampling / optics (ref: Source/Util.cpp:7-54) -------------------------------------------------------------------
__device__ __forceinline__ V3 ball_sample(uint32_t& rng)              // rejection loop of Util.cpp:10-13; draw order x, y, z
{
    V3 dir;
    do {
        float x = random_float(rng) * 2.0f - 1.0f;
        float y = random_float(rng) * 2.0f - 1.0f;
        float z = random_float(rng) * 2.0f - 1.0f;
        dir = mk(x, y, z);
    } while (dot(dir, dir) > 1.0f);
    return dir;
}
__device__ __forceinline__ V3 uniform_hemisphere_sample(uint32_t& rng, V3 normal)    // ref: Util.cpp:7-19
{
    V3 dir = ball_sample(rng);
    if (dot(dir, normal) < 0.0f) dir = dir * mk(-1.0f);
    return normalize(dir);
}
__device__ __forceinline__ V3 cosine_weighted_diffuse_reflection(uint32_t& rng, V3 normal)   // ref: Util.cpp:21-30
{
    V3 dir = ball_sample(rng);
    return normalize(normal + normalize(dir));
}
__device__ __forceinline__ float survival_probability_rr(V3 albedo)                   // ref: Util.cpp:32-35
{
    return clamp_std(max_std(max_std(albedo.x, albedo.y), albedo.z), 0.1f, 1.0f);
}
__device__ __forceinline__ V3 reflect(V3 dir, V3 normal) { return dir - (2.0f * normal) * dot(dir, normal); }   // ref: Util.cpp:37-40
__device__ __forceinline__ float fresnel(float in, float out, float ior_outside, float ior_inside)   // ref: Util.cpp:42-49
{
    float s_pol = (ior_outside * in - ior_inside * out) / (ior_outside * in + ior_inside * out);
    float p_pol = (ior_outside * out - ior_inside * in) / (ior_outside * out + ior_inside * in);
    return 0.5f * ((s_pol * s_pol) + (p_pol * p_pol));
}
__device__ __forceinline__ V3 refract(V3 dir, V3 normal, float eta, float cosi, float k)   // ref: Util.cpp:51-54
{
    return normalize(dir * eta + ((eta * cosi - sqrtf(k)) * normal));
}

// ---- materials ---------------------------------------------------------------------------------------------------------
struct Mat {
    V3 albedo; float specular, refractivity; V3 absorption; float ior; V3 emissive; float intensity; bool is_light;
};
__device__ __forceinline__ Mat load_material(const DevScene& sc, uint32_t index)
{
    const float4* p = sc.materials + 4u * (size_t)index;
    float4 a = p[0], b = p[1], c = p[2], d = p[3];
    Mat m;
    m.albedo = mk(a.x, a.y, a.z); m.specular = a.w;
    m.refractivity = b.x; m.absorption = mk(b.y, b.z, b.w);
    m.ior = c.x; m.emissive = mk(c.y, c.z, c.w);
    m.intensity = d.x; m.is_light = __float_as_uint(d.y) != 0u;
    return m;
}

// ---- camera (ref: Source/Main.cpp:133-140) ---------------------------------------------------------------------------
__device__ __forceinline__ Ray camera_ray(const DevCamera& cam, float u, float v)
{
    V3 tl = mk(cam.top_left), tr = mk(cam.top_right), bl = mk(cam.bottom_left), pos = mk(cam.pos);
    V3 pixel_pos = tl + u * (tr - tl) + v * (bl - tl);
    return make_ray(pos, normalize(pixel_pos - pos), 1e34f);
}

// ---- wave-level reduction of a per-lane counter, then one atomic per wave ---------------------------------------------
__device__ __forceinline__ void wave_add_u64(unsigned long long* dst, uint32_t v)
{
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63u) == 0u && s) atomicAdd(dst, s);
}
__device__ __forceinline__ void wave_add_f64(double* dst, double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63u) == 0u && v != 0.0) atomicAdd(dst, v);
}

// one atomic per 256-thread BLOCK (all threads of the block must call it): a kernel of 8 000 blocks that ends in one atomic per
// wave spends ~0.35 ms queueing 32 000 atomics on one L2 word (~88 per microsecond), more than its own work at 1080p
__device__ __forceinline__ void block_add_f64(double* dst, double v)
{
    __shared__ double partial[4];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63u) == 0u) partial[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0u) {
        const double s = (partial[0] + partial[1]) + (partial[2] + partial[3]);
        if (s != 0.0) atomicAdd(dst, s);
    }
}

}  // namespace dev
}  // namespace cgpt
