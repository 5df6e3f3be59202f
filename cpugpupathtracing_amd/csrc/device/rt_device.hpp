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

struct Counters { uint32_t rays, inner, tris, depth, hits; };

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

// Slab test with the SSE version's NaN behaviour (ref: Primitives.cpp:116-130; SURVEY A-18):
// _mm_max_ps(a,b) = a>b?a:b, _mm_min_ps(a,b) = a<b?a:b, then std::min / std::max across lanes.
__device__ __forceinline__ float intersect_aabb(float4 bmin, float4 bmax, V3 o, V3 inv, float ray_t)
{
    float t1x = (bmin.x - o.x) * inv.x, t2x = (bmax.x - o.x) * inv.x;
    float t1y = (bmin.y - o.y) * inv.y, t2y = (bmax.y - o.y) * inv.y;
    float t1z = (bmin.z - o.z) * inv.z, t2z = (bmax.z - o.z) * inv.z;
    float vmaxx = t1x > t2x ? t1x : t2x, vminx = t1x < t2x ? t1x : t2x;
    float vmaxy = t1y > t2y ? t1y : t2y, vminy = t1y < t2y ? t1y : t2y;
    float vmaxz = t1z > t2z ? t1z : t2z, vminz = t1z < t2z ? t1z : t2z;
    float tmax = min_std(vmaxx, min_std(vmaxy, vmaxz));
    float tmin = max_std(vminx, max_std(vminy, vminz));
    if (tmax >= tmin && tmin < ray_t && tmax > 0.0f) return tmin;
    return 1e30f;
}

// Same test for rays whose inverse direction has no infinite component (no axis-parallel direction): then no 0 * inf = NaN
// can occur, and for non-NaN operands v_min_f32 / v_max_f32 / v_min3 / v_max3 return the same values as the compare-and-
// select forms above except for the sign of a zero result, which none of the comparisons below (or the callers' ==, >)
// can observe.  ~20 VALU instead of ~45 per box.
__device__ __forceinline__ float intersect_aabb_finite(float4 bmin, float4 bmax, V3 o, V3 inv, float ray_t)
{
    float t1x = (bmin.x - o.x) * inv.x, t2x = (bmax.x - o.x) * inv.x;
    float t1y = (bmin.y - o.y) * inv.y, t2y = (bmax.y - o.y) * inv.y;
    float t1z = (bmin.z - o.z) * inv.z, t2z = (bmax.z - o.z) * inv.z;
    // the instructions are named explicitly: fminf/fmaxf make hipcc add a v_max_f32 x,x canonicalisation per operand
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
__device__ __forceinline__ bool has_infinite_component(V3 inv)
{
    return __builtin_isinf(inv.x) || __builtin_isinf(inv.y) || __builtin_isinf(inv.z);
}
// keeps a loaded value live at this point so the compiler issues all loads of a record together instead of sinking some
// of them behind a branch (a second dependent memory round trip)
__device__ __forceinline__ void keep_loaded(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

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
    for (;;) {
        if (code & kLeafBit) {
            uint32_t i = code & ~kLeafBit;
            for (;;) {
                const float4* rec = sc.tri_leaf + 3u * (size_t)i;
                float4 a = rec[0], b = rec[1], c = rec[2];
                keep_loaded(a); keep_loaded(b); keep_loaded(c);
                if (COUNT) cnt.tris++;
                if (intersect_triangle(mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), o, d, ray_t)) {
                    tri_idx = __float_as_uint(c.y);
                    result = true;
                }
                if (__float_as_uint(c.z) != 0u) break;
                ++i;
            }
            if (sp == 0) break;
            code = stack[(--sp) * stack_stride];
            continue;
        }
        const float4* pair = sc.node_pairs + 4u * (size_t)code;
        float4 lmin = pair[0], lmax = pair[1], rmin = pair[2], rmax = pair[3];
        if (COUNT) cnt.inner++;
        float left_dist, right_dist;
        if (__builtin_amdgcn_ballot_w64(exact_slab) == 0ull) {            // wave-uniform: nobody needs the NaN-exact form
            left_dist = intersect_aabb_finite(lmin, lmax, o, inv, ray_t);
            right_dist = intersect_aabb_finite(rmin, rmax, o, inv, ray_t);
        } else {
            left_dist = intersect_aabb(lmin, lmax, o, inv, ray_t);
            right_dist = intersect_aabb(rmin, rmax, o, inv, ray_t);
        }
        uint32_t left_code = __float_as_uint(lmin.w), right_code = __float_as_uint(rmin.w);
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

}  // namespace dev
}  // namespace cgpt
