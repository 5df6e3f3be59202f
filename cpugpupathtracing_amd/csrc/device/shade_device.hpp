// shade_device.hpp -- one bounce of TracePathAdvanced (ref: Source/Main.cpp:404-573) as a device function shared by the
// megakernel and the wavefront shade kernel.  Given the closest-hit record of the extend ray it performs, in the
// reference's order (RNG draw order: SURVEY Appendix C): hit reconstruction, emissive add, NEE light sample (-> shadow ray
// + pending contribution), Russian roulette, lobe choice (mirror / dielectric + Beer / diffuse) -> next ray.
// The shadow ray draws no random numbers, so computing the next ray before the shadow ray is traced changes nothing.
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.h"
#include "rt_device.hpp"

namespace cgpt {
namespace dev {

struct Hit { V3 pos, normal; uint32_t mat; };

// GetRayHitResult (ref: Main.cpp:325-338): flat shading normal = v0.normal of the hit triangle (SURVEY A-8)
template <bool COUNT>
__device__ __forceinline__ Hit get_hit(const DevScene& sc, const Ray& ray, Counters& cnt)
{
    Hit h;
    h.pos = ray.o + ray.d * ray.t;
    const DevObject& obj = sc.objects[ray.obj];
    if (obj.kind == 0u) {
        const float4 n = sc.tri_normal[obj.tri_base + ray.tri];
        h.normal = mk(n.x, n.y, n.z);
        if (COUNT) cnt.hits++;
    } else if (obj.kind == 1u) {
        h.normal = normalize(h.pos - mk(obj.sphere_center));                 // ref: Primitives.cpp:153-156
    } else {
        h.normal = mk(obj.plane_normal);                                     // ref: Primitives.cpp:158-161
    }
    h.mat = obj.mat_index;
    return h;
}

struct LightSample { V3 to_light, normal, emission; float distance, area; };

// GetRandomLightSourceForSample (ref: Main.cpp:351-394)
__device__ __forceinline__ LightSample sample_light(const DevScene& sc, uint32_t& rng, V3 hit_pos)
{
    LightSample ls;
    const uint32_t light_obj = sc.lights[random_range(rng, 0u, sc.n_lights - 1u)];
    const DevObject& light = sc.objects[light_obj];
    V3 pos;
    if (light.kind == 0u) {                                                   // mesh light, ref: Main.cpp:360-368
        const uint32_t t = random_range(rng, 0u, light.n_tris - 1u);
        const float4* rec = sc.tri_orig + 3u * (size_t)(light.tri_base + t);
        float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        float u0 = random_float(rng);                                         // RandomPointTriangle, ref: Primitives.cpp:170-186
        float u1 = random_float(rng);
        float alpha = u0, beta = u1;
        if (alpha + beta > 1.0f) { alpha = 1.0f - alpha; beta = 1.0f - beta; }
        float gamma = 1.0f - beta - alpha;
        pos = alpha * mk(r0.x, r0.y, r0.z) + beta * mk(r1.x, r1.y, r1.z) + gamma * mk(r2.x, r2.y, r2.z);
        ls.normal = mk(r0.w, r1.w, r2.w);
        ls.area = light.total_area / 2.0f;
    } else {                                                                  // sphere light, ref: Main.cpp:371-380
        V3 c = mk(light.sphere_center);
        V3 to_pos = normalize(hit_pos - c);                                   // RandomPointSphereFacing, ref: Primitives.cpp:214-220
        V3 dir = uniform_hemisphere_sample(rng, to_pos);
        pos = c + light.sphere_radius * dir;
        ls.normal = normalize(pos - c);
        ls.area = 2.0f * kPi * light.sphere_radius_sq;
    }
    ls.to_light = pos - hit_pos;
    ls.distance = length(ls.to_light);
    ls.to_light = normalize(ls.to_light);
    const float4* mp = sc.materials + 4u * (size_t)light.mat_index;
    float4 c2 = mp[2], c3 = mp[3];
    ls.emission = mk(c2.y, c2.z, c2.w) * c3.x;                                // emissive * intensity
    return ls;
}

struct PathState {
    V3 throughput, energy;
    uint32_t rng, depth;
    bool is_specular;
};

enum : uint32_t { kBounceTerminate = 1u, kBounceShadow = 2u, kBounceEnergy = 4u, kBounceBruteDone = 8u };   // kBounceEnergy: ps.energy was added to; kBounceBruteDone (wavefront shade): ps.energy is a finished TracePath's radiance

// Processes the hit of `ray` (already traced).  On return: `ray` is the next extend ray unless kBounceTerminate is set;
// if kBounceShadow is set, `shadow` / `pending` describe the NEE connection to trace (energy += pending when unoccluded,
// ref: Main.cpp:452-463).  Emissive energy is added here; the final debug-view overrides are applied by the caller.
template <bool COUNT>
__device__ __forceinline__ uint32_t shade_bounce(const DevScene& sc, const DevSettings& st, Ray& ray, PathState& ps, Ray& shadow,
                                                 V3& pending, Counters& cnt)
{
    if (ps.depth == 0 && st.debug_mode == 2u) {                               // BVH-depth view, ref: Main.cpp:408-412
        ps.energy = ps.energy + lerp(mk(0.0f, 1.0f, 0.0f), mk(1.0f, 0.0f, 0.0f), (float)ray.bvh_depth / 30.0f);
        return kBounceTerminate | kBounceEnergy;
    }
    if (ray.obj == kNoHit) return kBounceTerminate;                           // ref: Main.cpp:415-416

    const Hit hit = get_hit<COUNT>(sc, ray, cnt);
    const Mat mat = load_material(sc, hit.mat);
    if (mat.is_light) {                                                       // ref: Main.cpp:424-431
        if (!st.nee || ps.depth == 0 || ps.is_specular) {
            ps.energy = ps.energy + ps.throughput * mat.emissive * mat.intensity;
            return kBounceTerminate | kBounceEnergy;
        }
        return kBounceTerminate;
    }

    uint32_t result = 0;
    const float diffuse_weight = max_std(0.0f, 1.0f - mat.specular - mat.refractivity);
    if (sc.n_lights > 0 && st.nee && diffuse_weight > 0.001f) {               // ref: Main.cpp:439-465
        const LightSample ls = sample_light(sc, ps.rng, hit.pos);
        const float NdotL = dot(hit.normal, ls.to_light);
        const float NLdotL = dot(ls.normal, -ls.to_light);
        if (NdotL > 0.0f && NLdotL > 0.0f) {
            shadow = make_ray(hit.pos + ls.to_light * kNudge, ls.to_light, ls.distance - 2.0f * kNudge);
            const V3 brdf_diffuse = mat.albedo * kInvPi;
            const float solid_angle = (NLdotL * ls.area) / (ls.distance * ls.distance);
            const float light_pdf = 1.0f / solid_angle;
            pending = ps.throughput * (NdotL / light_pdf) * brdf_diffuse * ls.emission * (float)sc.n_lights * diffuse_weight;
            result |= kBounceShadow;
        }
    }

    // Russian roulette on albedo (ref: Main.cpp:468-475); the float is drawn even when p == 1
    if (st.rr) {
        const float p = survival_probability_rr(mat.albedo);
        if (p < random_float(ps.rng)) return result | kBounceTerminate;
        ps.throughput = ps.throughput * mk(1.0f / p);
    }

    const float r = random_float(ps.rng);                                     // ref: Main.cpp:478
    if (r < mat.specular) {                                                   // mirror, ref: Main.cpp:480-487
        const V3 sd = reflect(ray.d, hit.normal);
        ray = make_ray(hit.pos + sd * kNudge, sd, 1e34f);
        ps.throughput = ps.throughput * mat.albedo;
        ps.is_specular = true;
    } else if (r < mat.specular + mat.refractivity) {                         // dielectric, ref: Main.cpp:488-546
        V3 N = hit.normal;
        float cosi = clamp_std(dot(N, ray.d), -1.0f, 1.0f);
        float etai = 1.0f, etat = mat.ior;
        bool inside = true;
        if (cosi < 0.0f) { cosi = -cosi; inside = false; }
        else { float tmp = etai; etai = etat; etat = tmp; N = -N; }
        const float eta = etai / etat;
        const float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
        if (k >= 0.0f) {
            const V3 rd = refract(ray.d, N, eta, cosi, k);
            const float angle_in = dot(ray.d, hit.normal);
            const float angle_out = dot(rd, hit.normal);
            const float Fr = fresnel(angle_in, angle_out, etai, etat);
            if (random_float(ps.rng) > Fr) {
                ps.throughput = ps.throughput * mat.albedo;
                if (inside) {                                                 // Beer's law on the way out only (SURVEY A-4)
                    V3 ab;
                    ab.x = expf(-mat.absorption.x * ray.t);
                    ab.y = expf(-mat.absorption.y * ray.t);
                    ab.z = expf(-mat.absorption.z * ray.t);
                    ps.throughput = ps.throughput * ab;
                }
                ray = make_ray(hit.pos + rd * kNudge, rd, 1e34f);
                ps.is_specular = true;
            } else {
                const V3 sd = reflect(ray.d, hit.normal);
                ray = make_ray(hit.pos + sd * kNudge, sd, 1e34f);
                ps.throughput = ps.throughput * mat.albedo;
                ps.is_specular = true;
            }
        }
        // k < 0 (total internal reflection): the ray is left as it is -- t, obj, tri included -- and is traced again
        // next iteration (SURVEY A-3)
    } else {                                                                  // diffuse, ref: Main.cpp:547-570
        V3 dd; float NdotR, pdf;
        if (st.cosine) {
            dd = cosine_weighted_diffuse_reflection(ps.rng, hit.normal);
            NdotR = dot(dd, hit.normal);
            pdf = 1.0f / (2.0f * kPi);                                        // swapped pdfs kept: SURVEY A-7
        } else {
            dd = uniform_hemisphere_sample(ps.rng, hit.normal);
            NdotR = dot(dd, hit.normal);
            pdf = NdotR / kPi;
        }
        ray = make_ray(hit.pos + dd * kNudge, dd, 1e34f);
        ps.throughput = ps.throughput * ((NdotR / pdf) * (mat.albedo * kInvPi));
        ps.is_specular = false;
    }
    ps.depth++;
    if ((int32_t)ps.depth > st.max_ray_depth) result |= kBounceTerminate;     // loop condition, ref: Main.cpp:404
    return result;
}

// ---- brute-force integrator: TracePath (ref: Source/Main.cpp:581-689) ----------------------------------------------------
// The reference recurses; every level applies one multiplicative operation to what its single child returns, innermost
// first.  Float multiplication is not associative, so the chain is recorded on the way down (one BruteLevel per bounce) and
// applied on the way back up in the reference's order, which keeps the result bit-identical to the recursion.
struct BruteLevel {
    uint32_t kind;      // 0: L = 0 + albedo*L (mirror, dielectric reflect/refract from outside)   ref: Main.cpp:618,656,672
                        // 1: same, then L *= absorption (refract out of the medium, Beer)          ref: Main.cpp:658-666
                        // 2: L = 0 + (2*pi*brdf) * (cosi*L) (uniform-hemisphere diffuse)            ref: Main.cpp:679-685
    V3 a; float cosi; V3 absorb;
};
static constexpr uint32_t kMaxBruteLevels = 32;   // max_ray_depth + 1 levels are kept in per-lane scratch

enum : uint32_t { kBruteContinue = 0u, kBruteLeaf = 1u };

// One TracePath level after IntersectScene(ray): either a leaf (returns its radiance in `leaf`) or a bounce (fills `level`,
// replaces `ray` by the child ray).  RNG draw order as in the reference: r, then Fresnel choice or the hemisphere sample.
template <bool COUNT>
__device__ __forceinline__ uint32_t brute_bounce(const DevScene& sc, const DevSettings& st, Ray& ray, uint32_t& rng, uint32_t depth,
                                                 BruteLevel& level, V3& leaf, Counters& cnt)
{
    if (depth == 0 && st.debug_mode == 2u) {                                  // ref: Main.cpp:594-597
        leaf = lerp(mk(0.0f, 1.0f, 0.0f), mk(1.0f, 0.0f, 0.0f), (float)ray.bvh_depth / 30.0f);
        return kBruteLeaf;
    }
    if (ray.obj == kNoHit) { leaf = mk(0.0f); return kBruteLeaf; }            // ref: Main.cpp:600-601
    const Hit hit = get_hit<COUNT>(sc, ray, cnt);
    const Mat mat = load_material(sc, hit.mat);
    if (mat.is_light) { leaf = mat.emissive * mat.intensity; return kBruteLeaf; }   // ref: Main.cpp:606-609

    const float r = random_float(rng);                                        // ref: Main.cpp:611
    level.cosi = 0.0f; level.absorb = mk(1.0f);
    if (r < mat.specular) {                                                   // ref: Main.cpp:614-619
        const V3 sd = reflect(ray.d, hit.normal);
        ray = make_ray(hit.pos + sd * kNudge, sd, 1e34f);
        level.kind = 0u; level.a = mat.albedo;
    } else if (r < mat.specular + mat.refractivity) {                         // ref: Main.cpp:621-675
        V3 N = hit.normal;
        float cosi = clamp_std(dot(N, ray.d), -1.0f, 1.0f);
        float etai = 1.0f, etat = mat.ior;
        bool inside = true;
        if (cosi < 0.0f) { cosi = -cosi; inside = false; }
        else { float tmp = etai; etai = etat; etat = tmp; N = -N; }
        const float eta = etai / etat;
        const float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
        if (!(k >= 0.0f)) { leaf = mk(0.0f); return kBruteLeaf; }             // total internal reflection: black (ref: Main.cpp:645)
        const V3 rd = refract(ray.d, N, eta, cosi, k);
        const float angle_in = dot(ray.d, hit.normal);
        const float angle_out = dot(rd, hit.normal);
        const float Fr = fresnel(angle_in, angle_out, etai, etat);
        if (random_float(rng) > Fr) {
            level.kind = inside ? 1u : 0u; level.a = mat.albedo;
            if (inside) {
                level.absorb.x = expf(-mat.absorption.x * ray.t);
                level.absorb.y = expf(-mat.absorption.y * ray.t);
                level.absorb.z = expf(-mat.absorption.z * ray.t);
            }
            ray = make_ray(hit.pos + rd * kNudge, rd, 1e34f);
        } else {
            const V3 sd = reflect(ray.d, hit.normal);
            ray = make_ray(hit.pos + sd * kNudge, sd, 1e34f);
            level.kind = 0u; level.a = mat.albedo;
        }
    } else {                                                                  // ref: Main.cpp:677-686
        const V3 dd = uniform_hemisphere_sample(rng, hit.normal);
        level.kind = 2u;
        level.cosi = dot(dd, hit.normal);
        level.a = (2.0f * kPi) * (mat.albedo * kInvPi);
        ray = make_ray(hit.pos + dd * kNudge, dd, 1e34f);
    }
    return kBruteContinue;
}

// the parent's operation on its child's radiance L
__device__ __forceinline__ V3 brute_apply(const BruteLevel& lv, V3 L)
{
    if (lv.kind == 2u) {
        const V3 irr = mk(L.x * lv.cosi, L.y * lv.cosi, L.z * lv.cosi);
        return mk(0.0f) + lv.a * irr;
    }
    V3 out = mk(0.0f) + lv.a * L;
    if (lv.kind == 1u) out = out * lv.absorb;
    return out;
}

// final colour of a finished path (ray-depth debug view, ref: Main.cpp:575-576)
__device__ __forceinline__ V3 final_energy(const DevSettings& st, const PathState& ps)
{
    if (st.debug_mode == 1u) return lerp(mk(0.0f, 1.0f, 0.0f), mk(1.0f, 0.0f, 0.0f), (float)ps.depth / (float)st.max_ray_depth);
    return ps.energy;
}

}  // namespace dev
}  // namespace cgpt
