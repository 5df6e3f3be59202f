// path_kernels.hip -- the render kernels (gfx950).
//
// megakernel<COUNT>: one lane = one pixel, the sample loop inside the kernel, the whole TracePathAdvanced state machine
// (ref: Source/Main.cpp:396-579) in registers.  The loop is flattened so that every iteration traces exactly ONE ray per
// lane -- an extend ray or the NEE shadow ray -- and a lane whose path ended starts its next sample in the same
// iteration, so the 64 lanes of a wave stay in the traversal loop together instead of idling at bounce boundaries.
// Replaces: Render()'s tile loop + ThreadPool::Dispatch (ref: Main.cpp:699-754).
//
// intersect_rays_kernel: IntersectScene on a ray batch (the extend step alone), for bit-exact hit-record parity tests.
#include <hip/hip_runtime.h>

#include "device_scene.h"
#include "rt_device.hpp"

namespace cgpt {

using namespace dev;

extern __shared__ uint32_t lds_stack[];

struct Hit { V3 pos, normal; uint32_t mat; };

// GetRayHitResult (ref: Main.cpp:325-338): flat shading normal = v0.normal of the hit triangle (SURVEY A-8)
template <bool COUNT>
__device__ __forceinline__ Hit get_hit(const DevScene& sc, const Ray& ray, Counters& cnt)
{
    Hit h;
    h.pos = ray.o + ray.d * ray.t;
    const DevObject& obj = sc.objects[ray.obj];
    if (obj.kind == 0u) {
        const float4* rec = sc.tri_orig + 3u * (size_t)(obj.tri_base + ray.tri);
        h.normal = mk(rec[0].w, rec[1].w, rec[2].w);
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

// GetRandomLightSourceForSample (ref: Main.cpp:351-394); draw order per SURVEY Appendix C
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

template <bool COUNT>
__global__ void __launch_bounds__(256) megakernel(const DevRenderArgs args)
{
    const DevScene& sc = args.scene;
    uint32_t* const stack = lds_stack + threadIdx.x;
    const uint32_t stride = blockDim.x;

    const uint32_t tiles_x = (args.width + 15u) / 16u;
    const uint32_t tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t px = tx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t py = args.row_begin + ty * 16u + (wave >> 1) * 8u + (lane >> 3);
    const bool active = px < args.width && py < args.row_end;

    Counters cnt = { 0, 0, 0, 0, 0 };
    double energy_sum = 0.0;

    if (active) {
        const DevSettings& st = args.settings;
        const uint32_t pixel_index = py * args.width + px;                    // global index: RNG key, same for any tiling
        const size_t local_index = (size_t)(py - args.row_begin) * args.width + px;
        const float screen_u = (float)px * (1.0f / (float)args.width);        // ref: Main.cpp:700,713-714
        const float screen_v = (float)py * (1.0f / (float)args.height);
        float4 acc = args.accumulator[local_index];
        V3 last_color = mk(0.0f);

        uint32_t s = args.first_sample;
        const uint32_t s_end = args.first_sample + args.n_samples;

        // path state (TracePathAdvanced locals, ref: Main.cpp:398-402)
        Ray ray = make_ray(mk(0.0f), mk(0.0f), 0.0f);
        V3 throughput = mk(1.0f), energy = mk(0.0f);
        uint32_t depth = 0, rng = 0;
        bool is_specular = false, need_new = true, shadow_kind = false;
        // state kept across the shadow-ray trace
        Ray sray = make_ray(mk(0.0f), mk(0.0f), 0.0f);
        V3 pending = mk(0.0f);
        Hit hit; hit.pos = mk(0.0f); hit.normal = mk(0.0f); hit.mat = 0;

        for (;;) {
            if (need_new) {
                if (s == s_end) break;
                rng = pcg_seed(pixel_index, s, args.seed);
                ray = camera_ray(args.camera, screen_u, screen_v);            // no jitter: SURVEY A-14
                throughput = mk(1.0f); energy = mk(0.0f);
                depth = 0; is_specular = false; need_new = false; shadow_kind = false;
            }

            // ---- one ray per iteration: the extend ray or the pending shadow ray ----
            Ray cur = shadow_kind ? sray : ray;
            intersect_scene<COUNT>(sc, cur, stack, stride, cnt);

            bool terminate = false;
            if (!shadow_kind) {
                ray.t = cur.t; ray.obj = cur.obj; ray.tri = cur.tri; ray.bvh_depth = cur.bvh_depth;

                if (depth == 0 && st.debug_mode == 2u) {                      // ref: Main.cpp:408-412
                    energy = energy + lerp(mk(0.0f, 1.0f, 0.0f), mk(1.0f, 0.0f, 0.0f), (float)ray.bvh_depth / 30.0f);
                    terminate = true;
                } else if (ray.obj == kNoHit) {                               // ref: Main.cpp:415-416
                    terminate = true;
                } else {
                    hit = get_hit<COUNT>(sc, ray, cnt);
                    const Mat mat = load_material(sc, hit.mat);
                    if (mat.is_light) {                                       // ref: Main.cpp:424-431
                        if (!st.nee || depth == 0 || is_specular) energy = energy + throughput * mat.emissive * mat.intensity;
                        terminate = true;
                    } else {
                        const float diffuse_weight = max_std(0.0f, 1.0f - mat.specular - mat.refractivity);
                        if (sc.n_lights > 0 && st.nee && diffuse_weight > 0.001f) {          // ref: Main.cpp:439-465
                            const LightSample ls = sample_light(sc, rng, hit.pos);
                            const float NdotL = dot(hit.normal, ls.to_light);
                            const float NLdotL = dot(ls.normal, -ls.to_light);
                            if (NdotL > 0.0f && NLdotL > 0.0f) {
                                sray = make_ray(hit.pos + ls.to_light * kNudge, ls.to_light, ls.distance - 2.0f * kNudge);
                                const V3 brdf_diffuse = mat.albedo * kInvPi;
                                const float solid_angle = (NLdotL * ls.area) / (ls.distance * ls.distance);
                                const float light_pdf = 1.0f / solid_angle;
                                pending = throughput * (NdotL / light_pdf) * brdf_diffuse * ls.emission * (float)sc.n_lights * diffuse_weight;
                                shadow_kind = true;
                                continue;                                     // trace the shadow ray next iteration
                            }
                        }
                    }
                }
            } else {
                if (cur.obj == kNoHit) energy = energy + pending;             // ref: Main.cpp:454-463
                shadow_kind = false;
            }

            if (!terminate) {
                const Mat mat = load_material(sc, hit.mat);
                // Russian roulette on albedo (ref: Main.cpp:468-475); the float is drawn even when p == 1
                bool alive = true;
                if (st.rr) {
                    const float p = survival_probability_rr(mat.albedo);
                    if (p < random_float(rng)) alive = false;
                    else throughput = throughput * mk(1.0f / p);
                }
                if (!alive) {
                    terminate = true;
                } else {
                    const float r = random_float(rng);                        // ref: Main.cpp:478
                    if (r < mat.specular) {                                   // mirror, ref: Main.cpp:480-487
                        const V3 sd = reflect(ray.d, hit.normal);
                        ray = make_ray(hit.pos + sd * kNudge, sd, 1e34f);
                        throughput = throughput * mat.albedo;
                        is_specular = true;
                    } else if (r < mat.specular + mat.refractivity) {         // dielectric, ref: Main.cpp:488-546
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
                            if (random_float(rng) > Fr) {
                                throughput = throughput * mat.albedo;
                                if (inside) {                                 // Beer's law on the way out only (SURVEY A-4)
                                    V3 ab;
                                    ab.x = expf(-mat.absorption.x * ray.t);
                                    ab.y = expf(-mat.absorption.y * ray.t);
                                    ab.z = expf(-mat.absorption.z * ray.t);
                                    throughput = throughput * ab;
                                }
                                ray = make_ray(hit.pos + rd * kNudge, rd, 1e34f);
                                is_specular = true;
                            } else {
                                const V3 sd = reflect(ray.d, hit.normal);
                                ray = make_ray(hit.pos + sd * kNudge, sd, 1e34f);
                                throughput = throughput * mat.albedo;
                                is_specular = true;
                            }
                        }
                        // k < 0 (total internal reflection): the ray is left as it is, t included, and is traced
                        // again next iteration (SURVEY A-3)
                    } else {                                                  // diffuse, ref: Main.cpp:547-570
                        V3 dd; float NdotR, pdf;
                        if (st.cosine) {
                            dd = cosine_weighted_diffuse_reflection(rng, hit.normal);
                            NdotR = dot(dd, hit.normal);
                            pdf = 1.0f / (2.0f * kPi);                        // swapped pdfs kept: SURVEY A-7
                        } else {
                            dd = uniform_hemisphere_sample(rng, hit.normal);
                            NdotR = dot(dd, hit.normal);
                            pdf = NdotR / kPi;
                        }
                        ray = make_ray(hit.pos + dd * kNudge, dd, 1e34f);
                        throughput = throughput * ((NdotR / pdf) * (mat.albedo * kInvPi));
                        is_specular = false;
                    }
                    depth++;
                    if ((int32_t)depth > st.max_ray_depth) terminate = true;  // loop condition, ref: Main.cpp:404
                }
            }

            if (terminate) {
                if (st.debug_mode == 1u)                                      // ref: Main.cpp:575-576
                    energy = lerp(mk(0.0f, 1.0f, 0.0f), mk(1.0f, 0.0f, 0.0f), (float)depth / (float)st.max_ray_depth);
                energy_sum += (double)(energy.x + energy.y + energy.z) * 0.001;   // ref: Main.cpp:735
                if (st.debug_mode == 0u) { acc.x += energy.x; acc.y += energy.y; acc.z += energy.z; acc.w += 1.0f; }   // ref: Main.cpp:740
                else last_color = energy;
                ++s;
                need_new = true;
            }
        }

        if (args.n_samples > 0) {
            if (st.debug_mode == 0u) {                                        // ref: Main.cpp:738-746
                args.accumulator[local_index] = acc;
                const float n = (float)s_end;                                 // data.num_accumulated after these frames
                args.pixels[local_index] = vec4_to_uint(acc.x / n, acc.y / n, acc.z / n);
            } else {
                args.pixels[local_index] = vec4_to_uint(last_color.x, last_color.y, last_color.z);
            }
        }
    }

    // one atomic per wave per counter
    wave_add_u64(&args.counters->traced_rays, cnt.rays);
    wave_add_f64(&args.counters->total_energy, energy_sum);
    if (COUNT) {
        wave_add_u64(&args.counters->inner_steps, cnt.inner);
        wave_add_u64(&args.counters->tri_tests, cnt.tris);
        wave_add_u64(&args.counters->bvh_depth_sum, cnt.depth);
        wave_add_u64(&args.counters->closest_hits, cnt.hits);
    }
}

// host-side launcher (the ABI translation unit calls plain C++ functions, kernels stay in this one)
hipError_t LaunchMegakernel(const DevRenderArgs& args, bool count, hipStream_t stream)
{
    const uint32_t tiles_x = (args.width + 15u) / 16u, tiles_y = (args.row_end - args.row_begin + 15u) / 16u;
    const dim3 grid(tiles_x * tiles_y), block(256);
    const size_t lds = (size_t)args.scene.stack_depth * 256 * sizeof(uint32_t);
    if (count) hipLaunchKernelGGL(megakernel<true>, grid, block, lds, stream, args);
    else hipLaunchKernelGGL(megakernel<false>, grid, block, lds, stream, args);
    return hipGetLastError();
}

// IntersectScene on a batch of rays (ref: Main.cpp:299-316)
__global__ void __launch_bounds__(256) intersect_rays_kernel(const DevScene sc, const float* __restrict__ origins,
                                                             const float* __restrict__ dirs, const float* __restrict__ tmax,
                                                             uint32_t n, float* __restrict__ out_t, uint32_t* __restrict__ out_obj,
                                                             uint32_t* __restrict__ out_tri, uint32_t* __restrict__ out_depth,
                                                             DevCounters* counters)
{
    uint32_t* const stack = lds_stack + threadIdx.x;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Counters cnt = { 0, 0, 0, 0, 0 };
    if (i < n) {
        Ray ray = make_ray(mk(origins + 3 * (size_t)i), mk(dirs + 3 * (size_t)i), tmax ? tmax[i] : 1e34f);
        intersect_scene<true>(sc, ray, stack, blockDim.x, cnt);
        out_t[i] = ray.t; out_obj[i] = ray.obj; out_tri[i] = ray.tri; out_depth[i] = ray.bvh_depth;
    }
    wave_add_u64(&counters->traced_rays, cnt.rays);
    wave_add_u64(&counters->inner_steps, cnt.inner);
    wave_add_u64(&counters->tri_tests, cnt.tris);
    wave_add_u64(&counters->bvh_depth_sum, cnt.depth);
}

hipError_t LaunchIntersectRays(const DevScene& sc, const float* origins, const float* dirs, const float* tmax, uint32_t n, float* out_t,
                               uint32_t* out_obj, uint32_t* out_tri, uint32_t* out_depth, DevCounters* counters, hipStream_t stream)
{
    const size_t lds = (size_t)sc.stack_depth * 256 * sizeof(uint32_t);
    hipLaunchKernelGGL(intersect_rays_kernel, dim3((n + 255u) / 256u), dim3(256), lds, stream, sc, origins, dirs, tmax, n, out_t, out_obj,
                       out_tri, out_depth, counters);
    return hipGetLastError();
}

}  // namespace cgpt
