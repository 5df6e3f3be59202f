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

#include <algorithm>
#include <cstdlib>

#include "device_scene.h"
#include "rt_device.hpp"
#include "shade_device.hpp"

namespace cgpt {

using namespace dev;

extern __shared__ uint32_t lds_stack[];

// BRUTE: lanes whose pixel uses the brute-force integrator (RENDER_MODE_BRUTE_FORCE, or the left half of the image in
// RENDER_MODE_COMPARISON, ref: Main.cpp:719-729) run TracePath (ref: Main.cpp:581-689) instead; its per-level operations
// live in per-lane scratch, so the plain TracePathAdvanced instantiation carries no scratch at all.
// Block shape: 256 threads = a 16x16-pixel tile (the reference's job size, ref: Main.cpp:705-711), or -- one-sample calls -- 64 threads =
// one 8x8 tile per single-wave block: the wave's slot and its LDS are free the moment its own longest path ends instead of its block's,
// and the dispatcher places single waves (1080p, one sample: 1.71 -> 1.5x ms, profiles/r03/one_sample.md).
template <bool COUNT, bool BRUTE>
__global__ void __launch_bounds__(256) megakernel(const DevRenderArgs args)
{
    const DevScene& sc = args.scene;
    uint32_t* const stack = lds_stack + threadIdx.x;
    const uint32_t stride = blockDim.x;

    const bool tile8 = blockDim.x == 64u;
    const uint32_t tiles_x = tile8 ? (args.width + 7u) / 8u : (args.width + 15u) / 16u;
    const uint32_t tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t px = tile8 ? tx * 8u + (lane & 7u) : tx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t local_row = tile8 ? ty * 8u + (lane >> 3) : ty * 16u + (wave >> 1) * 8u + (lane >> 3);
    const bool active = px < args.width && local_row < args.n_rows;
    const uint32_t py = GlobalRow(local_row, args.band_first, args.band_h, args.band_stride);

    Counters cnt = { 0, 0, 0, 0, 0 };
    double energy_sum = 0.0;

    if (active) {
        const DevSettings& st = args.settings;
        const uint32_t pixel_index = py * args.width + px;                    // global index: RNG key, same for any tiling
        const size_t local_index = (size_t)local_row * args.width + px;
        const float screen_u = (float)px * (1.0f / (float)args.width);        // ref: Main.cpp:700,713-714
        const float screen_v = (float)py * (1.0f / (float)args.height);
        float4 acc = args.accumulator[local_index];
        V3 last_color = mk(0.0f);

        uint32_t s = args.first_sample;
        const uint32_t s_end = args.first_sample + args.n_samples;

        // path state (TracePathAdvanced locals, ref: Main.cpp:398-402)
        Ray ray = make_ray(mk(0.0f), mk(0.0f), 0.0f);
        PathState ps; ps.throughput = mk(1.0f); ps.energy = mk(0.0f); ps.rng = 0; ps.depth = 0; ps.is_specular = false;
        bool need_new = true, shadow_kind = false, dead = false;
        Ray sray = make_ray(mk(0.0f), mk(0.0f), 0.0f);                       // pending NEE connection
        V3 pending = mk(0.0f);
        const bool use_brute = BRUTE && (st.render_mode == 1u || (st.render_mode == 0u && px < args.width / 2u));
        BruteLevel levels[BRUTE ? kMaxBruteLevels : 1u];
        uint32_t n_levels = 0;

        for (;;) {
            if (need_new) {
                if (s == s_end) break;
                ps.rng = pcg_seed(pixel_index, s, args.seed);
                ray = camera_ray(args.camera, screen_u, screen_v);            // no jitter: SURVEY A-14
                ps.throughput = mk(1.0f); ps.energy = mk(0.0f);
                ps.depth = 0; ps.is_specular = false; need_new = false; shadow_kind = false; dead = false;
                n_levels = 0;
            }

            // ---- one ray per iteration: the extend ray or the pending shadow ray ----
            Ray cur = shadow_kind ? sray : ray;
            intersect_scene<COUNT>(sc, cur, stack, stride, cnt);

            bool finalize;
            if (BRUTE && use_brute) {
                ray.t = cur.t; ray.obj = cur.obj; ray.tri = cur.tri; ray.bvh_depth = cur.bvh_depth;
                BruteLevel lv; V3 leaf = mk(0.0f);
                finalize = brute_bounce<COUNT>(sc, st, ray, ps.rng, ps.depth, lv, leaf, cnt) == kBruteLeaf;
                if (!finalize) {
                    levels[n_levels++] = lv;
                    ps.depth++;
                    if ((int32_t)ps.depth > st.max_ray_depth) finalize = true;   // the child returns black before tracing (ref: Main.cpp:589-590)
                }
                if (finalize) {
                    V3 L = leaf;
                    for (uint32_t k = n_levels; k-- > 0u;) L = brute_apply(levels[k], L);
                    ps.energy = L;
                }
            } else if (shadow_kind) {
                if (cur.obj == kNoHit) ps.energy = ps.energy + pending;       // ref: Main.cpp:454-463
                shadow_kind = false;
                finalize = dead;
            } else {
                ray.t = cur.t; ray.obj = cur.obj; ray.tri = cur.tri; ray.bvh_depth = cur.bvh_depth;
                const uint32_t flags = shade_bounce<COUNT>(sc, st, ray, ps, sray, pending, cnt);
                dead = (flags & kBounceTerminate) != 0;
                shadow_kind = (flags & kBounceShadow) != 0;
                finalize = dead && !shadow_kind;
            }

            if (finalize) {
                const V3 e = (BRUTE && use_brute) ? ps.energy : final_energy(st, ps);
                energy_sum += (double)(e.x + e.y + e.z) * 0.001;              // ref: Main.cpp:735
                if (st.debug_mode == 0u) { acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += 1.0f; }   // ref: Main.cpp:740
                else last_color = e;
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
#ifdef CGPT_STEP_MAP
            if (COUNT) args.pixels[local_index] = cnt.inner + cnt.tris;       // diagnostic build (scripts/gpu_step_map.py): dependent fetches of this pixel's paths
#endif
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
static uint32_t MegakernelBlockThreads(const DevRenderArgs& args) { return args.n_samples == 1u ? 64u : 256u; }

hipError_t LaunchMegakernel(const DevRenderArgs& args, bool count, hipStream_t stream)
{
    const bool brute = args.settings.render_mode != 2u;
    static const uint32_t env_block = getenv("CGPT_MEGA_BLOCK") ? (uint32_t)atoi(getenv("CGPT_MEGA_BLOCK")) : 0u;   // experiments: 64 or 256 for every call
    const uint32_t bt = env_block == 64u || env_block == 256u ? env_block : MegakernelBlockThreads(args);
    const uint32_t edge = bt == 64u ? 8u : 16u;
    const uint32_t tiles_x = (args.width + edge - 1u) / edge, tiles_y = (args.n_rows + edge - 1u) / edge;
    const dim3 grid(tiles_x * tiles_y), block(bt);
    const size_t lds = (size_t)args.scene.stack_depth * bt * sizeof(uint32_t);
    if (brute) {
        if (count) hipLaunchKernelGGL((megakernel<true, true>), grid, block, lds, stream, args);
        else hipLaunchKernelGGL((megakernel<false, true>), grid, block, lds, stream, args);
    } else {
        if (count) hipLaunchKernelGGL((megakernel<true, false>), grid, block, lds, stream, args);
        else hipLaunchKernelGGL((megakernel<false, false>), grid, block, lds, stream, args);
    }
    return hipGetLastError();
}

uint32_t MegakernelWavesPerSimd(const DevRenderArgs& args)
{
    int b = 0;
    const uint32_t bt = MegakernelBlockThreads(args);
    const size_t lds = (size_t)args.scene.stack_depth * bt * sizeof(uint32_t);
    const hipError_t e = args.settings.render_mode != 2u ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (megakernel<false, true>), (int)bt, lds)
                                                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (megakernel<false, false>), (int)bt, lds);
    return e == hipSuccess && b > 0 ? std::max(1u, (uint32_t)b * bt / 256u) : 1u;    // blocks per CU -> waves per SIMD (4 SIMDs)
}

// data.pixels from data.accumulator / data.num_accumulated (ref: Main.cpp:741), for an accumulator restored from a checkpoint
__global__ void __launch_bounds__(256) pack_pixels_kernel(const float4* __restrict__ accumulator, uint32_t* __restrict__ pixels, size_t n_pixels, float n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pixels) {
        const float4 a = accumulator[i];
        pixels[i] = vec4_to_uint(a.x / n, a.y / n, a.z / n);
    }
}

hipError_t LaunchPackPixels(const float4* accumulator, uint32_t* pixels, size_t n_pixels, uint32_t num_accumulated, hipStream_t stream)
{
    if (n_pixels == 0 || num_accumulated == 0) return hipSuccess;             // nothing accumulated: pixels stay as they are
    hipLaunchKernelGGL(pack_pixels_kernel, dim3((uint32_t)((n_pixels + 255u) / 256u)), dim3(256), 0, stream, accumulator, pixels, n_pixels, (float)num_accumulated);
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
