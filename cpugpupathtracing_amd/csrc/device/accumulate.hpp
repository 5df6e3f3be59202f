// accumulate.hpp -- the accumulate + pack step shared by the wavefront pipeline and the persistent kernel (gfx950):
// adds the finished paths' radiance of one batch to the float4 accumulator IN SAMPLE ORDER (the reference adds one sample per
// Render(), ref: Source/Main.cpp:735-746; float addition is not associative, so the order is part of the result) and packs
// data.pixels (ref: Include/MathLib.h:144-152).
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.h"
#include "rt_device.hpp"
#include "shade_device.hpp"
#include "trace_steps.hpp"

namespace cgpt {
namespace dev {

static constexpr uint32_t kAccChunk = 8;                 // samples staged per pixel and pass in the pixel-major path

// All 256 threads of the block call it (ends in a block-level reduction).  One thread = one pixel, grid-stride over the band.
// st_en[path id] = {radiance.xyz, bits(final ray depth)}.  With pixel-major path ids a pixel's samples are one contiguous run: read
// directly, the 64 lanes of a wave would touch 64 different lines per load (measured: 9.3 ms instead of 1.3 per 256-spp frame),
// so the wave loads 64 pixels x kAccChunk samples in memory order (eight full 128-byte lines per load instruction) into LDS and
// every lane then reads its own pixel's samples from there, in order.
__device__ __forceinline__ void accumulate_batch(const DevRenderArgs& args, const float4* __restrict__ st_en, const PathGrid& g,
                                                 uint32_t batch_first, uint32_t batch_n)
{
    __shared__ float4 stage[4][64 * (kAccChunk + 1u)];                         // per wave: [pixel][sample], one float4 of padding per pixel
    const DevSettings& st = args.settings;
    const uint32_t lane = threadIdx.x & 63u;
    float4* const my_stage = stage[threadIdx.x >> 6];
    double energy_sum = 0.0;
    for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < g.n_pixels; p += gridDim.x * 256u) {     // grid-stride: a bounded number of blocks
        uint32_t px = 0, py = 0, local_row = 0;
        const bool is_pixel = pixel_of_index(args, g, p, px, py, local_row);
        const size_t local_index = (size_t)local_row * args.width + px;
        const bool brute = st.render_mode == 1u || (st.render_mode == 0u && px < args.width / 2u);   // TracePath has no ray-depth view (ref: Main.cpp:581-689)
        float4 acc = is_pixel ? args.accumulator[local_index] : float4{ 0.0f, 0.0f, 0.0f, 0.0f };
        V3 last = mk(0.0f);
        auto add_sample = [&](float4 e4) {
            PathState ps;
            ps.energy = mk(e4.x, e4.y, e4.z);
            ps.depth = __float_as_uint(e4.w) & 0xFFu;
            const V3 e = brute ? ps.energy : final_energy(st, ps);
            energy_sum += (double)(e.x + e.y + e.z) * 0.001;                  // ref: Main.cpp:735
            if (st.debug_mode == 0u) { acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += 1.0f; }
            else last = e;
        };
        if (g.order == kPixelMajor) {
            const uint32_t p_base = p - lane;                                 // the wave's 64 pixels: one 8x8 tile
            for (uint32_t s0 = 0; s0 < batch_n; s0 += kAccChunk) {
                const uint32_t cnt = min(kAccChunk, batch_n - s0);
#pragma unroll
                for (uint32_t k = 0; k < kAccChunk; ++k) {
                    const uint32_t e = k * 64u + lane, pix = e / kAccChunk, smp = e % kAccChunk;
                    if (smp < cnt) my_stage[pix * (kAccChunk + 1u) + smp] = st_en[(size_t)(p_base + pix) * g.n_samples + s0 + smp];
                }
                __builtin_amdgcn_wave_barrier();
                if (is_pixel)
                    for (uint32_t j = 0; j < cnt; ++j) add_sample(my_stage[lane * (kAccChunk + 1u) + j]);
                __builtin_amdgcn_wave_barrier();
            }
        } else if (is_pixel) {
            for (uint32_t s = 0; s < batch_n; ++s) add_sample(ld_stream(&st_en[path_id(g, s, p)]));
        }
        if (is_pixel) {
            if (st.debug_mode == 0u) {
                args.accumulator[local_index] = acc;
                const float n = (float)(batch_first + batch_n);               // data.num_accumulated after this batch
                args.pixels[local_index] = vec4_to_uint(acc.x / n, acc.y / n, acc.z / n);
            } else {
                args.pixels[local_index] = vec4_to_uint(last.x, last.y, last.z);
            }
        }
    }
    block_add_f64(&args.counters->total_energy, energy_sum);
}

}  // namespace dev
}  // namespace cgpt
