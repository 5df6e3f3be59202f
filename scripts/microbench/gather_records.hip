// gather_records.hip -- microbenchmark (gfx950): how fast can the lanes of a wave chase pointers through 64-byte records?
// The BVH inner step of the trace kernels fetches one 64-byte child-pair record per lane and step (56 useful bytes as
// 3 x dwordx4 + 1 x dwordx2) and the next record depends on what was loaded.  This measures lane-records per second for
// different ways of issuing that fetch, at several table sizes (L2 / Infinity Cache resident), waves per SIMD and active-lane
// fractions, to tell which unit the gather is bound by (per-access tag lookups vs returned dwords vs latency).
//   mode 0: per lane 3 x dwordx4 + dwordx2        (what load_pair() does)
//   mode 1: per lane 4 x dwordx4
//   mode 2: per lane 1 x dwordx4 + dwordx2 codes   (two accesses per record: a floor for per-lane fetches)
//   mode 3: quad-cooperative: the four lanes of a quad read the four 16-byte pieces of ONE record with one dwordx4 each,
//           four such loads cover the quad's four records (one 64-byte access per record instead of four)
//   mode 4: per lane dwordx4 x 3 + dwordx2 through a single-use "sc1" / nontemporal hint variant (cache policy check)
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench/gather_records.hip -o scripts/microbench/gather_records
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t quad_bcast(uint32_t v, int k)
{
    switch (k) {
    case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xf, 0xf, true);   // [1,1,1,1]
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xaa, 0xf, 0xf, true);   // [2,2,2,2]
    default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xff, 0xf, 0xf, true);  // [3,3,3,3]
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) chase(const float4* __restrict__ recs, uint32_t n_recs, uint32_t steps, uint32_t active_pct, float* sink)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (tid * 2654435761u) % n_recs;
    float acc = 0.0f;
    const bool active = ((lane * 37u + 11u) % 100u) < active_pct;
    const char* base = reinterpret_cast<const char*>(recs);
    if (MODE == 3) {
        // all lanes of a quad take part in the loads (inactive lanes still lend their load slot: their own record is simply not fetched)
        for (uint32_t s = 0; s < steps; ++s) {
            const uint32_t piece = (lane & 3u) << 4;
            f4v p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t rk = quad_bcast(idx, k);
                const uint32_t ak = quad_bcast(active ? 1u : 0u, k);
                p[k] = f4v{ 0, 0, 0, 0 };
                if (ak) p[k] = *reinterpret_cast<const f4v*>(base + ((size_t)rk << 6) + piece);
            }
            // the codes of record k sit in quad lane 3's p[k].zw; hand record k's pair to quad lane k
            uint32_t lc = 0, rc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t l = quad_bcast(__float_as_uint(p[k].z), 3), r = quad_bcast(__float_as_uint(p[k].w), 3);
                if ((int)(lane & 3u) == k) { lc = l; rc = r; }
            }
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += (p[k].x + p[k].y) + (p[k].z + p[k].w);
            acc += sum;
            if (active) idx = (__float_as_uint(sum) & 64u) ? lc : rc;
        }
    } else if (active) {
        for (uint32_t s = 0; s < steps; ++s) {
            const char* rec = base + ((size_t)idx << 6);
            f4v q0 = *reinterpret_cast<const f4v*>(rec);
            f4v q1 = f4v{ 0, 0, 0, 0 }, q2 = f4v{ 0, 0, 0, 0 };
            uint32_t lc, rc;
            if (MODE == 0 || MODE == 1 || MODE == 5) { q1 = *reinterpret_cast<const f4v*>(rec + 16); q2 = *reinterpret_cast<const f4v*>(rec + 32); }
            if (MODE == 4) q1 = *reinterpret_cast<const f4v*>(rec + 16);
            if (MODE == 1) { const f4v q3 = *reinterpret_cast<const f4v*>(rec + 48); lc = __float_as_uint(q3.z); rc = __float_as_uint(q3.w); acc += q3.x; }
            else if (MODE == 4) { lc = __float_as_uint(q1.z) % n_recs; rc = __float_as_uint(q1.w) % n_recs; if (lc > idx + 2048u || lc < idx) lc = (idx + 1u + (lc & 63u)) % n_recs; if (rc > idx + 4096u || rc < idx) rc = (idx + 64u + (rc & 1023u)) % n_recs; }
            else if (MODE == 5) { lc = __float_as_uint(q2.z) % n_recs; rc = __float_as_uint(q2.w) % n_recs; if (lc > idx + 2048u || lc < idx) lc = (idx + 1u + (lc & 63u)) % n_recs; if (rc > idx + 4096u || rc < idx) rc = (idx + 64u + (rc & 1023u)) % n_recs; }
            else if (MODE == 6) {                                             // two-phase: 32 bytes first, the other 24 only for about half of the lanes (a dependent fetch)
                q1 = *reinterpret_cast<const f4v*>(rec + 16);
                const float part = (q0.x + q0.y) + (q1.z + q1.w);
                if (__float_as_uint(part) & 32u) { q2 = *reinterpret_cast<const f4v*>(rec + 32); const u2v c = *reinterpret_cast<const u2v*>(rec + 56); lc = c.x; rc = c.y; }
                else { lc = (idx + 1u + (__float_as_uint(part) & 63u)) % n_recs; rc = (idx + 64u + (__float_as_uint(part) & 1023u)) % n_recs; }
            }
            else { const u2v c = *reinterpret_cast<const u2v*>(rec + 56); lc = c.x; rc = c.y; }
            const float sum = ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w)) + ((q2.x + q2.y) + (q2.z + q2.w));
            acc += sum;
            idx = (__float_as_uint(sum) & 64u) ? lc : rc;
        }
    }
    if (acc == 123.456f) *sink = acc + (float)idx;
}

int main(int argc, char** argv)
{
    const uint32_t steps = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000;
    int dev = 0, cus = 0;
    CHECK(hipSetDevice(dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    float* sink; CHECK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const uint32_t sizes[] = { 81920u, 1310720u };
    printf("# lane-records per second (G/s) over the chip; %d CUs, %u dependent steps per lane\n", cus, steps);
    printf("# mode: 0 = 3x b128 + b64 per lane | 1 = 4x b128 per lane | 2 = b128 + b64 per lane | 3 = quad-cooperative 4x b128 | 4 = 2x b128 | 5 = 3x b128 | 6 = 2x b128, then b128 + b64 for ~half of the lanes\n");
    for (uint32_t n : sizes) {
        std::vector<float4> h((size_t)n * 4);
        uint32_t s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
        for (uint32_t r = 0; r < n; ++r) {
            for (int j = 0; j < 4; ++j) {
                float4 v; v.x = (float)(rnd() >> 8) * 1e-7f; v.y = (float)(rnd() >> 8) * 1e-7f; v.z = (float)(rnd() >> 8) * 1e-7f; v.w = (float)(rnd() >> 8) * 1e-7f;
                h[(size_t)r * 4 + j] = v;
            }
            uint32_t lc = rnd() % n, rc = rnd() % n;
            // locality like a tree's: most children are near their parent
            if ((rnd() & 3u) != 0u) { lc = (r + 1 + (rnd() & 63u)) % n; rc = (r + 64 + (rnd() & 1023u)) % n; }
            h[(size_t)r * 4 + 3].z = __builtin_bit_cast(float, lc); h[(size_t)r * 4 + 3].w = __builtin_bit_cast(float, rc);
        }
        float4* d; CHECK(hipMalloc((void**)&d, h.size() * sizeof(float4)));
        CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
        for (uint32_t pct : { 100u, 55u }) {
            for (uint32_t wps : { 2u, 5u, 8u }) {
                printf("records %8u (%5.1f MB) active %3u%% waves/SIMD %u :", n, n * 64.0 / 1e6, pct, wps);
                for (int mode = 0; mode < 7; ++mode) {
                    const dim3 grid((uint32_t)cus * wps), block(256);
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; ++rep) {
                        CHECK(hipEventRecord(e0, 0));
                        switch (mode) {
                        case 0: hipLaunchKernelGGL(chase<0>, grid, block, 0, 0, d, n, steps, pct, sink); break;
                        case 1: hipLaunchKernelGGL(chase<1>, grid, block, 0, 0, d, n, steps, pct, sink); break;
                        case 2: hipLaunchKernelGGL(chase<2>, grid, block, 0, 0, d, n, steps, pct, sink); break;
                        case 3: hipLaunchKernelGGL(chase<3>, grid, block, 0, 0, d, n, steps, pct, sink); break;
                        case 4: hipLaunchKernelGGL(chase<4>, grid, block, 0, 0, d, n, steps, pct, sink); break;
                        case 5: hipLaunchKernelGGL(chase<5>, grid, block, 0, 0, d, n, steps, pct, sink); break;
                        default: hipLaunchKernelGGL(chase<6>, grid, block, 0, 0, d, n, steps, pct, sink); break;
                        }
                        CHECK(hipEventRecord(e1, 0));
                        CHECK(hipEventSynchronize(e1));
                        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                        if (ms < best) best = ms;
                    }
                    // active lanes: count exactly
                    uint32_t act = 0; for (uint32_t l = 0; l < 64; ++l) act += ((l * 37u + 11u) % 100u) < pct;
                    const double recs = (double)grid.x * 4.0 * act * steps;
                    printf("  m%d %7.2f", mode, recs / (best * 1e-3) / 1e9);
                }
                printf("\n");
                fflush(stdout);
            }
        }
        CHECK(hipFree(d));
    }
    return 0;
}
