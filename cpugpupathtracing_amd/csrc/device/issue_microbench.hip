// issue_microbench.hip -- measures the roof the trace kernel is priced against: vector-instruction issue on gfx950.
//
// wf_trace is branchy FP32 VALU work whose BVH is cache resident, so its ceiling is how many wave64 vector
// instructions the chip issues per second, not HBM bandwidth.  That ceiling is measured here rather than assumed: every wave
// runs a long stream of independent instructions of one kind (8 independent chains, so the ~4-cycle dependent latency never
// gates issue) at a chosen number of waves per SIMD, all 256 CUs busy.  bench.py runs it in the same process as the render and
// prints the measured rate as roofline.peak.
//   kind 0  v_mul_f32           the plain full-rate VALU instruction (every add / mul / cmp / cndmask of the kernels)
//   kind 1  v_pk_mul_f32        the packed-f32 form of the slab test (two multiplies per lane per instruction)
//   kind 2  v_pk_add_f32
//   kind 3  v_rcp_f32           quarter-rate transcendental (the 1/x of the triangle and ray set-up)
//   kind 4  v_mul_f32 x 3 + v_pk_mul_f32 x 1, interleaved (48 : 16 per 64)
//   kind 5  the same 48 : 16, the packed instructions grouped (16 v_pk_mul_f32, then 48 v_mul_f32)
//   kind 6  v_mul_f32 and v_pk_mul_f32 alternating 1 : 1
//   kind 7  v_cndmask_b32 (VCC select)          kind 8  v_mul_lo_u32 (the PCG / index multiply)
//   kind 9  v_cndmask_b32_e64 (SGPR-pair select) kind 10 v_cmp_lt_f32 + v_cndmask_b32 pairs   kind 11 v_add_u32   kind 12 v_min3_f32
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "cpugpupt_abi.h"

namespace cgpt {

hipStream_t CtxStream(cgpt_ctx* ctx);
int CtxDevice(cgpt_ctx* ctx);
int CtxFail(cgpt_ctx* ctx, int code, const char* fmt, ...);
cgpt_ctx* GroupFirstMemberOrNull(cgpt_ctx* ctx);
int GroupForwarded(cgpt_ctx* ctx, int rc);

typedef float mb_f2 __attribute__((ext_vector_type(2)));

static constexpr uint32_t kInstsPerIter = 64;

template <int KIND>
__global__ void __launch_bounds__(256) issue_stream(uint32_t iters, float seed, float* sink)
{
    float a0 = seed + (float)threadIdx.x, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    mb_f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, p4 = { a1, a0 }, p5 = { a3, a2 }, p6 = { a5, a4 }, p7 = { a7, a6 };
    const float c = 1.0000001f;
    const mb_f2 pc = { c, c };
    for (uint32_t i = 0; i < iters; ++i) {
        if (KIND == 0) {
#define MB8(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
            asm volatile(MB8("v_mul_f32") MB8("v_mul_f32") MB8("v_mul_f32") MB8("v_mul_f32") MB8("v_mul_f32") MB8("v_mul_f32") MB8("v_mul_f32") MB8("v_mul_f32")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (KIND == 1) {
            asm volatile(MB8("v_pk_mul_f32") MB8("v_pk_mul_f32") MB8("v_pk_mul_f32") MB8("v_pk_mul_f32") MB8("v_pk_mul_f32") MB8("v_pk_mul_f32") MB8("v_pk_mul_f32") MB8("v_pk_mul_f32")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
        } else if (KIND == 2) {
            asm volatile(MB8("v_pk_add_f32") MB8("v_pk_add_f32") MB8("v_pk_add_f32") MB8("v_pk_add_f32") MB8("v_pk_add_f32") MB8("v_pk_add_f32") MB8("v_pk_add_f32") MB8("v_pk_add_f32")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
#undef MB8
        } else if (KIND == 3) {
#define MB8R "v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\nv_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7\n"
            asm volatile(MB8R MB8R MB8R MB8R MB8R MB8R MB8R MB8R
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#undef MB8R
        } else if (KIND == 5) {
#define MBG_PK "v_pk_mul_f32 %8, %8, %13\nv_pk_mul_f32 %9, %9, %13\nv_pk_mul_f32 %10, %10, %13\nv_pk_mul_f32 %11, %11, %13\n"
#define MBG_S "v_mul_f32 %0, %0, %12\nv_mul_f32 %1, %1, %12\nv_mul_f32 %2, %2, %12\nv_mul_f32 %3, %3, %12\nv_mul_f32 %4, %4, %12\nv_mul_f32 %5, %5, %12\nv_mul_f32 %6, %6, %12\nv_mul_f32 %7, %7, %12\n"
            asm volatile(MBG_PK MBG_PK MBG_PK MBG_PK MBG_S MBG_S MBG_S MBG_S MBG_S MBG_S
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                         : "v"(c), "v"(pc));
#undef MBG_PK
#undef MBG_S
        } else if (KIND == 6) {
#define MBA "v_mul_f32 %0, %0, %12\nv_pk_mul_f32 %8, %8, %13\nv_mul_f32 %1, %1, %12\nv_pk_mul_f32 %9, %9, %13\nv_mul_f32 %2, %2, %12\nv_pk_mul_f32 %10, %10, %13\nv_mul_f32 %3, %3, %12\nv_pk_mul_f32 %11, %11, %13\n" \
            "v_mul_f32 %4, %4, %12\nv_pk_mul_f32 %8, %8, %13\nv_mul_f32 %5, %5, %12\nv_pk_mul_f32 %9, %9, %13\nv_mul_f32 %6, %6, %12\nv_pk_mul_f32 %10, %10, %13\nv_mul_f32 %7, %7, %12\nv_pk_mul_f32 %11, %11, %13\n"
            asm volatile(MBA MBA MBA MBA
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                         : "v"(c), "v"(pc));
#undef MBA
        } else if (KIND == 7) {
#define MB8C "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
            asm volatile(MB8C MB8C MB8C MB8C MB8C MB8C MB8C MB8C
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
#undef MB8C
        } else if (KIND == 9) {
#define MB8E "v_cndmask_b32_e64 %0, %0, %8, s[10:11]\nv_cndmask_b32_e64 %1, %1, %8, s[10:11]\nv_cndmask_b32_e64 %2, %2, %8, s[10:11]\nv_cndmask_b32_e64 %3, %3, %8, s[10:11]\nv_cndmask_b32_e64 %4, %4, %8, s[10:11]\nv_cndmask_b32_e64 %5, %5, %8, s[10:11]\nv_cndmask_b32_e64 %6, %6, %8, s[10:11]\nv_cndmask_b32_e64 %7, %7, %8, s[10:11]\n"
            asm volatile(MB8E MB8E MB8E MB8E MB8E MB8E MB8E MB8E
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "s10", "s11");
#undef MB8E
        } else if (KIND == 10) {
            // compare + select pairs: the shape of the branch-free traversal steps
#define MB8P "v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_lt_f32 vcc, %2, %8\nv_cndmask_b32 %3, %3, %8, vcc\nv_cmp_lt_f32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %8, vcc\nv_cmp_lt_f32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %8, vcc\n"
            asm volatile(MB8P MB8P MB8P MB8P MB8P MB8P MB8P MB8P
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
#undef MB8P
        } else if (KIND == 11) {
#define MB8A "v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
            asm volatile(MB8A MB8A MB8A MB8A MB8A MB8A MB8A MB8A
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
#undef MB8A
        } else if (KIND == 12) {
#define MB83 "v_min3_f32 %0, %0, %8, %1\nv_min3_f32 %1, %1, %8, %2\nv_min3_f32 %2, %2, %8, %3\nv_min3_f32 %3, %3, %8, %4\nv_min3_f32 %4, %4, %8, %5\nv_min3_f32 %5, %5, %8, %6\nv_min3_f32 %6, %6, %8, %7\nv_min3_f32 %7, %7, %8, %0\n"
            asm volatile(MB83 MB83 MB83 MB83 MB83 MB83 MB83 MB83
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
#undef MB83
        } else if (KIND == 8) {
#define MB8M "v_mul_lo_u32 %0, %0, %8\nv_mul_lo_u32 %1, %1, %8\nv_mul_lo_u32 %2, %2, %8\nv_mul_lo_u32 %3, %3, %8\nv_mul_lo_u32 %4, %4, %8\nv_mul_lo_u32 %5, %5, %8\nv_mul_lo_u32 %6, %6, %8\nv_mul_lo_u32 %7, %7, %8\n"
            asm volatile(MB8M MB8M MB8M MB8M MB8M MB8M MB8M MB8M
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
#undef MB8M
        } else {
            // 48 scalar-form + 16 packed instructions per iteration
#define MBMIX "v_mul_f32 %0, %0, %12\nv_mul_f32 %1, %1, %12\nv_mul_f32 %2, %2, %12\nv_pk_mul_f32 %8, %8, %13\n" \
              "v_mul_f32 %3, %3, %12\nv_mul_f32 %4, %4, %12\nv_mul_f32 %5, %5, %12\nv_pk_mul_f32 %9, %9, %13\n" \
              "v_mul_f32 %6, %6, %12\nv_mul_f32 %7, %7, %12\nv_mul_f32 %0, %0, %12\nv_pk_mul_f32 %10, %10, %13\n" \
              "v_mul_f32 %1, %1, %12\nv_mul_f32 %2, %2, %12\nv_mul_f32 %3, %3, %12\nv_pk_mul_f32 %11, %11, %13\n"
            asm volatile(MBMIX MBMIX MBMIX MBMIX
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                         : "v"(c), "v"(pc));
#undef MBMIX
        }
    }
    const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x;
    if (r == 12345.678f) *sink = r;                                           // keeps the chains alive; never true in practice
}

}  // namespace cgpt

using namespace cgpt;

static int MeasureOnDevice(cgpt_ctx* ctx, uint32_t kind, uint32_t waves_per_simd, uint32_t iters, double* wave_insts_per_sec, double* ms_out);

extern "C" int cgpt_measure_issue_rate(cgpt_ctx* ctx, uint32_t kind, uint32_t waves_per_simd, uint32_t iters, double* wave_insts_per_sec, double* ms_out)
{
    if (!ctx) return CGPT_ERR_INVALID;
    if (cgpt_ctx* first = GroupFirstMemberOrNull(ctx))                        // a multi-device context measures its first device
        return GroupForwarded(ctx, MeasureOnDevice(first, kind, waves_per_simd, iters, wave_insts_per_sec, ms_out));
    return MeasureOnDevice(ctx, kind, waves_per_simd, iters, wave_insts_per_sec, ms_out);
}

static int MeasureOnDevice(cgpt_ctx* ctx, uint32_t kind, uint32_t waves_per_simd, uint32_t iters, double* wave_insts_per_sec, double* ms_out)
{
    if (!wave_insts_per_sec || kind > 12u || waves_per_simd == 0u || waves_per_simd > 8u || iters == 0u || iters > (1u << 24))
        return CtxFail(ctx, CGPT_ERR_INVALID, "cgpt_measure_issue_rate: kind <= 12, 1 <= waves_per_simd <= 8, 1 <= iters <= 2^24");
#define MB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return CtxFail(ctx, CGPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
    MB_TRY(hipSetDevice(CtxDevice(ctx)));
    int cus = 0;
    MB_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, CtxDevice(ctx)));
    float* sink = nullptr;
    MB_TRY(hipMalloc((void**)&sink, sizeof(float)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    MB_TRY(hipEventCreate(&e0)); MB_TRY(hipEventCreate(&e1));
    hipStream_t st = CtxStream(ctx);
    // one 256-thread block = one wave on each of a CU's 4 SIMDs; waves_per_simd blocks per CU, all resident at once
    const dim3 grid((uint32_t)cus * waves_per_simd), block(256);
    auto launch = [&](uint32_t n) {
        switch (kind) {
        case 0: hipLaunchKernelGGL(issue_stream<0>, grid, block, 0, st, n, 1.0f, sink); break;
        case 1: hipLaunchKernelGGL(issue_stream<1>, grid, block, 0, st, n, 1.0f, sink); break;
        case 2: hipLaunchKernelGGL(issue_stream<2>, grid, block, 0, st, n, 1.0f, sink); break;
        case 3: hipLaunchKernelGGL(issue_stream<3>, grid, block, 0, st, n, 1.0f, sink); break;
        case 5: hipLaunchKernelGGL(issue_stream<5>, grid, block, 0, st, n, 1.0f, sink); break;
        case 6: hipLaunchKernelGGL(issue_stream<6>, grid, block, 0, st, n, 1.0f, sink); break;
        case 7: hipLaunchKernelGGL(issue_stream<7>, grid, block, 0, st, n, 1.0f, sink); break;
        case 8: hipLaunchKernelGGL(issue_stream<8>, grid, block, 0, st, n, 1.0f, sink); break;
        case 9: hipLaunchKernelGGL(issue_stream<9>, grid, block, 0, st, n, 1.0f, sink); break;
        case 10: hipLaunchKernelGGL(issue_stream<10>, grid, block, 0, st, n, 1.0f, sink); break;
        case 11: hipLaunchKernelGGL(issue_stream<11>, grid, block, 0, st, n, 1.0f, sink); break;
        case 12: hipLaunchKernelGGL(issue_stream<12>, grid, block, 0, st, n, 1.0f, sink); break;
        default: hipLaunchKernelGGL(issue_stream<4>, grid, block, 0, st, n, 1.0f, sink); break;
        }
    };
    launch(iters);                                                            // warm-up: code object load, clocks
    float ms = 0.0f;
    for (int rep = 0; rep < 3; ++rep) {                                       // fastest of three (the clock settles during the first)
        MB_TRY(hipEventRecord(e0, st));
        launch(iters);
        MB_TRY(hipEventRecord(e1, st));
        MB_TRY(hipEventSynchronize(e1));
        MB_TRY(hipGetLastError());
        float t = 0.0f;
        MB_TRY(hipEventElapsedTime(&t, e0, e1));
        if (rep == 0 || t < ms) ms = t;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
#undef MB_TRY
    const double insts = (double)grid.x * 4.0 * (double)iters * (double)kInstsPerIter;
    *wave_insts_per_sec = insts / ((double)ms * 1e-3);
    if (ms_out) *ms_out = ms;
    return CGPT_OK;
}
