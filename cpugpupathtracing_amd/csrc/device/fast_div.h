// fast_div.h -- exact unsigned division by a launch constant (Granlund-Montgomery round-up method), host + device.
// n / d == fast_div(n, MakeFastDiv(d)) for every 32-bit n and every d >= 1: 4 VALU instead of the ~25 of an emulated 32-bit
// division.  d == 1 is the one divisor the multiplier form cannot express (it would need mul = 2^32); it is flagged by
// mul == 0, which no other divisor produces, and returns n.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define CGPT_HD __host__ __device__
#else
#define CGPT_HD
#endif

namespace cgpt {

struct FastDiv { uint32_t mul, shift; };

CGPT_HD inline uint32_t fast_div(uint32_t n, FastDiv d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(d.mul, n);
#else
    const uint32_t t = (uint32_t)(((uint64_t)d.mul * n) >> 32);
#endif
    const uint32_t q = (t + ((n - t) >> 1)) >> d.shift;
    return d.mul == 0u ? n : q;                                               // wave-uniform select
}

inline FastDiv MakeFastDiv(uint32_t d)                                       // d >= 1
{
    FastDiv f;
    if (d <= 1u) { f.mul = 0u; f.shift = 0u; return f; }                      // identity (see above)
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;                                              // l = ceil(log2 d) >= 1
    f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1ull);
    f.shift = l - 1u;
    return f;
}

}  // namespace cgpt
