// vec.h -- host-side float3 helpers for the scene/BVH mirror.
// Operation order follows the reference's MathLib (ref: Include/MathLib.h:57-102) because BVH split
// decisions and bounds must be bit-identical to the reference build.  Compiled with
// -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>

namespace cgpt {

struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    Vec3() = default;
    explicit Vec3(float s) : x(s), y(s), z(s) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](uint32_t axis) const { return axis == 0 ? x : (axis == 1 ? y : z); }
};

inline Vec3 operator+(const Vec3& a, const Vec3& b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Vec3 operator*(const Vec3& a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline Vec3 operator*(float s, const Vec3& a) { return { s * a.x, s * a.y, s * a.z }; }
inline float dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(const Vec3& a) { return sqrtf(dot(a, a)); }
// std::min / std::max comparison direction (ref: MathLib.h:95-96)
inline float min_std(float a, float b) { return (b < a) ? b : a; }
inline float max_std(float a, float b) { return (a < b) ? b : a; }
inline Vec3 vmin(const Vec3& a, const Vec3& b) { return { min_std(a.x, b.x), min_std(a.y, b.y), min_std(a.z, b.z) }; }
inline Vec3 vmax(const Vec3& a, const Vec3& b) { return { max_std(a.x, b.x), max_std(a.y, b.y), max_std(a.z, b.z) }; }

static constexpr float kPi = 3.14159265f;  // ref: MathLib.h:5

}  // namespace cgpt
