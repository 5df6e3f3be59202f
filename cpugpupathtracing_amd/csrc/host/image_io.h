// image_io.h -- host framebuffer dump: the headless replacement for the DX12 presenter
// (ref: Source/DX12.cpp:277-322 CopyToBackBuffer/Present).
#pragma once
#include <cstdint>
#include <string>

namespace cgpt {
// RGBA8 pixels as packed by Vec4ToUint (ref: Include/MathLib.h:144-152: R | G<<8 | B<<16 | 0xFF<<24) -> binary PPM (P6)
bool WritePPM(const std::string& path, const uint32_t* pixels, uint32_t width, uint32_t height, std::string& error);
// float4 accumulator / num_accumulated -> PFM (PF, little-endian, bottom-up rows per the format)
bool WritePFM(const std::string& path, const float* accumulator_rgba, uint32_t num_accumulated, uint32_t width, uint32_t height, std::string& error);
// raw float4 accumulator + header (resume point: SURVEY 8f-3): "CGPTACC1" u32 W, H, num_accumulated, then W*H*4 floats
bool WriteAccumulator(const std::string& path, const float* accumulator_rgba, uint32_t num_accumulated, uint32_t width, uint32_t height, std::string& error);
bool ReadAccumulator(const std::string& path, float* accumulator_rgba, uint32_t* num_accumulated, uint32_t width, uint32_t height, std::string& error);
}  // namespace cgpt
