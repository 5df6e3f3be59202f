// image_io.cpp -- see image_io.h.
#include "image_io.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace cgpt {

bool WritePPM(const std::string& path, const uint32_t* pixels, uint32_t width, uint32_t height, std::string& error)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { error = "cannot write " + path; return false; }
    fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::vector<uint8_t> row(3 * (size_t)width);
    for (uint32_t y = 0; y < height; ++y) {
        for (uint32_t x = 0; x < width; ++x) {
            const uint32_t p = pixels[(size_t)y * width + x];
            row[3 * x] = (uint8_t)(p & 0xFF); row[3 * x + 1] = (uint8_t)((p >> 8) & 0xFF); row[3 * x + 2] = (uint8_t)((p >> 16) & 0xFF);
        }
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return true;
}

bool WritePFM(const std::string& path, const float* acc, uint32_t n, uint32_t width, uint32_t height, std::string& error)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { error = "cannot write " + path; return false; }
    fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
    const float inv = n ? 1.0f / (float)n : 0.0f;
    std::vector<float> row(3 * (size_t)width);
    for (uint32_t yy = 0; yy < height; ++yy) {
        const uint32_t y = height - 1 - yy;
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) row[3 * x + c] = acc[4 * ((size_t)y * width + x) + c] * inv;
        fwrite(row.data(), sizeof(float), row.size(), f);
    }
    fclose(f);
    return true;
}

bool WriteAccumulator(const std::string& path, const float* acc, uint32_t n, uint32_t width, uint32_t height, std::string& error)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { error = "cannot write " + path; return false; }
    const uint32_t hdr[3] = { width, height, n };
    fwrite("CGPTACC1", 1, 8, f);
    fwrite(hdr, 4, 3, f);
    fwrite(acc, sizeof(float), 4 * (size_t)width * height, f);
    fclose(f);
    return true;
}

bool ReadAccumulator(const std::string& path, float* acc, uint32_t* n, uint32_t width, uint32_t height, std::string& error)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { error = "cannot read " + path; return false; }
    char magic[8]; uint32_t hdr[3];
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "CGPTACC1", 8) == 0 && fread(hdr, 4, 3, f) == 3;
    if (ok && (hdr[0] != width || hdr[1] != height)) { error = "accumulator file has a different size"; ok = false; }
    else if (!ok) error = "not an accumulator file: " + path;
    if (ok) {
        const size_t count = 4 * (size_t)width * height;
        ok = fread(acc, sizeof(float), count, f) == count;
        if (!ok) error = "truncated accumulator file: " + path;
        else *n = hdr[2];
    }
    fclose(f);
    return ok;
}

}  // namespace cgpt
