// hdr.hpp -- Radiance RGBE (.hdr) reader for Scene::loadHDRI.
//
// The reference calls stbi_loadf(path, &w, &h, &n, 4) with stbi_set_flip_vertically_on_load(true)
// (scene/scene.cuh:964-970; stb_image is a third-party header the reference vendors).  This is an
// independent reader of the same public file format that yields the same floats: header lines up to
// an empty line (must include FORMAT=32-bit_rle_rgbe), the resolution line "-Y <h> +X <w>", then
// scanlines that are either flat RGBE quadruples or the "new" run-length form (2, 2, width_hi,
// width_lo, then each of the four channels run-length coded: count > 128 = a run of count-128
// copies of the next byte, else that many literal bytes).  A pixel is (r, g, b) * 2^(e - 136), or
// zero when e == 0; alpha is 1; rows are returned bottom-up (the vertical flip).
#pragma once
#include <cmath>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace ptrt_detail {

inline void rgbe_to_rgba(const unsigned char *p, float *out) {
    if (p[3] != 0) {
        const float f = std::ldexp(1.0f, (int)p[3] - (128 + 8));
        out[0] = p[0] * f;
        out[1] = p[1] * f;
        out[2] = p[2] * f;
    } else {
        out[0] = out[1] = out[2] = 0.0f;
    }
    out[3] = 1.0f;
}

// rgba: width*height*4 floats, first row = bottom row of the picture
inline void load_radiance_hdr(const std::string &path, int &width, int &height, std::vector<float> &rgba) {
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw std::runtime_error("Failed to load HDRI file: " + path);
    std::string line;
    std::getline(in, line);
    if (line != "#?RADIANCE" && line != "#?RGBE")
        throw std::runtime_error("Failed to load HDRI file: " + path + " (not a Radiance picture)");
    bool format_ok = false;
    while (std::getline(in, line) && !line.empty())
        if (line == "FORMAT=32-bit_rle_rgbe")
            format_ok = true;
    if (!format_ok)
        throw std::runtime_error("Failed to load HDRI file: " + path + " (unsupported FORMAT)");
    std::getline(in, line);
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &height, &width) != 2 || width < 1 || height < 1)
        throw std::runtime_error("Failed to load HDRI file: " + path + " (unsupported data layout)");
    std::vector<unsigned char> data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    std::vector<unsigned char> row((size_t)width * 4);
    rgba.assign((size_t)width * height * 4, 0.0f);
    size_t pos = 0;
    auto need = [&](size_t n) {
        if (pos + n > data.size())
            throw std::runtime_error("Failed to load HDRI file: " + path + " (truncated)");
    };
    const bool maybe_rle = width >= 8 && width < 32768;
    for (int y = 0; y < height; ++y) {
        bool rle = false;
        if (maybe_rle && pos + 4 <= data.size() && data[pos] == 2 && data[pos + 1] == 2 && !(data[pos + 2] & 0x80) &&
            ((data[pos + 2] << 8) | data[pos + 3]) == width)
            rle = true;
        if (rle) {
            pos += 4;
            for (int ch = 0; ch < 4; ++ch)
                for (int x = 0; x < width;) {
                    need(1);
                    int count = data[pos++];
                    if (count > 128) {
                        count -= 128;
                        need(1);
                        const unsigned char v = data[pos++];
                        if (count == 0 || x + count > width)
                            throw std::runtime_error("Failed to load HDRI file: " + path + " (corrupt run)");
                        for (int k = 0; k < count; ++k)
                            row[(size_t)(x++) * 4 + ch] = v;
                    } else {
                        if (count == 0 || x + count > width)
                            throw std::runtime_error("Failed to load HDRI file: " + path + " (corrupt run)");
                        need((size_t)count);
                        for (int k = 0; k < count; ++k)
                            row[(size_t)(x++) * 4 + ch] = data[pos++];
                    }
                }
        } else {
            need((size_t)width * 4);
            for (size_t k = 0; k < (size_t)width * 4; ++k)
                row[k] = data[pos++];
        }
        float *dst = rgba.data() + (size_t)(height - 1 - y) * width * 4; // flip_vertically_on_load(true)
        for (int x = 0; x < width; ++x)
            rgbe_to_rgba(&row[(size_t)x * 4], dst + (size_t)x * 4);
    }
}

} // namespace ptrt_detail
