// render.h -- the consumer of a scene: what main.cu:167-207 does with `render<<<...>>>`, the 8-bit
// conversion and the image file, on top of the C ABI.
#ifndef PT_HOST_RENDER_H
#define PT_HOST_RENDER_H
#include <cstdio>
#include <string>
#include <vector>

#include "camera.h"
#include "hittable_list.h"

namespace pt_host {

// Stands in for `render<<<blocks, threads>>>(fb, W, H, spp, background, cam->d_this, world->d_this, rand)`
// (main.cu:184): fills fb[j*W+i] with the gamma-corrected mean colour, exactly what the reference's
// frame buffer holds after the kernel (main.cu:95-99).
inline void render(vec3 *fb, int max_x, int max_y, int samples_per_pixel, color background, camera **cam, hittable **world,
                   pt_stats *stats = nullptr)
{
    if (!fb || !cam || !world) throw scene_error("render(): null frame buffer, camera or world handle");
    std::vector<float> gamma((size_t)max_x * max_y * 3);
    const float bg[3] = {background.x(), background.y(), background.z()};
    const float *sum = Session::get().render_sum(world, cam, bg, max_x, max_y, samples_per_pixel, stats);
    if (pt_film_finalize(sum, max_x, max_y, (uint32_t)samples_per_pixel, gamma.data(), nullptr) != PT_OK)
        throw scene_error("pt_film_finalize failed");
    for (size_t p = 0; p < (size_t)max_x * max_y; ++p) fb[p] = vec3(gamma[3 * p], gamma[3 * p + 1], gamma[3 * p + 2]);
}

// main.cu:192-205: 8-bit pixels, rows top-down from j = H-1
inline std::vector<uint8_t> quantise(const vec3 *fb, int width, int height)
{
    std::vector<uint8_t> px((size_t)width * height * 3);
    size_t w = 0;
    for (int j = height - 1; j >= 0; --j)
        for (int i = 0; i < width; ++i) {
            const vec3 &c = fb[(size_t)j * width + i];
            px[w++] = (uint8_t)static_cast<int>(256 * clamp(c.r(), 0.0f, 0.999f));
            px[w++] = (uint8_t)static_cast<int>(256 * clamp(c.g(), 0.0f, 0.999f));
            px[w++] = (uint8_t)static_cast<int>(256 * clamp(c.b(), 0.0f, 0.999f));
        }
    return px;
}

// 24-bit uncompressed BMP from top-down RGB rows (the job of stbi_write_bmp at main.cu:207).
inline bool write_bmp(const std::string &path, int width, int height, const uint8_t *rgb_top_down)
{
    const uint32_t row = ((uint32_t)width * 3 + 3) & ~3u, data = row * (uint32_t)height, off = 54;
    std::vector<uint8_t> out(off + data, 0);
    auto u32 = [&](size_t at, uint32_t v) { for (int k = 0; k < 4; ++k) out[at + k] = (uint8_t)(v >> (8 * k)); };
    out[0] = 'B'; out[1] = 'M';
    u32(2, off + data); u32(10, off); u32(14, 40); u32(18, (uint32_t)width); u32(22, (uint32_t)height);
    out[26] = 1; out[28] = 24; u32(34, data);
    for (int y = 0; y < height; ++y) { // BMP stores rows bottom-up, pixels as BGR
        const uint8_t *src = rgb_top_down + (size_t)(height - 1 - y) * width * 3;
        uint8_t *dst = out.data() + off + (size_t)y * row;
        for (int x = 0; x < width; ++x) { dst[3 * x] = src[3 * x + 2]; dst[3 * x + 1] = src[3 * x + 1]; dst[3 * x + 2] = src[3 * x]; }
    }
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return fclose(f) == 0 && ok;
}

// Portable float map (little-endian, bottom-up rows): the unquantised frame buffer, for tests.
inline bool write_pfm(const std::string &path, int width, int height, const vec3 *fb_bottom_up)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
    bool ok = true;
    for (size_t p = 0; p < (size_t)width * height && ok; ++p) ok = fwrite(fb_bottom_up[p].e, 4, 3, f) == 3;
    return fclose(f) == 0 && ok;
}

} // namespace pt_host
#endif
