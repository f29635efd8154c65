// image_out.cpp - tone map + PNG output, the reference's output path (main.cpp:78-131).
//
// Host-only post-process after the framebuffer gather; not accelerated.  LogAverageLuma sums
// logf(0.01 + luma) sequentially in row-major order in float (the order matters for byte-equal
// output), the per-pixel operator is Reinhard l/(1+l) with key 0.18, and Color_Pack truncates
// (u8)(clamp01(c) * 255) (color.h:93-111).  The PNG encoder is our own (8-bit, zlib deflate via the
// system libz) behind stb_image_write's call signature stbi_write_png(name, w, h, comp, data, stride)
// (main.cpp:129); pixels decode identically, the compressed byte stream is not stb's.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <zlib.h>

#include "prt_scene.h"

static inline float LumaOf(Vector4 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }   // color.h:93-96

static float LogAverageLuma(const Framebuffer * fb) {                 // main.cpp:78-99
    float acc = 0.0f;
    for (u32 y = 0; y < fb->height; ++y) {
        for (u32 x = 0; x < fb->width; ++x) {
            float l = LumaOf(fb->pixels[y * fb->width + x]);
            if (l > 0.0f) acc += logf(0.01f + l);
            else printf("Non-positive luma at (%u, %u): %f\n", x, y, l);
        }
    }
    return expf(acc / (float)(fb->width * fb->height));
}

static inline u8 PackChannel(float v) { return (u8)(PrtClamp(v, 0.0f, 1.0f) * 255.0f); }          // color.h:104-111

float TonemapFramebuffer(const Framebuffer * fb, u8 * rgba8) {       // main.cpp:107-127
    float scene_luma = LogAverageLuma(fb);
    for (u32 y = 0; y < fb->height; ++y) {
        for (u32 x = 0; x < fb->width; ++x) {
            u32 idx = y * fb->width + x;
            Vector4 c = fb->pixels[idx];
            float key_alpha = 0.18f;
            float pixel_luma = LumaOf(c);
            float l_xy = key_alpha * pixel_luma / scene_luma;
            float l_d = l_xy / (1.0f + l_xy);
            float scale = l_d / pixel_luma;
            c.x *= scale;
            c.y *= scale;
            c.z *= scale;
            rgba8[idx * 4 + 0] = PackChannel(c.x);
            rgba8[idx * 4 + 1] = PackChannel(c.y);
            rgba8[idx * 4 + 2] = PackChannel(c.z);
            rgba8[idx * 4 + 3] = PackChannel(c.w);
        }
    }
    return scene_luma;
}

static void PutChunk(FILE * fp, const char tag[4], const u8 * data, u32 len) {
    u8 hdr[8] = { (u8)(len >> 24), (u8)(len >> 16), (u8)(len >> 8), (u8)len, (u8)tag[0], (u8)tag[1], (u8)tag[2], (u8)tag[3] };
    fwrite(hdr, 1, 8, fp);
    if (len) fwrite(data, 1, len, fp);
    uLong crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32(crc, data, len);
    u8 tail[4] = { (u8)(crc >> 24), (u8)(crc >> 16), (u8)(crc >> 8), (u8)crc };
    fwrite(tail, 1, 4, fp);
}

extern "C" int stbi_write_png(char const * filename, int w, int h, int comp, const void * data, int stride_in_bytes) {
    if (w <= 0 || h <= 0 || comp < 1 || comp > 4 || !data) return 0;
    static const u8 colour_type[5] = { 0, 0, 4, 2, 6 };
    size_t row = (size_t)w * (size_t)comp;
    size_t stride = stride_in_bytes ? (size_t)stride_in_bytes : row;
    std::vector<u8> raw((row + 1) * (size_t)h);
    for (int y = 0; y < h; ++y) {
        raw[(row + 1) * (size_t)y] = 0;                               // filter type: none
        memcpy(&raw[(row + 1) * (size_t)y + 1], (const u8 *)data + stride * (size_t)y, row);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<u8> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return 0;

    FILE * fp = fopen(filename, "wb");
    if (!fp) return 0;
    static const u8 sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    fwrite(sig, 1, 8, fp);
    u8 ihdr[13] = { (u8)(w >> 24), (u8)(w >> 16), (u8)(w >> 8), (u8)w, (u8)(h >> 24), (u8)(h >> 16), (u8)(h >> 8), (u8)h,
                    8, colour_type[comp], 0, 0, 0 };
    PutChunk(fp, "IHDR", ihdr, 13);
    PutChunk(fp, "IDAT", z.data(), (u32)zlen);
    PutChunk(fp, "IEND", NULL, 0);
    fclose(fp);
    return 1;
}

void WriteFramebufferImage(Framebuffer * fb, const char * filename) {  // main.cpp:101-131
    std::vector<u8> buffer((size_t)fb->width * fb->height * 4);
    float scene_luma = TonemapFramebuffer(fb, buffer.data());
    printf("scene_luma = %f\n", scene_luma);
    stbi_write_png(filename, (int)fb->width, (int)fb->height, 4, buffer.data(), 0);
}
