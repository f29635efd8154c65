// dev_texture.h - texture lookups of the shading path (texture.cpp:5-83, raytracer.cpp:439-502).
//
// Layout: every texture is expanded on the host to RGBA8 with GetTexel's channel rules (texture.cpp:27-41:
// missing alpha = 255, one-channel maps are grey, two-channel maps are (r, g, 0)), so a texel is ONE 4-byte load.
// The per-texel sRGB -> linear powf of the reference (texture.cpp:44-48) only ever sees 256 different inputs
// (u8 * (1/255)): the host evaluates them once with its own powf into srgb_lut and the device does table
// lookups, which is exact.  Everything else - wrap, v flip, the (size - 2) scale, clamp, floor, the order of the
// three lerps - is restated expression by expression; lerp stays a + (b - a) * t (mathlib.h:10), unfused.
#pragma once

#include "dev_scene.h"

namespace prt {

PRT_D float tex_wrap(float uv) {                               // texture.cpp:5-13; fmodf(x, 1) == x - trunc(x), exactly
    const float frac = uv - truncf(uv);
    return uv >= 0.0f ? frac : 1.0f + frac;
}

struct TexTap {                                                // the 2x2 footprint of one lookup
    const unsigned int * px;                                   // texture's first texel
    unsigned int i00, i01, i10, i11;                           // texel indices: (x0,y0) (x0,y1) (x1,y0) (x1,y1)
    float fx, fy;
};

PRT_D TexTap tex_tap(const DevScene & sc, unsigned int tex, float u, float v) {
    const DevTexture t = sc.textures[tex];
    u = tex_wrap(u);
    v = 1.0f - tex_wrap(v);
    const float sx = (float)(t.size_x - 2u), sy = (float)(t.size_y - 2u);
    const float tx = ref_min(ref_max(u * sx, 0.0f), sx);       // Clamp = Min(Max(n, a), b), mathlib.h:7-9
    const float ty = ref_min(ref_max(v * sy, 0.0f), sy);
    const unsigned int tx0 = (unsigned int)floorf(tx), ty0 = (unsigned int)floorf(ty);
    TexTap k;
    k.px = sc.texels + t.first_texel;
    k.i00 = ty0 * t.size_x + tx0;
    k.i01 = k.i00 + t.size_x;
    k.i10 = k.i00 + 1u;
    k.i11 = k.i01 + 1u;
    k.fx = tx - (float)tx0;
    k.fy = ty - (float)ty0;
    return k;
}

PRT_D float tex_lerp(float a, float b, float t) { return a + (b - a) * t; }

// One channel (0 = r .. 3 = a) of Texture_SampleBilinear: Lerp(Lerp(s00, s01, fy), Lerp(s10, s11, fy), fx).
PRT_D float tex_channel(const DevScene & sc, const TexTap & k, unsigned int t00, unsigned int t01, unsigned int t10, unsigned int t11,
                        int shift) {
    const float s00 = sc.srgb_lut[(t00 >> shift) & 255u], s01 = sc.srgb_lut[(t01 >> shift) & 255u];
    const float s10 = sc.srgb_lut[(t10 >> shift) & 255u], s11 = sc.srgb_lut[(t11 >> shift) & 255u];
    return tex_lerp(tex_lerp(s00, s01, k.fy), tex_lerp(s10, s11, k.fy), k.fx);
}

PRT_D f3 tex_sample_rgb(const DevScene & sc, unsigned int tex, float u, float v) {
    const TexTap k = tex_tap(sc, tex, u, v);
    const unsigned int t00 = k.px[k.i00], t01 = k.px[k.i01], t10 = k.px[k.i10], t11 = k.px[k.i11];
    return mk3(tex_channel(sc, k, t00, t01, t10, t11, 0), tex_channel(sc, k, t00, t01, t10, t11, 8),
               tex_channel(sc, k, t00, t01, t10, t11, 16));
}

PRT_D float tex_sample_r(const DevScene & sc, unsigned int tex, float u, float v) {
    const TexTap k = tex_tap(sc, tex, u, v);
    return tex_channel(sc, k, k.px[k.i00], k.px[k.i01], k.px[k.i10], k.px[k.i11], 0);
}

// What the textures of a hit change (raytracer.cpp:439-502): colours, alpha, the shading normal.
struct TexturedHit {
    f3 ambient, diffuse, specular;
    float alpha;
};

}  // namespace prt
