// image_in.cpp - texture image decoders behind LoadTexture (obj_parser.cpp:197-213).
//
// The reference decodes textures with a third-party single-header library (lib/stb_image.h, not part of this
// repository).  This file is a fresh decoder for the formats scenes of this kind ship with; what it must
// reproduce is that library's OUTPUT CONVENTION for `stbi_load(name, &x, &y, &channels, 0)`, because the hot
// path indexes the bytes directly (texture.cpp:17-51):
//   * rows top to bottom, pixels left to right, `channels` interleaved bytes per pixel, channels = what the file
//     stores (1 grey, 2 grey+alpha, 3 RGB, 4 RGBA); palettes are expanded to RGB (RGBA with a tRNS chunk);
//   * PNG  1 / 2 / 4 / 8 / 16-bit samples (below 8 bits scaled to 0..255, 16-bit: high byte kept), all five scanline
//          filters, Adam7 interlacing, palettes with tRNS, colour-key tRNS on grey / RGB (with the reference decoder's
//          channel-count quirk, see DecodePng);
//   * TGA  types 1 / 2 / 3 (raw) and 9 / 10 / 11 (run-length): colour maps (8- / 16-bit indices; 15 / 16 / 24 / 32-bit
//          entries), 15 / 16-bit 5-5-5, 16-bit grey + alpha, 8 / 24 / 32 bits, BGR(A) -> RGB(A), bottom-up files flipped;
//   * BMP  4- / 8-bit palettes, 16-bit (5-5-5 or masks), 24-bit, 32-bit (plain or masks), OS/2 and V4 / V5 headers,
//          BGR(A) -> RGB(A), bottom-up files flipped; channels and alpha handling as in the reference's decoder;
//   * PNM  binary P5 (grey) / P6 (RGB), maxval <= 255;
//   * GIF  the first image on the logical screen, always RGBA (background and transparent pixels: alpha 0), interlaced
//          or not, global or local colour table;
//   * PSD  the flattened RGB composite, 8 / 16 bits, raw or PackBits, always RGBA, colours un-blended from white;
//   * HDR  Radiance RGBE (flat or run-length scanlines), tone-mapped to 8 bits with the reference decoder's gamma 2.2;
//   * PIC  Softimage: chained channel packets, raw / pure / mixed run-length, 3 or 4 channels.
//   * JPEG baseline and progressive (image_jpeg.cpp): grey -> 1 channel, colour -> 3, every sampling layout, restart intervals.
// Anything else (arithmetic-coded JPEG, run-length / 1-bit BMP ...) is reported and the texture slot
// stays empty, which is how the reference treats a file its decoder rejects (obj_parser.cpp:201-204).
// tests/test_host_side.py compares the decoded bytes with the reference's on generated files of every kind.
#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "prt_scene.h"

namespace prt_jpeg {
const char * Decode(const std::vector<u8> & file, u32 * w, u32 * h, u32 * channels, std::vector<u8> * px);   // image_jpeg.cpp
}

namespace {

std::string gImageError;

bool Fail(const char * why) {
    gImageError = why;
    return false;
}

struct Bytes {
    std::vector<u8> data;
    bool Read(const char * path) {
        FILE * f = fopen(path, "rb");
        if (!f) return Fail("cannot open file");
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (n < 0) { fclose(f); return Fail("cannot size file"); }
        data.resize((size_t)n);
        size_t got = n ? fread(data.data(), 1, (size_t)n, f) : 0;
        fclose(f);
        return got == (size_t)n ? true : Fail("short read");
    }
};

u32 Be32(const u8 * p) { return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3]; }
u32 Le16(const u8 * p) { return (u32)p[0] | ((u32)p[1] << 8); }
u32 Le32(const u8 * p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

struct Image {
    u32 w = 0, h = 0, channels = 0;
    std::vector<u8> px;
};

// ---- PNG ------------------------------------------------------------------------------------------------
int Paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

bool DecodePng(const std::vector<u8> & d, Image * out) {
    static const u8 sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (d.size() < 8 || memcmp(d.data(), sig, 8) != 0) return Fail("not a PNG");
    u32 w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<u8> idat, palette, trns;
    bool have_header = false;
    size_t pos = 8;
    while (pos + 12 <= d.size()) {
        u32 len = Be32(&d[pos]);
        const u8 * tag = &d[pos + 4];
        const u8 * body = &d[pos + 8];
        if (pos + 12 + (size_t)len > d.size()) return Fail("PNG chunk runs past the end of the file");
        if (!memcmp(tag, "IHDR", 4)) {
            if (len != 13) return Fail("bad IHDR");
            w = Be32(body); h = Be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            have_header = true;
        } else if (!memcmp(tag, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!memcmp(tag, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(tag, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_header || !w || !h) return Fail("PNG without a header");
    if ((unsigned long long)w * h > (1ull << 28)) return Fail("PNG larger than 2^28 pixels");
    if (interlace > 1) return Fail("bad PNG interlace method");
    u32 file_ch = 0;
    switch (ctype) {
        case 0: file_ch = 1; break;
        case 2: file_ch = 3; break;
        case 3: file_ch = 1; break;
        case 4: file_ch = 2; break;
        case 6: file_ch = 4; break;
        default: return Fail("bad PNG colour type");
    }
    const bool low = depth == 1 || depth == 2 || depth == 4;
    if (!(depth == 8 || (depth == 16 && ctype != 3) || (low && (ctype == 0 || ctype == 3)))) return Fail("bad PNG bit depth for the colour type");
    const bool colour_key = ctype != 3 && !trns.empty();         // tRNS on a grey / RGB image names ONE transparent colour
    if (colour_key && (ctype == 4 || ctype == 6)) return Fail("PNG tRNS chunk on an image with alpha");
    if (colour_key && trns.size() != 2u * file_ch) return Fail("bad PNG tRNS length");
    const size_t bps = depth == 16 ? 2 : 1;                    // bytes per sample once unpacked
    const size_t upp = bps * file_ch;                          // unpacked bytes per pixel
    const size_t bits = (size_t)depth * file_ch;               // bits per pixel in the filtered stream
    const size_t fbpp = bits >= 8 ? bits / 8 : 1;              // "bytes per pixel" of the scanline filters
    // the seven Adam7 passes (x0, y0, dx, dy), or the whole image as one pass
    static const u32 adam7[7][4] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    static const u32 whole[1][4] = { { 0, 0, 1, 1 } };
    const u32 (*passes)[4] = interlace ? adam7 : whole;
    const int n_pass = interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int k = 0; k < n_pass; ++k) {
        const u32 pw = (w - passes[k][0] + passes[k][2] - 1) / passes[k][2], ph = (h - passes[k][1] + passes[k][3] - 1) / passes[k][3];
        if (w > passes[k][0] && h > passes[k][1] && pw && ph) raw_size += (((size_t)pw * bits + 7) / 8 + 1) * (size_t)ph;
    }
    // deflate expands at most ~1032:1: a header that promises more than the data can hold is corrupt (and would
    // otherwise make us allocate whatever it says)
    if (raw_size > idat.size() * 1032 + 1024) return Fail("PNG data too short for the image size");
    std::vector<u8> raw(raw_size);
    uLongf raw_len = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || raw_len != raw.size()) return Fail("PNG data does not inflate to the image size");
    std::vector<u8> img(upp * (size_t)w * h);                  // unpacked: one byte per sample (two for 16-bit), raw values
    std::vector<u8> line, prev;
    size_t rp = 0;
    for (int k = 0; k < n_pass; ++k) {
        if (!(w > passes[k][0] && h > passes[k][1])) continue;
        const u32 pw = (w - passes[k][0] + passes[k][2] - 1) / passes[k][2], ph = (h - passes[k][1] + passes[k][3] - 1) / passes[k][3];
        if (!pw || !ph) continue;
        const size_t stride = ((size_t)pw * bits + 7) / 8;
        line.assign(stride, 0);
        prev.assign(stride, 0);
        for (u32 y = 0; y < ph; ++y) {
            const u8 filter = raw[rp++];
            const u8 * src = &raw[rp];
            rp += stride;
            for (size_t i = 0; i < stride; ++i) {
                int fa = i >= fbpp ? line[i - fbpp] : 0;
                int fb = y ? prev[i] : 0;
                int fc = (y && i >= fbpp) ? prev[i - fbpp] : 0;
                int v = src[i];
                switch (filter) {
                    case 0: break;
                    case 1: v += fa; break;
                    case 2: v += fb; break;
                    case 3: v += (fa + fb) >> 1; break;
                    case 4: v += Paeth(fa, fb, fc); break;
                    default: return Fail("bad PNG filter");
                }
                line[i] = (u8)v;
            }
            // unpack the scanline into its pixels of the full image
            const size_t oy = (size_t)passes[k][1] + (size_t)y * passes[k][3];
            for (u32 x = 0; x < pw; ++x) {
                u8 * o = &img[((size_t)oy * w + passes[k][0] + (size_t)x * passes[k][2]) * upp];
                if (low) {
                    const size_t bit = (size_t)x * depth;
                    o[0] = (u8)((line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u));
                } else {
                    memcpy(o, &line[(size_t)x * upp], upp);
                }
            }
            prev.swap(line);
        }
    }
    out->w = w;
    out->h = h;
    const size_t n_px = (size_t)w * h;
    if (ctype == 3) {
        const bool with_alpha = !trns.empty();
        out->channels = with_alpha ? 4 : 3;
        out->px.resize(n_px * out->channels);
        for (size_t i = 0; i < n_px; ++i) {
            u32 idx = img[i];
            u8 * o = &out->px[i * out->channels];
            for (int k = 0; k < 3; ++k) o[k] = (size_t)idx * 3 + k < palette.size() ? palette[(size_t)idx * 3 + k] : 0;
            if (with_alpha) o[3] = idx < trns.size() ? trns[idx] : 255;
        }
    } else {
        // grey / RGB samples below 8 bits are scaled to 0..255 (x 255, x 85, x 17); 16-bit samples keep their high byte; a
        // colour key (compared at the file's own precision) adds an alpha channel that is 0 on the key colour, 255 elsewhere
        const u32 scale = depth == 1 ? 255u : depth == 2 ? 85u : depth == 4 ? 17u : 1u;
        out->channels = file_ch + (colour_key ? 1 : 0);
        out->px.resize(n_px * out->channels);
        for (size_t i = 0; i < n_px; ++i) {
            const u8 * in = &img[i * upp];
            u8 * o = &out->px[i * out->channels];
            bool is_key = colour_key;
            for (u32 k = 0; k < file_ch; ++k) {
                if (bps == 2) {
                    o[k] = in[2 * k];
                    if (colour_key && !(in[2 * k] == trns[2 * k] && in[2 * k + 1] == trns[2 * k + 1])) is_key = false;
                } else {
                    o[k] = (u8)(in[k] * scale);
                    if (colour_key && o[k] != (u8)(trns[2 * k + 1] * scale)) is_key = false;
                }
            }
            if (colour_key) o[file_ch] = is_key ? 0 : 255;
        }
        if (colour_key) {
            // The reference's decoder (stb_image 2.14, stbi__do_png) hands back this (channels + 1)-interleaved buffer but
            // reports the FILE's channel count, and the reference indexes texels with what is reported (texture.cpp:17-51):
            // it sees a w x h x channels texture made of the first w * h * channels bytes.  Same here - the picture is
            // scrambled, but it is the reference's picture.
            out->channels = file_ch;
            out->px.resize(n_px * file_ch);
        }
    }
    return true;
}

// ---- TGA ------------------------------------------------------------------------------------------------
// The reference decoder's conventions: the channel count comes from the palette entry size of a colour-mapped file, else
// from the pixel size - 8 bits: 1 (grey), 16 bits of a grey image: 2 (grey + alpha), 15 / 16 bits: 3 (5-5-5, each channel
// c * 255 / 31, no alpha), 24 / 32 bits: 3 / 4 (BGR(A) -> RGB(A)); run-length packets may cross rows; an index past the
// palette reads entry 0; the palette is found after the image id plus "first entry" BYTES.
void TgaRgb16(const u8 * p, u8 * o) {
    const u32 px = Le16(p);
    o[0] = (u8)((((px >> 10) & 31u) * 255u) / 31u);
    o[1] = (u8)((((px >> 5) & 31u) * 255u) / 31u);
    o[2] = (u8)(((px & 31u) * 255u) / 31u);
}

bool DecodeTga(const std::vector<u8> & d, Image * out) {
    if (d.size() < 18) return Fail("TGA header truncated");
    const u32 id_len = d[0], cmap_type = d[1];
    u32 type = d[2];
    const u32 pal_start = Le16(&d[3]), pal_len = Le16(&d[5]), pal_bits = d[7];
    const u32 w = Le16(&d[12]), h = Le16(&d[14]), bits = d[16], desc = d[17];
    if (cmap_type > 1) return Fail("bad TGA colour map type");
    const bool rle = type >= 8;
    if (rle) type -= 8;
    const bool indexed = cmap_type == 1;
    if (indexed ? type != 1 : (type != 2 && type != 3)) return Fail("unsupported TGA image type");
    if (!w || !h) return Fail("empty TGA");
    if (indexed && bits != 8 && bits != 16) return Fail("unsupported TGA index size");
    const u32 fmt_bits = indexed ? pal_bits : bits;
    u32 ch = 0;
    bool rgb16 = false;
    switch (fmt_bits) {
        case 8: ch = 1; break;
        case 16: if (!indexed && type == 3) { ch = 2; break; }   // grey + alpha; else 5-5-5
        // fall through
        case 15: ch = 3; rgb16 = true; break;
        case 24: ch = 3; break;
        case 32: ch = 4; break;
        default: return Fail("unsupported TGA pixel depth");
    }
    const u32 in_bytes = indexed ? bits / 8 : (rgb16 ? 2 : ch);   // bytes per pixel in the file
    size_t pos = 18 + (size_t)id_len;
    std::vector<u8> palette;
    if (indexed) {
        const size_t entry = rgb16 ? 2 : ch;
        pos += pal_start;
        if (pos > d.size() || (size_t)pal_len * entry > d.size() - pos) return Fail("TGA palette truncated");
        palette.resize((size_t)pal_len * ch);
        for (u32 i = 0; i < pal_len; ++i) {
            if (rgb16) TgaRgb16(&d[pos + (size_t)i * 2], &palette[(size_t)i * 3]);
            else memcpy(&palette[(size_t)i * ch], &d[pos + (size_t)i * ch], ch);
        }
        pos += (size_t)pal_len * entry;
        if (!pal_len) return Fail("TGA palette is empty");
    }
    const size_t n_px = (size_t)w * h;
    // a raw image needs n_px * in_bytes bytes of data, a run-length one at least one packet per 128 pixels
    if (pos > d.size() || (rle ? (n_px + 127) / 128 * (1 + in_bytes) : n_px * in_bytes) > d.size() - pos) return Fail("TGA pixel data truncated");
    std::vector<u8> px(n_px * ch);
    auto pixel = [&](const u8 * p, u8 * o) {                     // one file pixel -> ch output bytes (still BGR order for 24 / 32 bits)
        if (indexed) {
            u32 idx = bits == 8 ? p[0] : Le16(p);
            if (idx >= pal_len) idx = 0;
            memcpy(o, &palette[(size_t)idx * ch], ch);
        } else if (rgb16) {
            TgaRgb16(p, o);
        } else {
            memcpy(o, p, ch);
        }
    };
    if (!rle) {
        for (size_t i = 0; i < n_px; ++i) pixel(&d[pos + i * in_bytes], &px[i * ch]);
    } else {
        size_t i = 0;
        while (i < n_px) {
            if (pos >= d.size()) return Fail("TGA run-length data truncated");
            const u32 head = d[pos++];
            size_t count = (head & 127u) + 1u;
            if (count > n_px - i) count = n_px - i;               // the last packet may promise more pixels than are left
            if (head & 128u) {
                if (in_bytes > d.size() - pos) return Fail("TGA run-length data truncated");
                u8 one[4];
                pixel(&d[pos], one);
                for (size_t k = 0; k < count; ++k) memcpy(&px[(i + k) * ch], one, ch);
                pos += in_bytes;
            } else {
                if (count * in_bytes > d.size() - pos) return Fail("TGA run-length data truncated");
                for (size_t k = 0; k < count; ++k) pixel(&d[pos + k * in_bytes], &px[(i + k) * ch]);
                pos += count * in_bytes;
            }
            i += count;
        }
    }
    if (ch >= 3 && !rgb16) for (size_t i = 0; i < n_px; ++i) std::swap(px[i * ch], px[i * ch + 2]);        // BGR(A) -> RGB(A)
    const bool top_down = (desc & 0x20u) != 0;
    out->w = w; out->h = h; out->channels = ch;
    out->px.resize(px.size());
    const size_t stride = (size_t)w * ch;
    for (u32 y = 0; y < h; ++y) memcpy(&out->px[stride * y], &px[stride * (top_down ? y : h - 1 - y)], stride);
    return true;
}

// ---- BMP ------------------------------------------------------------------------------------------------
// What the reference's decoder accepts, with its conventions: header sizes 12 (OS/2) / 40 / 56 / 108 / 124; 4- and 8-bit
// palettes, 16-bit (5-5-5, or BITFIELDS masks), 24-bit, 32-bit; no RLE, no 1-bit.  Channels: 4 when the format has an
// alpha mask - a plain 32-bit file counts as having one, and if every alpha byte of it is 0 they all become 255 - else
// 3.  Masked channels of fewer than 8 bits are widened by repeating their top bits.  A positive height is bottom-up.
int BmpHighBit(u32 z) {
    int n = 0;
    if (z == 0) return -1;
    if (z >= 0x10000) { n += 16; z >>= 16; }
    if (z >= 0x00100) { n += 8; z >>= 8; }
    if (z >= 0x00010) { n += 4; z >>= 4; }
    if (z >= 0x00004) { n += 2; z >>= 2; }
    if (z >= 0x00002) { n += 1; }
    return n;
}

int BmpBitCount(u32 a) {
    int n = 0;
    for (; a; a >>= 1) n += (int)(a & 1u);
    return n;
}

// the masked bits moved so that their top bit is bit 7, then repeated downwards until 8 bits are filled (32-bit
// two's-complement arithmetic: a mask that reaches bit 31 shifts in sign bits, of which only the low byte is kept)
u8 BmpChannel(u32 masked, int shift, int bits) {
    int32_t v = (int32_t)masked;
    if (shift < 0) v = (int32_t)((u32)v << (-shift)); else v >>= shift;
    int32_t result = v;
    for (int z = bits; z < 8 && bits > 0; z += bits) result += v >> z;
    return (u8)(result & 255);
}

bool DecodeBmp(const std::vector<u8> & d, Image * out) {
    if (d.size() < 26 || d[0] != 'B' || d[1] != 'M') return Fail("not a BMP");
    const u32 offset = Le32(&d[10]), hsz = Le32(&d[14]);
    if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return Fail("unknown BMP header size");
    if (d.size() < 14 + (size_t)hsz) return Fail("BMP header truncated");
    int32_t w, hs;
    u32 planes, bpp, compress = 0;
    u32 mr = 0, mg = 0, mb = 0, ma = 0;
    bool plain32 = false;
    if (hsz == 12) {
        w = (int32_t)Le16(&d[18]); hs = (int32_t)Le16(&d[20]); planes = Le16(&d[22]); bpp = Le16(&d[24]);
    } else {
        w = (int32_t)Le32(&d[18]); hs = (int32_t)Le32(&d[22]); planes = Le16(&d[26]); bpp = Le16(&d[28]);
        compress = Le32(&d[30]);
    }
    if (planes != 1) return Fail("bad BMP");
    if (bpp == 1) return Fail("1-bit BMP is not supported");
    if (compress == 1 || compress == 2) return Fail("run-length BMP is not supported");
    if (hsz == 40 || hsz == 56) {
        if (bpp == 16 || bpp == 32) {
            if (compress == 0) {
                if (bpp == 32) { mr = 0xFFu << 16; mg = 0xFFu << 8; mb = 0xFFu; ma = 0xFFu << 24; plain32 = true; }
                else { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
            } else if (compress == 3) {
                const size_t at = 14 + (size_t)hsz;                // the three masks follow the header
                if (d.size() < at + 12) return Fail("BMP masks truncated");
                mr = Le32(&d[at]); mg = Le32(&d[at + 4]); mb = Le32(&d[at + 8]);
                if (mr == mg && mg == mb) return Fail("bad BMP masks");
            } else {
                return Fail("bad BMP compression");
            }
        }
    } else if (hsz == 108 || hsz == 124) {
        mr = Le32(&d[54]); mg = Le32(&d[58]); mb = Le32(&d[62]); ma = Le32(&d[66]);
    }
    if (w <= 0 || hs == 0 || hs == INT32_MIN) return Fail("empty BMP");
    const bool bottom_up = hs > 0;
    const u32 h = (u32)(hs < 0 ? -hs : hs);
    if ((unsigned long long)w * h > (1ull << 28)) return Fail("BMP larger than 2^28 pixels");
    const u32 channels = ma ? 4 : 3;
    std::vector<u8> px;
    size_t row_bytes;
    u8 pal[256][3];
    if (bpp < 16) {
        if (bpp != 4 && bpp != 8) return Fail("bad BMP bit depth");
        const long entry = hsz == 12 ? 3 : 4;
        const long psize = hsz == 12 ? ((long)offset - 14 - 24) / 3 : ((long)offset - 14 - (long)hsz) >> 2;
        if (psize <= 0 || psize > 256) return Fail("bad BMP palette size");
        if (d.size() < 14 + (size_t)hsz + (size_t)psize * entry) return Fail("BMP palette truncated");
        memset(pal, 0, sizeof(pal));
        for (long i = 0; i < psize; ++i) {
            const u8 * e = &d[14 + (size_t)hsz + (size_t)i * entry];
            pal[i][0] = e[2]; pal[i][1] = e[1]; pal[i][2] = e[0];
        }
        const size_t width = bpp == 4 ? ((size_t)w + 1) >> 1 : (size_t)w;
        row_bytes = (width + 3) & ~(size_t)3;
    } else if (bpp == 16) {
        row_bytes = ((size_t)w * 2 + 3) & ~(size_t)3;
    } else if (bpp == 24) {
        row_bytes = ((size_t)w * 3 + 3) & ~(size_t)3;
    } else if (bpp == 32) {
        row_bytes = (size_t)w * 4;
    } else {
        return Fail("bad BMP bit depth");
    }
    // With a 40- or 56-byte header and BITFIELDS masks the reference's decoder (stb_image 2.14) has already consumed the
    // twelve mask bytes when it skips "offset - 14 - header size" bytes to the pixels: it starts twelve bytes late and
    // reads zeros past the end of the file.  Same here, so that such a file gives the reference's (shifted) picture.
    const size_t late = ((hsz == 40 || hsz == 56) && compress == 3 && (bpp == 16 || bpp == 32)) ? 12 : 0;
    std::vector<u8> shifted;
    const u8 * base = d.data();
    size_t avail = d.size();
    if (late) {
        if ((size_t)offset > d.size() || row_bytes * h > d.size() - offset) return Fail("BMP pixel data truncated");
        shifted.assign(d.begin(), d.end());
        shifted.resize(d.size() + late, 0);
        base = shifted.data();
        avail = shifted.size();
    }
    if ((size_t)offset + late > avail || row_bytes * h > avail - offset - late) return Fail("BMP pixel data truncated");
    const bool easy24 = bpp == 24;
    const bool easy32 = bpp == 32 && mb == 0xFFu && mg == 0xFF00u && mr == 0x00FF0000u && ma == 0xFF000000u;
    int rshift = 0, gshift = 0, bshift = 0, ashift = 0, rcount = 0, gcount = 0, bcount = 0, acount = 0;
    if (bpp >= 16 && !easy24 && !easy32) {
        if (!mr || !mg || !mb) return Fail("bad BMP masks");
        rshift = BmpHighBit(mr) - 7; rcount = BmpBitCount(mr);
        gshift = BmpHighBit(mg) - 7; gcount = BmpBitCount(mg);
        bshift = BmpHighBit(mb) - 7; bcount = BmpBitCount(mb);
        ashift = BmpHighBit(ma) - 7; acount = BmpBitCount(ma);
    }
    px.resize((size_t)w * h * channels);
    u32 all_a = plain32 ? 0u : 255u;
    for (u32 y = 0; y < h; ++y) {
        const u8 * src = base + (size_t)offset + late + row_bytes * (bottom_up ? h - 1 - y : y);
        u8 * dst = &px[(size_t)w * channels * y];
        for (int32_t x = 0; x < w; ++x) {
            u8 * o = dst + (size_t)x * channels;
            u32 a = 255;
            if (bpp < 16) {
                const u32 v = bpp == 8 ? src[x] : (x & 1 ? src[x >> 1] & 15u : src[x >> 1] >> 4);
                o[0] = pal[v][0]; o[1] = pal[v][1]; o[2] = pal[v][2];
            } else if (easy24 || easy32) {
                const u8 * e = src + (size_t)x * (bpp / 8);
                o[0] = e[2]; o[1] = e[1]; o[2] = e[0];
                if (easy32) a = e[3];
            } else {
                const u32 v = bpp == 16 ? Le16(src + 2 * (size_t)x) : Le32(src + 4 * (size_t)x);
                o[0] = BmpChannel(v & mr, rshift, rcount);
                o[1] = BmpChannel(v & mg, gshift, gcount);
                o[2] = BmpChannel(v & mb, bshift, bcount);
                if (ma) {                                      // the library keeps this one as a full int for its all-zero test
                    int32_t t = (int32_t)(v & ma);
                    if (ashift < 0) t = (int32_t)((u32)t << (-ashift)); else t >>= ashift;
                    int32_t r = t;
                    for (int z = acount; z < 8 && acount > 0; z += acount) r += t >> z;
                    a = (u32)r;
                }
            }
            all_a |= a;
            if (channels == 4) o[3] = (u8)(a & 255u);
        }
    }
    if (channels == 4 && all_a == 0)                           // an alpha channel that is 0 everywhere was never meant as one
        for (size_t i = 3; i < px.size(); i += 4) px[i] = 255;
    out->w = (u32)w; out->h = h; out->channels = channels;
    out->px.swap(px);
    return true;
}

// ---- GIF ------------------------------------------------------------------------------------------------
// The reference decoder's reading of a GIF: the FIRST image of the file on the logical screen, always 4 channels.  The
// screen starts as the background colour (global palette entry `bgindex`; black without a global palette) with alpha 0;
// the image's pixels overwrite it with alpha 255, except pixels of the transparent index of a preceding graphic control
// extension, which leave the background (alpha 0) in place.  Interlaced images are de-interlaced.
bool DecodeGif(const std::vector<u8> & d, Image * out) {
    if (d.size() < 13 || memcmp(d.data(), "GIF8", 4) != 0 || (d[4] != '7' && d[4] != '9') || d[5] != 'a') return Fail("not a GIF");
    const u32 W = Le16(&d[6]), H = Le16(&d[8]), flags = d[10], bgindex = d[11];
    if (!W || !H) return Fail("empty GIF");
    if ((unsigned long long)W * H > (1ull << 28)) return Fail("GIF larger than 2^28 pixels");
    size_t pos = 13;
    u8 gpal[256][4], lpal[256][4];
    memset(gpal, 0, sizeof(gpal));
    memset(lpal, 0, sizeof(lpal));
    auto table = [&](u8 pal[256][4], u32 n, int transp) -> bool {
        if (pos + (size_t)n * 3 > d.size()) return false;
        for (u32 i = 0; i < n; ++i) {
            pal[i][0] = d[pos]; pal[i][1] = d[pos + 1]; pal[i][2] = d[pos + 2];
            pal[i][3] = transp == (int)i ? 0 : 255;
            pos += 3;
        }
        return true;
    };
    if ((flags & 0x80u) && !table(gpal, 2u << (flags & 7u), -1)) return Fail("GIF colour table truncated");
    // a header that promises far more pixels than the file could code (LZW: at most ~4096 pixels per 12-bit code)
    if ((unsigned long long)W * H > ((unsigned long long)d.size() + 64u) * 4096ull) return Fail("GIF header promises more pixels than the file can hold");
    int eflags = 0, transparent = -1;
    for (;;) {
        if (pos >= d.size()) return Fail("GIF without an image");
        const u8 tag = d[pos++];
        if (tag == 0x3B) return Fail("GIF without an image");
        if (tag == 0x21) {                                       // extension
            if (pos >= d.size()) return Fail("GIF extension truncated");
            const u8 label = d[pos++];
            if (label == 0xF9) {                                 // graphic control: flags, delay, transparent index
                if (pos >= d.size()) return Fail("GIF extension truncated");
                const u32 len = d[pos++];
                if (len == 4) {
                    if (pos + 4 > d.size()) return Fail("GIF extension truncated");
                    eflags = d[pos]; transparent = d[pos + 3];
                    pos += 4;
                } else {
                    pos += len;
                    continue;                                    // (the library goes straight back to the next block here)
                }
            }
            for (;;) {                                           // skip the data sub-blocks
                if (pos >= d.size()) return Fail("GIF extension truncated");
                const u32 len = d[pos++];
                if (!len) break;
                pos += len;
            }
            continue;
        }
        if (tag != 0x2C) return Fail("unknown GIF block");
        if (pos + 9 > d.size()) return Fail("GIF image descriptor truncated");
        const u32 x0 = Le16(&d[pos]), y0 = Le16(&d[pos + 2]), w = Le16(&d[pos + 4]), h = Le16(&d[pos + 6]), lflags = d[pos + 8];
        pos += 9;
        if (x0 + w > W || y0 + h > H) return Fail("bad GIF image descriptor");
        const u8 (*pal)[4];
        if (lflags & 0x80u) {
            if (!table(lpal, 2u << (lflags & 7u), (eflags & 1) ? transparent : -1)) return Fail("GIF colour table truncated");
            pal = lpal;
        } else if (flags & 0x80u) {
            if (transparent >= 0 && (eflags & 1)) gpal[transparent][3] = 0;
            pal = gpal;
        } else {
            return Fail("GIF without a colour table");
        }
        out->w = W; out->h = H; out->channels = 4;
        out->px.resize((size_t)W * H * 4);
        for (size_t i = 0; i < (size_t)W * H; ++i) {             // the logical screen: background colour, alpha 0
            u8 * o = &out->px[i * 4];
            o[0] = gpal[bgindex][0]; o[1] = gpal[bgindex][1]; o[2] = gpal[bgindex][2]; o[3] = 0;
        }
        // LZW (variable code size, least significant bit first, data in sub-blocks)
        if (pos >= d.size()) return Fail("GIF raster truncated");
        const u32 lzw_cs = d[pos++];
        if (lzw_cs > 12) return Fail("bad GIF code size");
        const int clear = 1 << lzw_cs;
        struct Code { int16_t prefix; u8 first, suffix; };
        std::vector<Code> codes(8192);
        for (int i = 0; i < clear; ++i) { codes[i].prefix = -1; codes[i].first = (u8)i; codes[i].suffix = (u8)i; }
        int codesize = (int)lzw_cs + 1, codemask = (1 << codesize) - 1, avail = clear + 2, oldcode = -1;
        bool first = true;
        uint32_t bits = 0;
        int valid = 0;
        u32 block = 0;
        // pixel cursor with the four interlace passes (rows 0, 8, ..; 4, 12, ..; 2, 6, ..; 1, 3, ..)
        const bool interlaced = (lflags & 0x40u) != 0;
        u32 cx = 0, cy = 0, step = interlaced ? 8 : 1;
        int parse = interlaced ? 3 : 0;
        std::vector<u8> chain;
        auto emit = [&](u8 index) {
            if (cy >= h) return;
            const u8 * c = pal[index];
            if (c[3] >= 128) {
                u8 * o = &out->px[((size_t)(y0 + cy) * W + x0 + cx) * 4];
                o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3];
            }
            if (++cx >= w) {
                cx = 0;
                cy += step;
                while (cy >= h && parse > 0) {
                    step = 1u << parse;
                    cy = step >> 1;
                    --parse;
                }
            }
        };
        if (w == 0) return true;                                 // an empty image: the background stands
        for (;;) {
            if (valid < codesize) {
                if (block == 0) {
                    if (pos >= d.size()) return true;            // the data just stops: what was decoded stands (as in the library)
                    block = d[pos++];
                    if (block == 0) return true;
                }
                --block;
                const u32 byte = pos < d.size() ? d[pos] : 0;
                ++pos;
                bits |= byte << valid;
                valid += 8;
                continue;
            }
            const int code = (int)(bits & (uint32_t)codemask);
            bits >>= codesize;
            valid -= codesize;
            if (code == clear) {
                codesize = (int)lzw_cs + 1; codemask = (1 << codesize) - 1; avail = clear + 2; oldcode = -1;
                first = false;
            } else if (code == clear + 1) {
                return true;                                     // end of the image
            } else if (code <= avail) {
                if (first) return Fail("GIF raster without a clear code");
                if (oldcode >= 0) {
                    if (avail >= 4096) return Fail("too many GIF codes");
                    Code & n = codes[avail++];
                    n.prefix = (int16_t)oldcode;
                    n.first = codes[oldcode].first;
                    n.suffix = code == avail ? n.first : codes[code].first;
                } else if (code == avail) {
                    return Fail("illegal code in GIF raster");
                }
                chain.clear();
                for (int c = code; c >= 0; c = codes[c].prefix) chain.push_back(codes[c].suffix);
                for (size_t i = chain.size(); i-- > 0;) emit(chain[i]);
                if ((avail & codemask) == 0 && avail <= 0x0FFF) { ++codesize; codemask = (1 << codesize) - 1; }
                oldcode = code;
            } else {
                return Fail("illegal code in GIF raster");
            }
        }
    }
}

// ---- PSD ------------------------------------------------------------------------------------------------
// The reference decoder's reading of a Photoshop file: the flattened composite at the end of the file, RGB mode, 8 or 16
// bits (high byte kept), raw or PackBits rows; ALWAYS four channels - missing colour channels are 0, a missing alpha is 255 -
// and where 0 < alpha < 255 the colours are un-blended from the white matte in float: c / a' + 255 (1 - 1 / a'), a' = a / 255.
u32 Be16(const u8 * p) { return ((u32)p[0] << 8) | (u32)p[1]; }

bool DecodePsd(const std::vector<u8> & d, Image * out) {
    if (d.size() < 26 + 12 + 2 || Be32(&d[0]) != 0x38425053u) return Fail("not a PSD");
    if (Be16(&d[4]) != 1) return Fail("unsupported PSD version");
    const u32 channels = Be16(&d[12]);
    if (channels > 16) return Fail("unsupported number of PSD channels");
    const u32 h = Be32(&d[14]), w = Be32(&d[18]), depth = Be16(&d[22]), mode = Be16(&d[24]);
    if (depth != 8 && depth != 16) return Fail("PSD bit depth is not 8 or 16");
    if (mode != 3) return Fail("PSD is not in RGB colour mode");
    if (!w || !h || (unsigned long long)w * h > (1ull << 28)) return Fail("bad PSD size");
    size_t pos = 26;
    for (int section = 0; section < 3; ++section) {              // colour mode data, image resources, layers and masks
        if (pos + 4 > d.size()) return Fail("PSD truncated");
        const u32 len = Be32(&d[pos]);
        pos += 4;
        if ((size_t)len > d.size() - pos) return Fail("PSD truncated");
        pos += len;
    }
    if (pos + 2 > d.size()) return Fail("PSD truncated");
    const u32 compression = Be16(&d[pos]);
    pos += 2;
    if (compression > 1) return Fail("unknown PSD compression");
    if (compression == 1 && depth == 16) return Fail("run-length coded 16-bit PSD is not supported");
    const size_t n = (size_t)w * h;
    if (n > ((size_t)d.size() + 64) * 130) return Fail("PSD header promises more pixels than the file can hold");
    std::vector<u8> px(n * 4);
    if (compression) {
        const size_t table = (size_t)h * channels * 2;           // the per-row byte counts are not needed
        if (table > d.size() - pos) return Fail("PSD truncated");
        pos += table;
    }
    for (u32 c = 0; c < 4; ++c) {
        u8 * p = px.data() + c;
        if (c >= channels) {
            for (size_t i = 0; i < n; ++i, p += 4) *p = c == 3 ? 255 : 0;
        } else if (compression) {                                // PackBits over the whole channel
            size_t count = 0;
            while (count < n) {
                if (pos >= d.size()) return Fail("bad PSD run-length data");
                u32 len = d[pos++];
                if (len == 128) continue;
                if (len < 128) {
                    ++len;
                    if (len > n - count || (size_t)len > d.size() - pos) return Fail("bad PSD run-length data");
                    for (u32 k = 0; k < len; ++k, p += 4) *p = d[pos++];
                } else {
                    len = 257 - len;
                    if (len > n - count || pos >= d.size()) return Fail("bad PSD run-length data");
                    const u8 v = d[pos++];
                    for (u32 k = 0; k < len; ++k, p += 4) *p = v;
                }
                count += len;
            }
        } else {
            const size_t bytes = depth == 16 ? 2 : 1;
            if (n * bytes > d.size() - pos) return Fail("PSD pixel data truncated");
            for (size_t i = 0; i < n; ++i, p += 4, pos += bytes) *p = d[pos];      // 16 bit: the high byte comes first
        }
    }
    if (channels >= 4) {
        for (size_t i = 0; i < n; ++i) {
            u8 * q = &px[4 * i];
            if (q[3] != 0 && q[3] != 255) {
                const float a = q[3] / 255.0f;
                const float ra = 1.0f / a;
                const float inv_a = 255.0f * (1 - ra);
                for (int k = 0; k < 3; ++k) q[k] = (u8)(int)(q[k] * ra + inv_a);
            }
        }
    }
    out->w = w; out->h = h; out->channels = 4;
    out->px.swap(px);
    return true;
}

// ---- Radiance HDR ------------------------------------------------------------------------------------
// The reference loads .hdr files through its decoder's 8-bit entry point: RGBE pixels become floats (mantissa x
// 2^(exponent - 136), exponent 0 = black) and are then tone-mapped the library's way - pow(v, 1 / 2.2) in double, x 255 +
// 0.5 in float, clamped, truncated - to three channels.  Header: "#?RADIANCE" or "#?RGBE", a FORMAT=32-bit_rle_rgbe
// line, a blank line, "-Y h +X w".  Scanlines are new-style run-length coded (2, 2, width) for 8 <= width < 32768, flat
// RGBE otherwise; a scanline that does not start with (2, 2, < 128) switches the WHOLE image to flat pixels, re-read from
// pixel 1 of row 0 (the library's control flow, kept).
bool HdrLine(const std::vector<u8> & d, size_t * pos, std::string * line) {
    line->clear();
    if (*pos >= d.size()) return false;
    while (*pos < d.size() && d[*pos] != '\n') {
        if (line->size() < 1022) line->push_back((char)d[*pos]);
        ++*pos;
    }
    if (*pos < d.size()) ++*pos;                                 // the newline
    return true;
}

void HdrPixel(const u8 * rgbe, u8 * o) {
    float v[3] = { 0.0f, 0.0f, 0.0f };
    if (rgbe[3] != 0) {
        const float f1 = (float)ldexp(1.0f, (int)rgbe[3] - (int)(128 + 8));
        for (int k = 0; k < 3; ++k) v[k] = rgbe[k] * f1;
    }
    for (int k = 0; k < 3; ++k) {
        float z = (float)pow((double)(v[k] * 1.0f), (double)(1.0f / 2.2f)) * 255 + 0.5f;
        if (z < 0) z = 0;
        if (z > 255) z = 255;
        o[k] = (u8)(int)z;
    }
}

bool DecodeHdr(const std::vector<u8> & d, Image * out) {
    size_t pos = 0;
    std::string line;
    if (!HdrLine(d, &pos, &line) || (line != "#?RADIANCE" && line != "#?RGBE")) return Fail("not a Radiance HDR file");
    bool valid = false;
    for (;;) {
        if (!HdrLine(d, &pos, &line)) return Fail("HDR header truncated");
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") valid = true;
    }
    if (!valid) return Fail("unsupported HDR format");
    if (!HdrLine(d, &pos, &line)) return Fail("HDR header truncated");
    if (line.compare(0, 3, "-Y ") != 0) return Fail("unsupported HDR data layout");
    char * end = nullptr;
    const long height = strtol(line.c_str() + 3, &end, 10);
    while (*end == ' ') ++end;
    if (strncmp(end, "+X ", 3) != 0) return Fail("unsupported HDR data layout");
    const long width = strtol(end + 3, nullptr, 10);
    if (width <= 0 || height <= 0 || (unsigned long long)width * (unsigned long long)height > (1ull << 28)) return Fail("bad HDR size");
    const size_t w = (size_t)width, h = (size_t)height;
    if (w * h > ((size_t)d.size() + 64) * 130) return Fail("HDR header promises more pixels than the file can hold");
    std::vector<u8> px(w * h * 3);
    auto get = [&]() -> u8 { return pos < d.size() ? d[pos++] : (u8)0; };
    auto flat_from = [&](size_t first) {                          // flat RGBE pixels for pixel `first` .. the end, in reading order
        for (size_t i = first; i < w * h; ++i) {
            u8 rgbe[4] = { get(), get(), get(), get() };
            HdrPixel(rgbe, &px[i * 3]);
        }
    };
    if (w < 8 || w >= 32768) {
        flat_from(0);
    } else {
        std::vector<u8> scan(w * 4);
        for (size_t j = 0; j < h; ++j) {
            const u8 c1 = get(), c2 = get(), hi = get();
            if (c1 != 2 || c2 != 2 || (hi & 0x80)) {              // not run-length coded: these four bytes are pixel 0
                u8 rgbe[4] = { c1, c2, hi, get() };
                HdrPixel(rgbe, &px[0]);
                flat_from(1);
                break;
            }
            const size_t len = ((size_t)hi << 8) | get();
            if (len != w) return Fail("invalid HDR scanline length");
            for (int k = 0; k < 4; ++k) {
                size_t i = 0;
                while (i < w) {
                    if (pos >= d.size()) return Fail("HDR run-length data truncated");
                    u32 count = get();
                    if (count > 128) {
                        const u8 value = get();
                        count -= 128;
                        if (count > w - i) return Fail("bad run-length data in HDR");
                        for (u32 z = 0; z < count; ++z) scan[i++ * 4 + k] = value;
                    } else {
                        if (count > w - i) return Fail("bad run-length data in HDR");
                        if (count == 0) return Fail("bad run-length data in HDR");      // (the library would spin here)
                        for (u32 z = 0; z < count; ++z) scan[i++ * 4 + k] = get();
                    }
                }
            }
            for (size_t i = 0; i < w; ++i) HdrPixel(&scan[i * 4], &px[(j * w + i) * 3]);
        }
    }
    out->w = (u32)w; out->h = (u32)h; out->channels = 3;
    out->px.swap(px);
    return true;
}

// ---- Softimage PIC -------------------------------------------------------------------------------------
// Header: magic 53 80 F6 34, "PICT" at byte 88, width and height at 92 (big endian); then a chain of channel packets
// (8 bits each; channel mask 0x80 R, 0x40 G, 0x20 B, 0x10 A) and, per scanline and packet, the data: raw, pure
// run-length (count, value) or mixed run-length (count >= 128: a run of count - 127, or of a 16-bit count after 128;
// else count + 1 raw values).  Channels no packet writes stay 255.  4 channels if a packet carries alpha, else 3.
bool DecodePic(const std::vector<u8> & d, Image * out) {
    if (d.size() < 104 || d[0] != 0x53 || d[1] != 0x80 || d[2] != 0xF6 || d[3] != 0x34 || memcmp(&d[88], "PICT", 4) != 0) return Fail("not a PIC");
    const u32 w = Be16(&d[92]), h = Be16(&d[94]);
    if (!w || !h) return Fail("empty PIC");
    size_t pos = 104;                                            // after ratio, fields, pad
    struct Packet { u8 size, type, channel; } packets[10];
    int n_packets = 0, act = 0;
    for (;;) {
        if (n_packets == 10) return Fail("too many PIC packets");
        if (pos + 4 > d.size()) return Fail("PIC file too short (reading packets)");
        const u8 chained = d[pos];
        Packet & p = packets[n_packets++];
        p.size = d[pos + 1]; p.type = d[pos + 2]; p.channel = d[pos + 3];
        pos += 4;
        act |= p.channel;
        if (pos >= d.size()) return Fail("PIC file too short (reading packets)");
        if (p.size != 8) return Fail("PIC packet is not 8 bits per channel");
        if (!chained) break;
    }
    if ((size_t)w * h > ((size_t)d.size() + 64) * 70000) return Fail("PIC header promises more pixels than the file can hold");
    std::vector<u8> px((size_t)w * h * 4, 0xFF);
    auto readval = [&](int channel, u8 * dest) -> bool {
        for (int i = 0, mask = 0x80; i < 4; ++i, mask >>= 1)
            if (channel & mask) {
                if (pos >= d.size()) return false;
                dest[i] = d[pos++];
            }
        return true;
    };
    auto copyval = [](int channel, u8 * dest, const u8 * src) {
        for (int i = 0, mask = 0x80; i < 4; ++i, mask >>= 1)
            if (channel & mask) dest[i] = src[i];
    };
    for (u32 y = 0; y < h; ++y)
        for (int k = 0; k < n_packets; ++k) {
            const Packet & p = packets[k];
            u8 * dest = &px[(size_t)y * w * 4];
            if (p.type == 0) {
                for (u32 x = 0; x < w; ++x, dest += 4)
                    if (!readval(p.channel, dest)) return Fail("PIC file too short");
            } else if (p.type == 1) {
                int left = (int)w;
                while (left > 0) {
                    if (pos >= d.size()) return Fail("PIC file too short (pure read count)");
                    int count = d[pos++];
                    if (pos >= d.size()) return Fail("PIC file too short (pure read count)");
                    if (count > left) count = left;
                    u8 value[4] = { 0, 0, 0, 0 };
                    if (!readval(p.channel, value)) return Fail("PIC file too short");
                    for (int i = 0; i < count; ++i, dest += 4) copyval(p.channel, dest, value);
                    left -= count;
                    if (count == 0) return Fail("bad PIC run");       // (the library would spin on a zero count)
                }
            } else if (p.type == 2) {
                int left = (int)w;
                while (left > 0) {
                    if (pos >= d.size()) return Fail("PIC file too short (mixed read count)");
                    int count = d[pos++];
                    if (pos >= d.size()) return Fail("PIC file too short (mixed read count)");
                    if (count >= 128) {
                        if (count == 128) {
                            if (pos + 2 > d.size()) return Fail("PIC file too short (mixed read count)");
                            count = (int)Be16(&d[pos]);
                            pos += 2;
                        } else {
                            count -= 127;
                        }
                        if (count > left) return Fail("PIC scanline overrun");
                        u8 value[4] = { 0, 0, 0, 0 };
                        if (!readval(p.channel, value)) return Fail("PIC file too short");
                        for (int i = 0; i < count; ++i, dest += 4) copyval(p.channel, dest, value);
                        if (count == 0) return Fail("bad PIC run");
                    } else {
                        ++count;
                        if (count > left) return Fail("PIC scanline overrun");
                        for (int i = 0; i < count; ++i, dest += 4)
                            if (!readval(p.channel, dest)) return Fail("PIC file too short");
                    }
                    left -= count;
                }
            } else {
                return Fail("PIC packet has a bad compression type");
            }
        }
    const u32 ch = (act & 0x10) ? 4 : 3;
    out->w = w; out->h = h; out->channels = ch;
    if (ch == 4) {
        out->px.swap(px);
    } else {
        out->px.resize((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; ++i) memcpy(&out->px[i * 3], &px[i * 4], 3);
    }
    return true;
}

// ---- PNM ------------------------------------------------------------------------------------------------
bool PnmNumber(const std::vector<u8> & d, size_t * pos, u32 * value) {
    for (;;) {
        while (*pos < d.size() && isspace(d[*pos])) ++*pos;
        if (*pos < d.size() && d[*pos] == '#') { while (*pos < d.size() && d[*pos] != '\n') ++*pos; continue; }
        break;
    }
    if (*pos >= d.size() || !isdigit(d[*pos])) return false;
    u32 v = 0;
    while (*pos < d.size() && isdigit(d[*pos])) { v = v * 10 + (u32)(d[*pos] - '0'); ++*pos; }
    *value = v;
    return true;
}

bool DecodePnm(const std::vector<u8> & d, Image * out) {
    if (d.size() < 3 || d[0] != 'P' || (d[1] != '5' && d[1] != '6')) return Fail("not a binary PGM / PPM");
    const u32 ch = d[1] == '5' ? 1 : 3;
    size_t pos = 2;
    u32 w = 0, h = 0, maxval = 0;
    if (!PnmNumber(d, &pos, &w) || !PnmNumber(d, &pos, &h) || !PnmNumber(d, &pos, &maxval)) return Fail("bad PNM header");
    if (maxval == 0 || maxval > 255) return Fail("PNM maxval above 255 is not supported");
    ++pos;                                                     // the single whitespace byte after maxval
    if (!w || !h || pos + (size_t)w * h * ch > d.size()) return Fail("PNM pixel data truncated");
    out->w = w; out->h = h; out->channels = ch;
    out->px.assign(d.begin() + (long)pos, d.begin() + (long)(pos + (size_t)w * h * ch));
    return true;
}

}  // namespace

const char * TextureLoadError() { return gImageError.c_str(); }

// obj_parser.cpp:197-213: NULL (with a message) when the decoder rejects the file.
Texture * LoadTexture(const char * filename) {
    Bytes file;
    Image img;
    bool ok = file.Read(filename);
    if (ok) try {
        const std::vector<u8> & d = file.data;
        if (d.size() >= 8 && d[0] == 0x89 && d[1] == 'P') ok = DecodePng(d, &img);
        else if (d.size() >= 2 && d[0] == 'B' && d[1] == 'M') ok = DecodeBmp(d, &img);
        else if (d.size() >= 6 && d[0] == 'G' && d[1] == 'I' && d[2] == 'F' && d[3] == '8') ok = DecodeGif(d, &img);
        else if (d.size() >= 4 && d[0] == '8' && d[1] == 'B' && d[2] == 'P' && d[3] == 'S') ok = DecodePsd(d, &img);
        else if (d.size() >= 92 && d[0] == 0x53 && d[1] == 0x80 && d[2] == 0xF6 && d[3] == 0x34 && !memcmp(&d[88], "PICT", 4)) ok = DecodePic(d, &img);
        else if ((d.size() >= 11 && !memcmp(d.data(), "#?RADIANCE\n", 11)) || (d.size() >= 7 && !memcmp(d.data(), "#?RGBE\n", 7))) ok = DecodeHdr(d, &img);
        else if (d.size() >= 2 && d[0] == 'P' && (d[1] == '5' || d[1] == '6')) ok = DecodePnm(d, &img);
        else if (d.size() >= 3 && d[0] == 0xFF && d[1] == 0xD8) {
            const char * err = prt_jpeg::Decode(d, &img.w, &img.h, &img.channels, &img.px);
            ok = err ? Fail(err) : true;
        }
        else ok = DecodeTga(d, &img);                          // TGA has no signature: last
    } catch (const std::bad_alloc &) {
        ok = Fail("out of memory while decoding");
    }
    if (!ok) {
        fprintf(stderr, "Failed to load image [%s] :: %s\n", filename, gImageError.c_str());
        return NULL;
    }
    Texture * result = (Texture *)calloc(1, sizeof(Texture));
    result->size_x = img.w;
    result->size_y = img.h;
    result->channels = img.channels;
    result->texels = (u8 *)malloc(img.px.size());
    memcpy(result->texels, img.px.data(), img.px.size());
    return result;
}

void FreeTexture(Texture * t) {
    if (!t) return;
    free(t->texels);
    free(t);
}

// ---- texture.cpp:85-143: height map -> tangent-space normal map ------------------------------------------
namespace {

float SrgbToLinear(float srgb) {                               // color.h:13-21
    if (srgb <= 0.04045f) return srgb / 12.92f;
    return powf((srgb + 0.055f) / 1.055f, 2.4f);
}

float LinearToSrgb(float linear) {                             // color.h:3-11
    if (linear <= 0.0031308f) return 12.92f * linear;
    return 1.055f * powf(linear, 1.0f / 2.4f) - 0.055f;
}

// GetTexel(...).x of texture.cpp:17-51: first channel, through the sRGB curve.
float HeightAt(const Texture * t, u32 x, u32 y) {
    const float one_over_255 = 1.0f / 255.0f;
    u8 r = t->texels[((size_t)y * t->size_x + x) * t->channels];
    return SrgbToLinear((float)r * one_over_255);
}

}  // namespace

// Forward differences with wrap-around, slope scale 2.5, the normal stored sRGB-encoded in 3 bytes (truncated,
// not rounded).  Note the reference's axes: the x component is the difference along +y of the image, y along +x.
Texture * ConvertHeightMapToNormalMap(const Texture * height_map) {
    Texture * result = (Texture *)calloc(1, sizeof(Texture));
    result->size_x = height_map->size_x;
    result->size_y = height_map->size_y;
    result->channels = 3;
    result->texels = (u8 *)calloc((size_t)result->size_x * result->size_y, 3);
    for (u32 y = 0; y < result->size_y; ++y) {
        for (u32 x = 0; x < result->size_x; ++x) {
            u32 x1 = (x + 1) % result->size_x;
            u32 y1 = (y + 1) % result->size_y;
            float h00 = HeightAt(height_map, x, y);
            float h10 = HeightAt(height_map, x1, y);
            float h01 = HeightAt(height_map, x, y1);
            float a = 2.5f;
            Vector3 n = Normalize(Vector3((h01 - h00) * a, (h10 - h00) * a, 1.0f));
            n = (n + Vector3(1.0f, 1.0f, 1.0f)) * 0.5f;                       // (-1, 1) -> (0, 1), texture.cpp:93
            u8 * o = result->texels + ((size_t)y * result->size_x + x) * 3;
            o[0] = (u8)(LinearToSrgb(n.x) * 255.0f);
            o[1] = (u8)(LinearToSrgb(n.y) * 255.0f);
            o[2] = (u8)(LinearToSrgb(n.z) * 255.0f);
        }
    }
    return result;
}
