// image_jpeg.cpp - baseline and progressive JPEG decoder behind LoadTexture (obj_parser.cpp:197-213).
//
// The reference reads JPEG textures through its third-party image library (lib/stb_image.h v2.14,
// `stbi_load(name, &x, &y, &channels, 0)`).  The hot path indexes the decoded bytes directly (texture.cpp:17-51), so a
// drop-in has to produce THAT library's bytes, and a JPEG decoder has freedom exactly where bytes are made: the inverse
// DCT, the chroma upsampling filter and the YCbCr -> RGB conversion.  This is a fresh decoder of the standard
// (ITU T.81: markers, Huffman decoding per F.2.2, restart intervals, interleaved and single-component scans) with
// those three stages restated from the library's published algorithm:
//   * inverse DCT: the IJG "islow" butterfly with 12-bit constants, columns first keeping 2 extra bits
//     ((x + 512) >> 10), then rows ((x + 65536 + (128 << 17)) >> 17), clamped to 0..255; a column whose AC terms
//     are all zero is the DC term << 2;
//   * chroma upsampling: 2x horizontally by the (3, 1) / 4 filter, 2x vertically by (3, 1) / 4 between the nearer and the
//     farther row, 2x2 by both ((3a + b) * 3 + (3c + d) + 8) >> 4 with (.. + 2) >> 2 at the row ends; any other
//     ratio repeats samples; the nearer / farther row bookkeeping follows the library's row stepping;
//   * colour: r = y + 1.40200 cr, g = y - 0.71414 cr - 0.34414 cb, b = y + 1.77200 cb in 20-bit fixed point with
//     12-bit constants, the cb term of g truncated to 16 bits before the sum, rounding by + 2^19; files whose component
//     ids are 'R', 'G', 'B' are copied through.
// Output convention: 1 channel for a grey file, 3 for a colour file, rows top to bottom.
// Progressive files (SOF2, T.81 Annex G: spectral selection and successive approximation, end-of-band runs, DC and AC
// refinement scans) build their coefficients up scan by scan and are dequantised and transformed at the end.
// Not handled (reported, the texture slot stays empty like a decoder failure in the reference): arithmetic coding,
// lossless / hierarchical processes, 12-bit samples, 4 components.  tests/test_host_side.py compares the decoded bytes with the
// reference's on generated files of every sampling layout.
#include <cstdint>
#include <cstring>
#include <vector>

#include "prt_scene.h"

namespace prt_jpeg {

namespace {

const unsigned char kZigzag[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct Huffman {             // T.81 F.2.2.3: codes of length l are mincode[l] .. maxcode[l], values from valptr[l]
    int mincode[17], maxcode[18], valptr[17];
    unsigned char values[256];
    bool defined = false;
    bool Build(const int counts[16]) {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += counts[l - 1];
            k += counts[l - 1];
            maxcode[l] = counts[l - 1] ? code - 1 : -1;
            if (code > (1 << l)) return false;            // more codes than the length can hold
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        defined = true;
        return k <= 256;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
    int x = 0, y = 0;        // real size in samples
    int w2 = 0, h2 = 0;      // plane size, whole MCUs
    int dc_pred = 0;
    std::vector<u8> plane;
    std::vector<short> coeff;   // progressive: 64 coefficients per block of the (MCU-padded) plane, coeff_w blocks per row
    int coeff_w = 0;
};

struct Decoder {
    const u8 * p, * end;
    const char * error = nullptr;
    int width = 0, height = 0, ncomp = 0, rgb_ids = 0;
    int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    int restart_interval = 0;
    bool progressive = false;
    int ss = 0, se = 63, ah = 0, al = 0, eob_run = 0;      // progressive: the current scan's band, bit positions and pending end-of-band run
    unsigned char dequant[4][64];
    bool have_dqt[4] = { false, false, false, false };
    Huffman dc[4], ac[4];
    Component comp[3];
    // entropy-coded segment reader
    uint32_t bits = 0;
    int nbits = 0;
    int marker = 0;          // marker met inside the entropy-coded data (0 = none)

    bool Fail(const char * why) { if (!error) error = why; return false; }
    int Get8() { return p < end ? *p++ : 0; }
    int Get16() { int a = Get8(); return (a << 8) | Get8(); }

    void Fill() {
        while (nbits <= 24) {
            int b = 0;
            if (!marker) {
                b = Get8();
                if (b == 0xFF) {
                    int c = Get8();
                    while (c == 0xFF) c = Get8();            // fill bytes
                    if (c != 0) { marker = c; b = 0; }
                }
            }
            bits |= (uint32_t)b << (24 - nbits);
            nbits += 8;
        }
    }
    int GetBits(int n) {                                     // n <= 16
        if (n == 0) return 0;
        if (nbits < n) Fill();
        const int v = (int)(bits >> (32 - n));
        bits <<= n;
        nbits -= n;
        return v;
    }
    int Decode(const Huffman & h) {                          // -1: no such code
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = (code << 1) | GetBits(1);
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.values[h.valptr[l] + code - h.mincode[l]];
        }
        return -1;
    }
    static int Extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }   // T.81 F.2.2.1
    void ResetEntropy() {
        bits = 0;
        nbits = 0;
        marker = 0;
        eob_run = 0;
        for (int i = 0; i < 3; ++i) comp[i].dc_pred = 0;
    }

    bool DecodeBlock(Component & c, short data[64]) {
        const Huffman & hd = dc[c.hd];
        const Huffman & ha = ac[c.ha];
        const unsigned char * dq = dequant[c.tq];
        memset(data, 0, 64 * sizeof(short));
        const int t = Decode(hd);
        if (t < 0 || t > 15) return Fail("bad huffman code");
        const int diff = t ? Extend(GetBits(t), t) : 0;
        c.dc_pred += diff;
        data[0] = (short)(c.dc_pred * dq[0]);
        for (int k = 1; k < 64;) {
            const int rs = Decode(ha);
            if (rs < 0) return Fail("bad huffman code");
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xF0) break;                       // end of block
                k += 16;
            } else {
                k += r;
                if (k > 63) return Fail("bad huffman code");
                const int zig = kZigzag[k++];
                data[zig] = (short)(Extend(GetBits(s), s) * dq[zig]);
            }
        }
        return true;
    }

    // ---- progressive scans (T.81 G.1.2): coefficients are built up over several scans in `coeff`, natural order ------
    bool ProgDc(Component & c, short * data) {
        if (ah == 0) {                                       // first DC scan: the difference, scaled by the point transform
            memset(data, 0, 64 * sizeof(short));
            const int t = Decode(dc[c.hd]);
            if (t < 0 || t > 15) return Fail("bad huffman code");
            const int diff = t ? Extend(GetBits(t), t) : 0;
            c.dc_pred += diff;
            data[0] = (short)((unsigned int)c.dc_pred << al);
        } else if (GetBits(1)) {                             // refinement: one more bit
            data[0] = (short)(data[0] + (short)(1 << al));
        }
        return true;
    }
    void Refine(short * p, short bit) {                      // a coefficient that is already non-zero takes one correction bit
        if (GetBits(1) && (*p & bit) == 0) *p = (short)(*p > 0 ? *p + bit : *p - bit);
    }
    bool ProgAc(Component & c, short * data) {
        const Huffman & ha = ac[c.ha];
        if (ah == 0) {                                       // first scan of the band ss..se
            if (eob_run) { --eob_run; return true; }
            int k = ss;
            do {
                const int rs = Decode(ha);
                if (rs < 0) return Fail("bad huffman code");
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {                            // EOBn: this block and eob_run more end here
                        eob_run = (1 << r) - 1;
                        if (r) eob_run += GetBits(r);
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) return Fail("bad huffman code");
                    data[kZigzag[k++]] = (short)((unsigned int)Extend(GetBits(s), s) << al);
                }
            } while (k <= se);
            return true;
        }
        const short bit = (short)(1 << al);                  // refinement scan of the band
        if (eob_run) {
            --eob_run;
            for (int k = ss; k <= se; ++k) {
                short * p = &data[kZigzag[k]];
                if (*p != 0) Refine(p, bit);
            }
            return true;
        }
        int k = ss;
        do {
            const int rs = Decode(ha);
            if (rs < 0) return Fail("bad huffman code");
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1;
                    if (r) eob_run += GetBits(r);
                    r = 64;                                  // the rest of the band only takes correction bits
                }                                            // else ZRL: sixteen zero-history coefficients are skipped
            } else {
                if (s != 1) return Fail("bad huffman code");
                s = GetBits(1) ? bit : -bit;                 // the new coefficient's sign
            }
            while (k <= se) {
                short * p = &data[kZigzag[k++]];
                if (*p != 0) {
                    Refine(p, bit);
                } else {
                    if (r == 0) { *p = (short)s; break; }
                    --r;
                }
            }
        } while (k <= se);
        return true;
    }
    void FinishProgressive() {                               // dequantise and transform what the scans have built
        for (int n = 0; n < ncomp; ++n) {
            Component & c = comp[n];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    short * data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                    for (int q = 0; q < 64; ++q) data[q] = (short)(data[q] * dequant[c.tq][q]);
                    IdctBlock(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
                }
        }
    }

    // ---- inverse DCT -------------------------------------------------------------------------------------------
    static int F2F(double x) { return (int)(x * 4096 + 0.5); }
    static u8 Clamp(int x) { return (u8)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
    // 32-bit arithmetic that wraps (corrupt files can push the butterfly past INT_MAX; on valid data nothing wraps)
    struct W {
        uint32_t v;
        W() : v(0) {}
        W(int x) : v((uint32_t)x) {}
        W operator+(W o) const { W r; r.v = v + o.v; return r; }
        W operator-(W o) const { W r; r.v = v - o.v; return r; }
        W operator*(W o) const { W r; r.v = v * o.v; return r; }
        int i() const { return (int)v; }
    };
    struct Odd { W x0, x1, x2, x3, t0, t1, t2, t3; };
    static Odd Idct1D(W s0, W s1, W s2, W s3, W s4, W s5, W s6, W s7) {
        static const int c0541 = F2F(0.5411961f), c1847 = F2F(-1.847759065f), c0765 = F2F(0.765366865f), c1175 = F2F(1.175875602f),
                         c0298 = F2F(0.298631336f), c2053 = F2F(2.053119869f), c3072 = F2F(3.072711026f), c1501 = F2F(1.501321110f),
                         c0899 = F2F(-0.899976223f), c2562 = F2F(-2.562915447f), c1961 = F2F(-1.961570560f), c0390 = F2F(-0.390180644f);
        Odd o;
        W p2 = s2, p3 = s6;
        W p1 = (p2 + p3) * c0541;
        W t2 = p1 + p3 * c1847;
        W t3 = p1 + p2 * c0765;
        p2 = s0;
        p3 = s4;
        W t0 = (p2 + p3) * 4096;
        W t1 = (p2 - p3) * 4096;
        o.x0 = t0 + t3;
        o.x3 = t0 - t3;
        o.x1 = t1 + t2;
        o.x2 = t1 - t2;
        t0 = s7;
        t1 = s5;
        t2 = s3;
        t3 = s1;
        p3 = t0 + t2;
        W p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        const W p5 = (p3 + p4) * c1175;
        t0 = t0 * c0298;
        t1 = t1 * c2053;
        t2 = t2 * c3072;
        t3 = t3 * c1501;
        p1 = p5 + p1 * c0899;
        p2 = p5 + p2 * c2562;
        p3 = p3 * c1961;
        p4 = p4 * c0390;
        o.t3 = t3 + p1 + p4;
        o.t2 = t2 + p2 + p3;
        o.t1 = t1 + p2 + p4;
        o.t0 = t0 + p1 + p3;
        return o;
    }
    static void IdctBlock(u8 * out, int stride, const short d[64]) {
        int val[64];
        for (int i = 0; i < 8; ++i) {
            int * v = val + i;
            const short * c = d + i;
            if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
                const int dcterm = c[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
            } else {
                Odd o = Idct1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
                const W r = 512;
                o.x0 = o.x0 + r; o.x1 = o.x1 + r; o.x2 = o.x2 + r; o.x3 = o.x3 + r;
                v[0] = (o.x0 + o.t3).i() >> 10;
                v[56] = (o.x0 - o.t3).i() >> 10;
                v[8] = (o.x1 + o.t2).i() >> 10;
                v[48] = (o.x1 - o.t2).i() >> 10;
                v[16] = (o.x2 + o.t1).i() >> 10;
                v[40] = (o.x2 - o.t1).i() >> 10;
                v[24] = (o.x3 + o.t0).i() >> 10;
                v[32] = (o.x3 - o.t0).i() >> 10;
            }
        }
        for (int i = 0; i < 8; ++i) {
            const int * v = val + 8 * i;
            u8 * o8 = out + (size_t)stride * i;
            Odd o = Idct1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            const W bias = 65536 + (128 << 17);
            o.x0 = o.x0 + bias; o.x1 = o.x1 + bias; o.x2 = o.x2 + bias; o.x3 = o.x3 + bias;
            o8[0] = Clamp((o.x0 + o.t3).i() >> 17);
            o8[7] = Clamp((o.x0 - o.t3).i() >> 17);
            o8[1] = Clamp((o.x1 + o.t2).i() >> 17);
            o8[6] = Clamp((o.x1 - o.t2).i() >> 17);
            o8[2] = Clamp((o.x2 + o.t1).i() >> 17);
            o8[5] = Clamp((o.x2 - o.t1).i() >> 17);
            o8[3] = Clamp((o.x3 + o.t0).i() >> 17);
            o8[4] = Clamp((o.x3 - o.t0).i() >> 17);
        }
    }

    // ---- markers -----------------------------------------------------------------------------------------------
    int NextMarker() {                                       // 0 at the end of the data
        if (marker) { const int m = marker; marker = 0; return m; }
        int x = Get8();
        if (x != 0xFF) return 0;
        while (x == 0xFF) x = Get8();
        return x;
    }
    bool ReadDqt() {
        int len = Get16() - 2;
        while (len > 0) {
            const int q = Get8(), prec = q >> 4, t = q & 15;
            if (prec != 0) return Fail("16-bit quantisation tables are not supported");
            if (t > 3) return Fail("bad DQT table");
            for (int i = 0; i < 64; ++i) dequant[t][kZigzag[i]] = (unsigned char)Get8();
            have_dqt[t] = true;
            len -= 65;
        }
        return len == 0 ? true : Fail("bad DQT length");
    }
    bool ReadDht() {
        int len = Get16() - 2;
        while (len > 0) {
            const int q = Get8(), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) return Fail("bad DHT header");
            int counts[16], n = 0;
            for (int i = 0; i < 16; ++i) { counts[i] = Get8(); n += counts[i]; }
            if (n > 256) return Fail("bad DHT counts");
            Huffman & h = tc ? ac[th] : dc[th];
            if (!h.Build(counts)) return Fail("bad code lengths");
            for (int i = 0; i < n; ++i) h.values[i] = (unsigned char)Get8();
            len -= 17 + n;
        }
        return len == 0 ? true : Fail("bad DHT length");
    }
    bool ReadSof() {
        const int len = Get16();
        if (len < 11) return Fail("bad SOF length");
        if (Get8() != 8) return Fail("only 8-bit samples are supported");
        height = Get16();
        width = Get16();
        if (height == 0 || width == 0) return Fail("empty image");
        ncomp = Get8();
        if (ncomp != 1 && ncomp != 3) return Fail("bad component count");
        if (len != 8 + 3 * ncomp) return Fail("bad SOF length");
        if ((uint64_t)width * (uint64_t)height > (1ull << 28)) return Fail("JPEG larger than 2^28 pixels");
        // a 16 x 16 MCU of a flat image still takes more than a byte: a header that promises far more pixels than the
        // file could hold is not worth the allocation
        if ((uint64_t)width * (uint64_t)height > ((uint64_t)(end - p) + 1024u) * 1024u) return Fail("JPEG header promises more pixels than the file can hold");
        rgb_ids = 0;
        for (int i = 0; i < ncomp; ++i) {
            static const int rgb[3] = { 'R', 'G', 'B' };
            Component & c = comp[i];
            c.id = Get8();
            if (c.id != i + 1 && c.id != i) {
                if (c.id != rgb[i]) return Fail("bad component id");
                ++rgb_ids;
            }
            const int q = Get8();
            c.h = q >> 4;
            c.v = q & 15;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return Fail("bad sampling factor");
            c.tq = Get8();
            if (c.tq > 3) return Fail("bad quantisation table index");
        }
        h_max = v_max = 1;
        for (int i = 0; i < ncomp; ++i) { if (comp[i].h > h_max) h_max = comp[i].h; if (comp[i].v > v_max) v_max = comp[i].v; }
        mcu_x = (width + 8 * h_max - 1) / (8 * h_max);
        mcu_y = (height + 8 * v_max - 1) / (8 * v_max);
        for (int i = 0; i < ncomp; ++i) {
            Component & c = comp[i];
            c.x = (width * c.h + h_max - 1) / h_max;
            c.y = (height * c.v + v_max - 1) / v_max;
            c.w2 = mcu_x * c.h * 8;
            c.h2 = mcu_y * c.v * 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) {
                c.coeff_w = c.w2 / 8;
                c.coeff.assign((size_t)c.w2 * c.h2, 0);
            }
        }
        return true;
    }
    bool ReadScan() {
        const int len = Get16();
        const int n = Get8();
        if (n < 1 || n > ncomp) return Fail("bad SOS component count");
        if (len != 6 + 2 * n) return Fail("bad SOS length");
        int order[3];
        for (int i = 0; i < n; ++i) {
            const int id = Get8(), q = Get8();
            int which = 0;
            while (which < ncomp && comp[which].id != id) ++which;
            if (which == ncomp) return Fail("bad SOS component");
            comp[which].hd = q >> 4;
            comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) return Fail("bad huffman table index");
            if (!progressive && !have_dqt[comp[which].tq]) return Fail("scan uses an undefined quantisation table");
            order[i] = which;
        }
        ss = Get8();
        se = Get8();
        const int aa = Get8();
        ah = aa >> 4;
        al = aa & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return Fail("bad SOS");
            if (ss == 0 && se != 0) return Fail("a progressive scan cannot hold DC and AC coefficients");
            if (ss != 0 && n != 1) return Fail("a progressive AC scan holds one component");
        } else {
            if (ss != 0 || ah != 0 || al != 0) return Fail("bad SOS");
            se = 63;
        }
        for (int i = 0; i < n; ++i) {
            const Component & c = comp[order[i]];
            const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss != 0;
            if ((need_dc && !dc[c.hd].defined) || (need_ac && !ac[c.ha].defined)) return Fail("scan uses an undefined huffman table");
        }
        ResetEntropy();
        int todo = restart_interval ? restart_interval : 0x7FFFFFFF;
        short data[64];
        auto restart = [&]() -> int {                        // 1: go on, 0: the scan ends here
            if (--todo > 0) return 1;
            if (nbits < 24) Fill();
            if (marker < 0xD0 || marker > 0xD7) return 0;
            ResetEntropy();
            todo = restart_interval ? restart_interval : 0x7FFFFFFF;
            return 1;
        };
        auto block = [&](Component & c, int bx, int by) -> bool {      // one 8 x 8 block of component c at block position (bx, by)
            if (progressive) {
                short * cf = c.coeff.data() + 64 * ((size_t)bx + (size_t)by * c.coeff_w);
                return ss == 0 ? ProgDc(c, cf) : ProgAc(c, cf);
            }
            if (!DecodeBlock(c, data)) return false;
            IdctBlock(c.plane.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2, data);
            return true;
        };
        if (n == 1) {                                        // one component: its blocks in raster order
            Component & c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    if (!block(c, i, j)) return false;
                    if (!restart()) return true;
                }
        } else {                                             // interleaved: MCU by MCU, h x v blocks of every component
            for (int j = 0; j < mcu_y; ++j)
                for (int i = 0; i < mcu_x; ++i) {
                    for (int k = 0; k < n; ++k) {
                        Component & c = comp[order[k]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x)
                                if (!block(c, i * c.h + x, j * c.v + y)) return false;
                    }
                    if (!restart()) return true;
                }
        }
        return true;
    }

    bool Run() {
        if (Get8() != 0xFF || Get8() != 0xD8) return Fail("no SOI");
        bool have_frame = false, have_scan = false;
        for (;;) {
            const int m = NextMarker();
            if (m == 0) {
                if (p >= end) break;                          // ran out of data: what was decoded stands (or nothing was)
                continue;                                     // junk between segments
            }
            if (m == 0xD9) break;
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_frame) return Fail("two frame headers");
                progressive = m == 0xC2;
                if (!ReadSof()) return false;
                have_frame = true;
            } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
                return Fail("this JPEG process (lossless / hierarchical / arithmetic) is not supported");
            } else if (m == 0xDB) {
                if (!ReadDqt()) return false;
            } else if (m == 0xC4) {
                if (!ReadDht()) return false;
            } else if (m == 0xDD) {
                if (Get16() != 4) return Fail("bad DRI length");
                restart_interval = Get16();
            } else if (m == 0xDA) {
                if (!have_frame) return Fail("scan before the frame header");
                if (!ReadScan()) return false;
                have_scan = true;
            } else if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
                const int len = Get16();
                if (len < 2 || (long)(end - p) < len - 2) return Fail("bad segment length");
                p += len - 2;
            } else if (m >= 0xD0 && m <= 0xD7) {
                // a restart marker outside a scan: ignore
            } else {
                return Fail("unknown marker");
            }
        }
        if (!have_frame || !have_scan) return Fail("no image data");
        if (progressive) {
            for (int i = 0; i < ncomp; ++i)
                if (!have_dqt[comp[i].tq]) return Fail("component uses an undefined quantisation table");
            FinishProgressive();
        }
        return true;
    }

    // ---- planes -> pixels --------------------------------------------------------------------------------------
    typedef const u8 * (*ResampleFn)(u8 * out, const u8 * nearer, const u8 * farther, int w, int hs);
    static const u8 * Row1(u8 *, const u8 * nearer, const u8 *, int, int) { return nearer; }
    static const u8 * RowV2(u8 * out, const u8 * nearer, const u8 * farther, int w, int) {
        for (int i = 0; i < w; ++i) out[i] = (u8)((3 * nearer[i] + farther[i] + 2) >> 2);
        return out;
    }
    static const u8 * RowH2(u8 * out, const u8 * in, const u8 *, int w, int) {
        if (w == 1) { out[0] = out[1] = in[0]; return out; }
        out[0] = in[0];
        out[1] = (u8)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; ++i) {
            const int n = 3 * in[i] + 2;
            out[i * 2 + 0] = (u8)((n + in[i - 1]) >> 2);
            out[i * 2 + 1] = (u8)((n + in[i + 1]) >> 2);
        }
        out[i * 2 + 0] = (u8)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[i * 2 + 1] = in[w - 1];
        return out;
    }
    static const u8 * RowHV2(u8 * out, const u8 * nearer, const u8 * farther, int w, int) {
        if (w == 1) { out[0] = out[1] = (u8)((3 * nearer[0] + farther[0] + 2) >> 2); return out; }
        int t1 = 3 * nearer[0] + farther[0];
        out[0] = (u8)((t1 + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            const int t0 = t1;
            t1 = 3 * nearer[i] + farther[i];
            out[i * 2 - 1] = (u8)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (u8)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (u8)((t1 + 2) >> 2);
        return out;
    }
    static const u8 * RowRepeat(u8 * out, const u8 * nearer, const u8 *, int w, int hs) {
        for (int i = 0; i < w; ++i)
            for (int j = 0; j < hs; ++j) out[i * hs + j] = nearer[i];
        return out;
    }

    void Output(std::vector<u8> * px) {
        const int n = ncomp;
        px->assign((size_t)width * height * n, 0);
        struct Res { ResampleFn fn; const u8 * line0, * line1; int hs, vs, w_lores, ystep, ypos; std::vector<u8> buf; } res[3];
        for (int k = 0; k < n; ++k) {
            Res & r = res[k];
            const Component & c = comp[k];
            r.hs = h_max / c.h;
            r.vs = v_max / c.v;
            r.ystep = r.vs >> 1;
            r.w_lores = (width + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.line0 = r.line1 = c.plane.data();
            r.buf.assign((size_t)r.w_lores * r.hs + 8, 0);
            if (r.hs == 1 && r.vs == 1) r.fn = Row1;
            else if (r.hs == 1 && r.vs == 2) r.fn = RowV2;
            else if (r.hs == 2 && r.vs == 1) r.fn = RowH2;
            else if (r.hs == 2 && r.vs == 2) r.fn = RowHV2;
            else r.fn = RowRepeat;
        }
        const int cr_r = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, cr_g = -(((int)(0.71414f * 4096.0f + 0.5f)) << 8),
                  cb_g = -(((int)(0.34414f * 4096.0f + 0.5f)) << 8), cb_b = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
        for (int j = 0; j < height; ++j) {
            u8 * out = px->data() + (size_t)n * width * j;
            const u8 * row[3] = { nullptr, nullptr, nullptr };
            for (int k = 0; k < n; ++k) {
                Res & r = res[k];
                const bool y_bot = r.ystep >= (r.vs >> 1);
                row[k] = r.fn(r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
                }
            }
            if (n == 1) {
                memcpy(out, row[0], (size_t)width);
            } else if (rgb_ids == 3) {
                for (int i = 0; i < width; ++i) { out[3 * i] = row[0][i]; out[3 * i + 1] = row[1][i]; out[3 * i + 2] = row[2][i]; }
            } else {
                for (int i = 0; i < width; ++i) {
                    const int y_fixed = (row[0][i] << 20) + (1 << 19);
                    const int cb = row[1][i] - 128, cr = row[2][i] - 128;
                    int r = y_fixed + cr * cr_r;
                    int g = y_fixed + cr * cr_g + (int)((unsigned int)(cb * cb_g) & 0xFFFF0000u);
                    int b = y_fixed + cb * cb_b;
                    r >>= 20; g >>= 20; b >>= 20;
                    out[3 * i] = Clamp(r);
                    out[3 * i + 1] = Clamp(g);
                    out[3 * i + 2] = Clamp(b);
                }
            }
        }
    }
};

}  // namespace

// Decodes a baseline JPEG file image.  Returns NULL on success, else a static message.
const char * Decode(const std::vector<u8> & file, u32 * w, u32 * h, u32 * channels, std::vector<u8> * px) {
    Decoder * d = new Decoder();
    d->p = file.data();
    d->end = file.data() + file.size();
    memset(d->dequant, 0, sizeof(d->dequant));
    const char * err = nullptr;
    if (!d->Run()) err = d->error ? d->error : "corrupt JPEG";
    if (!err) {
        *w = (u32)d->width;
        *h = (u32)d->height;
        *channels = (u32)d->ncomp;
        d->Output(px);
    }
    delete d;
    return err;
}

}  // namespace prt_jpeg
