// Baseline JPEG decoding for the scene reader (the reference loads its textures
// with cv::imread(file, IMREAD_COLOR), util/reader.cpp:67-69).
//
// Supported: sequential DCT (SOF0 / SOF1), 8-bit samples, Huffman coding, one or
// three components with sampling factors 1 or 2, restart intervals; JFIF YCbCr
// and grey-scale input, BGR output as cv::imread stores it.  Progressive,
// arithmetic-coded and 12-bit files are refused (the caller then asks for a
// decoded .ppm next to the file).
//
// The arithmetic follows the integer pipeline every libjpeg-derived decoder
// uses by default, so that texels agree with what OpenCV produced for the
// reference: the "slow integer" inverse DCT (13-bit constants, two passes),
// triangle-filter ("fancy") chroma upsampling for 2x1 and 2x2, and the 16-bit
// fixed-point YCbCr -> RGB tables.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "scene.hpp"

namespace pth {
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    int32_t maxcode[18];     // largest code of each length, -1 if none
    int32_t valptr[17];      // index of the first symbol of each length
    int32_t mincode[17];
    uint8_t vals[256];
    void build(const uint8_t bits[17], const uint8_t *symbols, int n) {
        std::memcpy(vals, symbols, (size_t)n);
        int32_t code = 0;
        int k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k;
            mincode[len] = code;
            code += bits[len];
            k += bits[len];
            maxcode[len] = bits[len] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int width = 0, height = 0;        // real (downsampled) size in samples
    int stride = 0, rows = 0;         // allocated size: whole blocks of whole MCUs
    std::vector<uint8_t> plane;
    int pred = 0;
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    void fill() {
        while (n <= 24) {
            uint32_t byte = 0;
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;             // stuffed zero
                    else { hit_marker = true; byte = 0; }                 // a marker: feed zeros from here on
                } else {
                    p++;
                }
            }
            acc |= byte << (24 - n);
            n += 8;
        }
    }
    int bit() {
        if (n == 0) fill();
        int b = (int)(acc >> 31);
        acc <<= 1; n--;
        return b;
    }
    int bits(int k) {
        if (k == 0) return 0;
        if (n < k) fill();
        int v = (int)(acc >> (32 - k));
        acc <<= k; n -= k;
        return v;
    }
    // byte-align and step over an RSTn marker
    bool restart() {
        acc = 0; n = 0; hit_marker = false;
        while (p + 1 < end) {
            if (p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) { p += 2; return true; }
            if (p[0] == 0xFF && p[1] == 0xFF) { p++; continue; }
            if (p[0] == 0xFF && p[1] != 0x00) return false;               // some other marker
            p++;
        }
        return false;
    }
};

inline int decode_symbol(BitReader &br, const Huff &h) {
    int32_t code = br.bit();
    int len = 1;
    while (len <= 16 && (h.maxcode[len] < 0 || code > h.maxcode[len])) {
        code = (code << 1) | br.bit();
        len++;
    }
    if (len > 16) return -1;
    return h.vals[h.valptr[len] + (code - h.mincode[len])];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// Inverse DCT, slow-but-accurate integer method: 13-bit constants, column pass kept at 2 extra bits.
void idct_block(const int32_t *in, uint8_t *out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299,
                      F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    int64_t ws[64];
    auto descale = [](int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; };
    for (int pass = 0; pass < 2; pass++) {
        for (int k = 0; k < 8; k++) {
            int64_t s[8];
            if (pass == 0) for (int j = 0; j < 8; j++) s[j] = in[8 * j + k];
            else for (int j = 0; j < 8; j++) s[j] = ws[8 * k + j];
            int64_t z2 = s[2], z3 = s[6];
            int64_t z1 = (z2 + z3) * F0541;
            int64_t t2 = z1 - z3 * F1847, t3 = z1 + z2 * F0765;
            int64_t t0 = (s[0] + s[4]) * ((int64_t)1 << CB), t1 = (s[0] - s[4]) * ((int64_t)1 << CB);   // a product: the sums may be negative
            int64_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            int64_t o0 = s[7], o1 = s[5], o2 = s[3], o3 = s[1];
            z1 = o0 + o3; z2 = o1 + o2; z3 = o0 + o2;
            int64_t z4 = o1 + o3, z5 = (z3 + z4) * F1175;
            o0 *= F0298; o1 *= F2053; o2 *= F3072; o3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
            const int64_t r[8] = {t10 + o3, t11 + o2, t12 + o1, t13 + o0, t13 - o0, t12 - o1, t11 - o2, t10 - o3};
            if (pass == 0) {
                for (int j = 0; j < 8; j++) ws[8 * j + k] = descale(r[j], CB - P1);
            } else {
                for (int j = 0; j < 8; j++) out[(size_t)k * stride + j] = clamp8((int)descale(r[j], CB + P1 + 3) + 128);
            }
        }
    }
}

uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }

// One output row of a component at full resolution.
// hx, vx: horizontal and vertical expansion (1 or 2).  Triangle filter for 2x1 and 2x2, replication for 1x2.
void upsample_row(const Component &c, int hx, int vx, int out_y, int out_w, std::vector<uint8_t> &row) {
    const int W = c.width, H = c.height;
    auto src = [&](int y) { return c.plane.data() + (size_t)(y < 0 ? 0 : (y >= H ? H - 1 : y)) * c.stride; };
    if (hx == 1) {
        const uint8_t *s = src(vx == 2 ? out_y / 2 : out_y);
        for (int x = 0; x < out_w; x++) row[x] = s[x < W ? x : W - 1];
        return;
    }
    std::vector<int> sum((size_t)W);
    int round_even, round_odd, shift, edge_mul;
    if (vx == 1) {
        const uint8_t *s = src(out_y);
        for (int x = 0; x < W; x++) sum[x] = s[x];
        round_even = 1; round_odd = 2; shift = 2; edge_mul = 1;
    } else {
        // the nearer input row weighs 3, the farther one 1
        const int y0 = out_y / 2, y1 = (out_y & 1) ? y0 + 1 : y0 - 1;
        const uint8_t *a = src(y0), *b = src(y1);
        for (int x = 0; x < W; x++) sum[x] = 3 * a[x] + b[x];
        round_even = 8; round_odd = 7; shift = 4; edge_mul = 4;
    }
    for (int x = 0; x < W; x++) {
        int left, right;
        if (vx == 1) {
            left = x == 0 ? sum[x] : (3 * sum[x] + sum[x - 1] + round_even) >> shift;
            right = x == W - 1 ? sum[x] : (3 * sum[x] + sum[x + 1] + round_odd) >> shift;
        } else {
            left = x == 0 ? (sum[x] * edge_mul + 8) >> shift : (3 * sum[x] + sum[x - 1] + round_even) >> shift;
            right = x == W - 1 ? (sum[x] * edge_mul + 7) >> shift : (3 * sum[x] + sum[x + 1] + round_odd) >> shift;
        }
        if (2 * x < out_w) row[2 * x] = (uint8_t)left;
        if (2 * x + 1 < out_w) row[2 * x + 1] = (uint8_t)right;
    }
}

}  // namespace

bool decode_jpeg_file(const std::string &path, Texture &out) {
    std::vector<uint8_t> buf;
    {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END);
        long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (n <= 4) { std::fclose(f); return false; }
        buf.resize((size_t)n);
        size_t got = std::fread(buf.data(), 1, (size_t)n, f);
        std::fclose(f);
        if (got != (size_t)n) return false;
    }
    const uint8_t *p = buf.data(), *end = p + buf.size();
    if (p[0] != 0xFF || p[1] != 0xD8) return false;
    p += 2;
    uint16_t quant[4][64];
    bool have_q[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Component> comp;
    int width = 0, height = 0, restart_interval = 0, adobe_transform = -1;
    bool have_frame = false;

    while (p + 4 <= end) {
        if (p[0] != 0xFF) return false;
        while (p + 1 < end && p[1] == 0xFF) p++;               // fill bytes
        const uint8_t m = p[1];
        p += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (p + 2 > end) return false;
        const int len = be16(p);
        if (len < 2 || p + len > end) return false;
        const uint8_t *seg = p + 2, *seg_end = p + len;
        if (m == 0xDB) {                                        // quantisation tables
            while (seg < seg_end) {
                const int pq = seg[0] >> 4, tq = seg[0] & 15;
                seg++;
                if (tq > 3 || seg + (pq ? 128 : 64) > seg_end) return false;
                for (int k = 0; k < 64; k++) {
                    quant[tq][kZigzag[k]] = pq ? be16(seg + 2 * k) : seg[k];
                }
                seg += pq ? 128 : 64;
                have_q[tq] = true;
            }
        } else if (m == 0xC4) {                                 // Huffman tables
            while (seg < seg_end) {
                const int tc = seg[0] >> 4, th = seg[0] & 15;
                if (th > 3 || tc > 1 || seg + 17 > seg_end) return false;
                uint8_t bits[17] = {0};
                int n = 0;
                for (int k = 1; k <= 16; k++) { bits[k] = seg[k]; n += bits[k]; }
                seg += 17;
                if (n > 256 || seg + n > seg_end) return false;
                (tc ? ac[th] : dc[th]).build(bits, seg, n);
                seg += n;
            }
        } else if (m == 0xC0 || m == 0xC1) {                    // frame header, sequential DCT
            if (seg_end - seg < 6 || seg[0] != 8) return false;
            height = be16(seg + 1); width = be16(seg + 3);
            const int nc = seg[5];
            if ((nc != 1 && nc != 3) || width <= 0 || height <= 0 || seg_end - seg < 6 + 3 * nc) return false;
            comp.resize((size_t)nc);
            for (int k = 0; k < nc; k++) {
                comp[k].id = seg[6 + 3 * k];
                comp[k].h = seg[7 + 3 * k] >> 4; comp[k].v = seg[7 + 3 * k] & 15;
                comp[k].tq = seg[8 + 3 * k];
                if (comp[k].h < 1 || comp[k].h > 2 || comp[k].v < 1 || comp[k].v > 2 || comp[k].tq > 3) return false;
                for (int q = 0; q < k; q++)
                    if (comp[q].id == comp[k].id) return false;         // component ids must differ
            }
            have_frame = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return false;                                       // progressive, lossless, arithmetic: not handled
        } else if (m == 0xDD) {
            if (seg_end - seg < 2) return false;
            restart_interval = be16(seg);
        } else if (m == 0xEE) {
            if (seg_end - seg >= 12 && std::memcmp(seg, "Adobe", 5) == 0) adobe_transform = seg[11];
        } else if (m == 0xDA) {                                 // scan: entropy-coded data follows
            if (!have_frame || seg_end - seg < 1) return false;
            const int ns = seg[0];
            if (ns != (int)comp.size() || seg_end - seg < 1 + 2 * ns + 3) return false;    // one interleaved scan
            std::vector<char> named(comp.size(), 0);
            for (int k = 0; k < ns; k++) {
                const int cid = seg[1 + 2 * k];
                Component *c = nullptr;
                for (auto &q : comp) if (q.id == cid) c = &q;
                if (!c || named[c - comp.data()]) return false;         // unknown component, or one named twice
                named[c - comp.data()] = 1;
                c->td = seg[2 + 2 * k] >> 4; c->ta = seg[2 + 2 * k] & 15;
                if (c->td > 3 || c->ta > 3 || !dc[c->td].present || !ac[c->ta].present || !have_q[c->tq]) return false;
            }
            p = seg_end;
            int hmax = 1, vmax = 1;
            for (auto &c : comp) { hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax; }
            if (comp.size() == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }      // a single component is never interleaved
            const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
            const int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
            for (auto &c : comp) {
                c.width = (width * c.h + hmax - 1) / hmax;
                c.height = (height * c.v + vmax - 1) / vmax;
                c.stride = mcus_x * c.h * 8;
                c.rows = mcus_y * c.v * 8;
                c.plane.assign((size_t)c.stride * c.rows, 0);
                c.pred = 0;
            }
            BitReader br(p, end);
            int until_restart = restart_interval;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    if (restart_interval && until_restart == 0) {
                        if (!br.restart()) return false;
                        for (auto &c : comp) c.pred = 0;
                        until_restart = restart_interval;
                    }
                    for (auto &c : comp)
                        for (int by = 0; by < c.v; by++)
                            for (int bx = 0; bx < c.h; bx++) {
                                int32_t coef[64] = {0};
                                int s = decode_symbol(br, dc[c.td]);
                                if (s < 0 || s > 15) return false;
                                if (s) c.pred += extend(br.bits(s), s);
                                coef[0] = c.pred * quant[c.tq][0];
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_symbol(br, ac[c.ta]);
                                    if (rs < 0) return false;
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (r != 15) break;
                                        k += 16;
                                        continue;
                                    }
                                    k += r;
                                    if (k > 63) return false;
                                    coef[kZigzag[k]] = extend(br.bits(sz), sz) * quant[c.tq][kZigzag[k]];
                                    k++;
                                }
                                uint8_t *dst = c.plane.data() + (size_t)((my * c.v + by) * 8) * c.stride + (size_t)(mx * c.h + bx) * 8;
                                idct_block(coef, dst, c.stride);
                            }
                    until_restart--;
                }
            // ---- upsample, convert, store BGR ----
            out.rows = height; out.cols = width;
            out.bgr.assign((size_t)width * height * 3, 0);
            std::vector<std::vector<uint8_t>> rows(comp.size(), std::vector<uint8_t>((size_t)width + 2));
            const bool ycc = comp.size() == 3 && adobe_transform != 0;
            for (int y = 0; y < height; y++) {
                for (size_t k = 0; k < comp.size(); k++)
                    upsample_row(comp[k], hmax / comp[k].h, vmax / comp[k].v, y, width, rows[k]);
                uint8_t *o = out.bgr.data() + (size_t)y * width * 3;
                for (int x = 0; x < width; x++) {
                    if (comp.size() == 1) {
                        o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = rows[0][x];
                    } else if (ycc) {
                        const int Y = rows[0][x], cb = rows[1][x] - 128, cr = rows[2][x] - 128;
                        // 16-bit fixed point: 1.402, 1.772, 0.71414, 0.34414
                        const int r = Y + ((91881 * cr + 32768) >> 16);
                        const int b = Y + ((116130 * cb + 32768) >> 16);
                        const int g = Y + ((-22554 * cb - 46802 * cr + 32768) >> 16);
                        o[3 * x] = clamp8(b); o[3 * x + 1] = clamp8(g); o[3 * x + 2] = clamp8(r);
                    } else {
                        o[3 * x] = rows[2][x]; o[3 * x + 1] = rows[1][x]; o[3 * x + 2] = rows[0][x];
                    }
                }
            }
            return true;
        }
        p = seg_end;
    }
    return false;
}
}  // namespace pth
