// Baseline JPEG writer for rendered frames: the reference's data sets are `cv::imwrite("<scene><k>.jpg", image)` of its
// 512 x 512 BGR frames (ml/data/<SCENE>/<SCENE>{4ssp,4096ssp,dif}/, mainOMP.cpp:242 "change to imwrite").  Sequential DCT,
// 8-bit, YCbCr 4:2:0 (what OpenCV / libjpeg write by default), the standard quantisation tables of ITU-T T.81 Annex K
// scaled by a quality setting the way libjpeg does, and the standard Huffman tables of Annex K.3 -- no optimisation pass,
// no restart markers.  Any JPEG reader takes the result; this build's own decoder is one of the checks.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "scene.hpp"

namespace pth {
namespace {

const uint8_t kZig[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                          41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                          30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
// T.81 Table K.1 / K.2 (natural order)
const uint8_t kLumaQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                            69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55, 64,
                            81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                              99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// T.81 Tables K.3 - K.6: code counts per length 1..16, then the symbols in code order
const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};

struct HuffEnc {
    uint16_t code[256];
    uint8_t len[256];
    void build(const uint8_t *bits, const uint8_t *vals) {          // canonical codes, T.81 Annex C
        std::memset(len, 0, sizeof len);
        uint32_t c = 0;
        int k = 0;
        for (int l = 1; l <= 16; l++) {
            for (int i = 0; i < bits[l - 1]; i++, k++) { code[vals[k]] = (uint16_t)c++; len[vals[k]] = (uint8_t)l; }
            c <<= 1;
        }
    }
};

struct BitWriter {
    std::vector<uint8_t> &out;
    uint32_t acc = 0;
    int n = 0;
    explicit BitWriter(std::vector<uint8_t> &o) : out(o) {}
    void put(uint32_t bits, int len) {                              // len <= 16, most significant bit first
        acc = (acc << len) | (bits & ((1u << len) - 1u));
        n += len;
        while (n >= 8) {
            const uint8_t b = (uint8_t)(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xFF) out.push_back(0);                        // byte stuffing
            n -= 8;
        }
    }
    void flush() { if (n) put(0x7F, 8 - n); }                       // pad with ones
};

// Forward 8x8 DCT (separable, double precision: a frame is 4 096 blocks, speed is no concern) of level-shifted samples.
void fdct(const float *in, float *out) {
    // filled by the initialiser of a function-local static: once, and complete before any thread reads it (frames are
    // written from several threads)
    struct Table { double c[8][8]; };
    static const Table table = [] {
        Table t;
        for (int u = 0; u < 8; u++)
            for (int x = 0; x < 8; x++) t.c[u][x] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        return t;
    }();
    const double (&c)[8][8] = table.c;
    double tmp[64];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++) {
            double s = 0;
            for (int x = 0; x < 8; x++) s += c[u][x] * in[8 * y + x];
            tmp[8 * y + u] = s;
        }
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) {
            double s = 0;
            for (int y = 0; y < 8; y++) s += c[v][y] * tmp[8 * y + u];
            out[8 * v + u] = (float)s;
        }
}

void put16(std::vector<uint8_t> &o, int v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }

void encode_block(BitWriter &bw, const float *samples, const uint16_t *q, int &pred, const HuffEnc &dc, const HuffEnc &ac) {
    float coef[64];
    fdct(samples, coef);
    int zz[64];
    for (int k = 0; k < 64; k++) zz[k] = (int)std::lround(coef[kZig[k]] / (float)q[kZig[k]]);
    auto magnitude = [](int v, int &size, uint32_t &bits) {
        int a = v < 0 ? -v : v;
        size = 0;
        while (a) { size++; a >>= 1; }
        bits = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << size) - 1u);
    };
    int size; uint32_t bits;
    magnitude(zz[0] - pred, size, bits);
    pred = zz[0];
    bw.put(dc.code[size], dc.len[size]);
    if (size) bw.put(bits, size);
    int run = 0;
    for (int k = 1; k < 64; k++) {
        if (zz[k] == 0) { run++; continue; }
        while (run > 15) { bw.put(ac.code[0xF0], ac.len[0xF0]); run -= 16; }
        magnitude(zz[k], size, bits);
        bw.put(ac.code[(run << 4) | size], ac.len[(run << 4) | size]);
        bw.put(bits, size);
        run = 0;
    }
    if (run) bw.put(ac.code[0], ac.len[0]);                         // end of block
}

}  // namespace

bool encode_jpeg(const uint8_t *bgr, int rows, int cols, int quality, std::vector<uint8_t> &out) {
    if (!bgr || rows <= 0 || cols <= 0 || rows > 65535 || cols > 65535) return false;
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;      // libjpeg's jpeg_quality_scaling
    uint16_t q[2][64];
    for (int k = 0; k < 64; k++) {
        int a = ((int)kLumaQ[k] * scale + 50) / 100, b = ((int)kChromaQ[k] * scale + 50) / 100;
        q[0][k] = (uint16_t)(a < 1 ? 1 : (a > 255 ? 255 : a));
        q[1][k] = (uint16_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
    out.clear();
    out.push_back(0xFF); out.push_back(0xD8);
    static const uint8_t jfif[18] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    out.insert(out.end(), jfif, jfif + 18);
    for (int t = 0; t < 2; t++) {
        out.push_back(0xFF); out.push_back(0xDB); put16(out, 67); out.push_back((uint8_t)t);
        for (int k = 0; k < 64; k++) out.push_back((uint8_t)q[t][kZig[k]]);
    }
    out.push_back(0xFF); out.push_back(0xC0); put16(out, 17); out.push_back(8); put16(out, rows); put16(out, cols); out.push_back(3);
    const uint8_t comps[9] = {1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1};   // Y 2x2, Cb and Cr 1x1: 4:2:0
    out.insert(out.end(), comps, comps + 9);
    auto dht = [&](int cls_id, const uint8_t *bits, const uint8_t *vals, int n) {
        out.push_back(0xFF); out.push_back(0xC4); put16(out, 19 + n); out.push_back((uint8_t)cls_id);
        out.insert(out.end(), bits, bits + 16);
        out.insert(out.end(), vals, vals + n);
    };
    dht(0x00, kDcLumaBits, kDcVals, 12); dht(0x10, kAcLumaBits, kAcLumaVals, 162);
    dht(0x01, kDcChromaBits, kDcVals, 12); dht(0x11, kAcChromaBits, kAcChromaVals, 162);
    static const uint8_t sos[14] = {0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
    out.insert(out.end(), sos, sos + 14);

    HuffEnc dcl, acl, dcc, acc_;
    dcl.build(kDcLumaBits, kDcVals); acl.build(kAcLumaBits, kAcLumaVals);
    dcc.build(kDcChromaBits, kDcVals); acc_.build(kAcChromaBits, kAcChromaVals);
    BitWriter bw(out);
    int pred[3] = {0, 0, 0};
    const int mcus_x = (cols + 15) / 16, mcus_y = (rows + 15) / 16;
    float Y[16][16], Cb[16][16], Cr[16][16], blk[64];
    for (int my = 0; my < mcus_y; my++)
        for (int mx = 0; mx < mcus_x; mx++) {
            for (int y = 0; y < 16; y++)
                for (int x = 0; x < 16; x++) {                      // edge pixels are repeated past the image
                    const int r = my * 16 + y < rows ? my * 16 + y : rows - 1, c = mx * 16 + x < cols ? mx * 16 + x : cols - 1;
                    const uint8_t *p = bgr + ((size_t)r * cols + c) * 3;
                    const float B = p[0], G = p[1], R = p[2];
                    Y[y][x] = 0.299f * R + 0.587f * G + 0.114f * B - 128.0f;
                    Cb[y][x] = -0.168736f * R - 0.331264f * G + 0.5f * B;
                    Cr[y][x] = 0.5f * R - 0.418688f * G - 0.081312f * B;
                }
            for (int by = 0; by < 2; by++)
                for (int bx = 0; bx < 2; bx++) {
                    for (int y = 0; y < 8; y++)
                        for (int x = 0; x < 8; x++) blk[8 * y + x] = Y[8 * by + y][8 * bx + x];
                    encode_block(bw, blk, q[0], pred[0], dcl, acl);
                }
            for (int ch = 0; ch < 2; ch++) {
                float (*C)[16] = ch ? Cr : Cb;
                for (int y = 0; y < 8; y++)
                    for (int x = 0; x < 8; x++)
                        blk[8 * y + x] = 0.25f * (C[2 * y][2 * x] + C[2 * y][2 * x + 1] + C[2 * y + 1][2 * x] + C[2 * y + 1][2 * x + 1]);
                encode_block(bw, blk, q[1], pred[1 + ch], dcc, acc_);
            }
        }
    bw.flush();
    out.push_back(0xFF); out.push_back(0xD9);
    return true;
}

bool write_jpeg_file(const std::string &path, const uint8_t *bgr, int rows, int cols, int quality) {
    std::vector<uint8_t> buf;
    if (!encode_jpeg(bgr, rows, cols, quality, buf)) return false;
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    return (std::fclose(f) == 0) && ok;
}

}  // namespace pth
