// PNG decoding for the scene reader.  Some of the reference's texture files carry
// PNG data under a .jpg name (all of scenes/sponza); cv::imread, which the reader
// uses (util/reader.cpp:67-69), goes by content, and with IMREAD_COLOR it drops
// the alpha channel and returns BGR.
//
// Supported: 8 bits per sample, non-interlaced, colour types grey, grey + alpha,
// RGB, RGBA and palette; zlib streams with stored, fixed and dynamic Huffman
// blocks.  Anything else is refused (the caller then asks for a decoded .ppm).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "scene.hpp"

namespace pth {
namespace {

struct Bits {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool bad = false;
    Bits(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    uint32_t get(int k) {                                   // k <= 16, LSB first
        while (n < k) {
            if (p >= end) { bad = true; return 0; }
            acc |= (uint32_t)*p++ << n;
            n += 8;
        }
        uint32_t v = acc & ((1u << k) - 1u);
        acc >>= k; n -= k;
        return v;
    }
    // to the next byte boundary: drop the rest of the current byte and hand back the whole bytes that get() / peek()
    // have already pulled into the accumulator
    void align() { n -= n % 8; p -= n / 8; acc = 0; n = 0; }
    // the next k bits without consuming them (zero-padded past the end); k <= 16
    uint32_t peek(int k) {
        while (n < k && p < end) { acc |= (uint32_t)*p++ << n; n += 8; }
        return acc & ((1u << k) - 1u);
    }
    void skip(int k) { acc >>= k; n -= k; }
};

// Canonical Huffman code over `count` symbols with the given code lengths (deflate, RFC 1951 section 3.2.2).
struct Code {
    static constexpr int FAST = 10;        // codes of up to FAST bits are looked up in one step
    uint16_t first[16], offset[16], nlen[16];
    std::vector<uint16_t> symbols;
    uint16_t fast[1 << FAST];              // symbol << 4 | length, indexed by the next FAST bits of the stream; 0: longer code
    bool build(const uint8_t *lengths, int count) {
        std::memset(nlen, 0, sizeof nlen);
        for (int i = 0; i < count; i++) nlen[lengths[i]]++;
        nlen[0] = 0;
        uint32_t code = 0, total = 0;
        for (int len = 1; len < 16; len++) {
            code = (code + nlen[len - 1]) << 1;
            first[len] = (uint16_t)code;
            offset[len] = (uint16_t)total;
            total += nlen[len];
            if (code + nlen[len] > (1u << len)) return false;          // over-subscribed
        }
        symbols.assign(total, 0);
        uint16_t next[16];
        std::memcpy(next, offset, sizeof next);
        std::memset(fast, 0, sizeof fast);
        uint16_t nextcode[16];
        std::memcpy(nextcode, first, sizeof nextcode);
        for (int i = 0; i < count; i++) {
            const int len = lengths[i];
            if (!len) continue;
            symbols[next[len]++] = (uint16_t)i;
            const uint32_t c = nextcode[len]++;
            if (len <= FAST) {
                uint32_t rev = 0;                                     // the stream carries a code most significant bit first
                for (int k = 0; k < len; k++) rev |= ((c >> k) & 1u) << (len - 1 - k);
                for (uint32_t fill = rev; fill < (1u << FAST); fill += 1u << len) fast[fill] = (uint16_t)((i << 4) | len);
            }
        }
        return true;
    }
    int decode(Bits &b) const {
        const uint16_t f = fast[b.peek(FAST)];
        if (f) {
            if ((f & 15) > b.n) { b.bad = true; return -1; }
            b.skip(f & 15);
            return f >> 4;
        }
        uint32_t code = 0;
        for (int len = 1; len < 16; len++) {
            code = (code << 1) | b.get(1);
            if (b.bad) return -1;
            if (nlen[len] && code >= first[len] && code - first[len] < nlen[len]) return symbols[offset[len] + code - first[len]];
        }
        return -1;
    }
};

bool inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t expect) {
    if (n < 6) return false;
    if ((src[0] & 0x0F) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return false;     // zlib header
    Bits b(src + 2, src + n);
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    out.clear();
    out.reserve(expect);
    for (;;) {
        const uint32_t last = b.get(1), type = b.get(2);
        if (b.bad || type == 3) return false;
        if (type == 0) {
            b.align();
            if (b.end - b.p < 4) return false;
            const uint32_t len = b.p[0] | (b.p[1] << 8), nlen = b.p[2] | (b.p[3] << 8);
            b.p += 4;
            if ((len ^ nlen) != 0xFFFFu || (size_t)(b.end - b.p) < len) return false;
            out.insert(out.end(), b.p, b.p + len);
            b.p += len;
        } else {
            Code lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 288; i++) lengths[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
                if (!lit.build(lengths, 288)) return false;
                for (int i = 0; i < 30; i++) lengths[i] = 5;
                if (!dist.build(lengths, 30)) return false;
            } else {
                const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
                if (b.bad || hlit > 286 || hdist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < hclen; i++) cl[order[i]] = (uint8_t)b.get(3);
                Code clc;
                if (b.bad || !clc.build(cl, 19)) return false;
                int i = 0;
                while (i < hlit + hdist) {
                    const int sym = clc.decode(b);
                    if (sym < 0) return false;
                    if (sym < 16) { lengths[i++] = (uint8_t)sym; continue; }
                    int rep, val = 0;
                    if (sym == 16) { if (i == 0) return false; val = lengths[i - 1]; rep = 3 + (int)b.get(2); }
                    else if (sym == 17) rep = 3 + (int)b.get(3);
                    else rep = 11 + (int)b.get(7);
                    if (b.bad || i + rep > hlit + hdist) return false;
                    while (rep--) lengths[i++] = (uint8_t)val;
                }
                if (lengths[256] == 0 || !lit.build(lengths, hlit) || !dist.build(lengths + hlit, hdist)) return false;
            }
            for (;;) {
                const int sym = lit.decode(b);
                if (sym < 0) return false;
                if (out.size() > expect + 64) return false;            // checked while expanding, not only per block
                if (sym < 256) { out.push_back((uint8_t)sym); continue; }
                if (sym == 256) break;
                if (sym > 285) return false;
                const uint32_t len = len_base[sym - 257] + b.get(len_extra[sym - 257]);
                const int ds = dist.decode(b);
                if (ds < 0 || ds > 29) return false;
                const uint32_t d = dist_base[ds] + b.get(dist_extra[ds]);
                if (b.bad || d > out.size()) return false;
                const size_t from = out.size() - d;
                for (uint32_t k = 0; k < len; k++) out.push_back(out[from + k]);
            }
        }
        if (out.size() > expect + 64) return false;
        if (last) break;
    }
    return true;
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

}  // namespace

bool decode_png_file(const std::string &path, Texture &out) {
    std::vector<uint8_t> buf;
    {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END);
        long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (n <= 8) { std::fclose(f); return false; }
        buf.resize((size_t)n);
        size_t got = std::fread(buf.data(), 1, (size_t)n, f);
        std::fclose(f);
        if (got != (size_t)n) return false;
    }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    if (std::memcmp(buf.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    uint32_t width = 0, height = 0;
    int ctype = -1;
    std::vector<uint8_t> idat, palette;
    bool end = false;
    while (!end && pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        const uint8_t *type = &buf[pos + 4], *data = &buf[pos + 8];
        if (len > buf.size() || pos + 12 + len > buf.size()) return false;
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return false;
            width = be32(data); height = be32(data + 4);
            ctype = data[9];
            if (data[8] != 8 || data[10] != 0 || data[11] != 0 || data[12] != 0) return false;       // 8 bits, no interlace
            if (width == 0 || height == 0 || width > 32768 || height > 32768) return false;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            end = true;
        }
        pos += 12 + (size_t)len;
    }
    int ch;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 3: ch = 1; if (palette.size() < 3) return false; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: return false;
    }
    const size_t stride = (size_t)width * ch, expect = (stride + 1) * height;
    std::vector<uint8_t> raw;
    if (!inflate(idat.data(), idat.size(), raw, expect) || raw.size() < expect) return false;
    // undo the per-row filters in place (PNG specification, section 9)
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < height; y++) {
        uint8_t *row = &raw[(stride + 1) * y + 1];
        const uint8_t *up = y ? &raw[(stride + 1) * (y - 1) + 1] : zero.data();
        const int filter = raw[(stride + 1) * y];
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= (size_t)ch ? row[x - ch] : 0, b = up[x], c = x >= (size_t)ch ? up[x - ch] : 0;
            int pred;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return false;
            }
            row[x] = (uint8_t)(row[x] + pred);
        }
    }
    out.rows = (int)height; out.cols = (int)width;
    out.bgr.assign((size_t)width * height * 3, 0);
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t *row = &raw[(stride + 1) * y + 1];
        uint8_t *o = &out.bgr[(size_t)y * width * 3];
        for (uint32_t x = 0; x < width; x++) {
            uint8_t r, g, b;
            if (ctype == 0 || ctype == 4) { r = g = b = row[(size_t)x * ch]; }
            else if (ctype == 3) {
                const size_t k = (size_t)row[x] * 3;
                if (k + 2 >= palette.size()) return false;
                r = palette[k]; g = palette[k + 1]; b = palette[k + 2];
            } else { r = row[(size_t)x * ch]; g = row[(size_t)x * ch + 1]; b = row[(size_t)x * ch + 2]; }
            o[3 * x] = b; o[3 * x + 1] = g; o[3 * x + 2] = r;
        }
    }
    return true;
}

bool decode_image_file(const std::string &path, Texture &out) {
    return decode_jpeg_file(path, out) || decode_png_file(path, out);
}
}  // namespace pth
