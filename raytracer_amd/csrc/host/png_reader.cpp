// PNG decoding for texture files (the reference reads them through its vendored lodepng,
// src/texture.cpp:77-87; this is an independent decoder written against the PNG and DEFLATE specifications).
#include "png_reader.h"

#include <cstdio>
#include <cstring>

namespace rt {
namespace {

// ---- DEFLATE (RFC 1951) ----
struct Bits {
    const uint8_t *p;
    size_t n, pos = 0;
    uint32_t hold = 0;
    int have = 0;
    bool overrun = false;

    uint32_t take(int need)
    {
        while (have < need) {
            if (pos >= n) { overrun = true; return 0; }
            hold |= (uint32_t)p[pos++] << have;
            have += 8;
        }
        const uint32_t v = hold & ((1u << need) - 1u);
        hold >>= need;
        have -= need;
        return v;
    }
    void to_byte() { hold = 0; have = 0; }
};

// canonical Huffman code given as the number of codes of each length and the symbols in code order
struct Code {
    uint16_t per_length[16];
    uint16_t symbols[288];

    bool build(const uint8_t *lengths, int n)
    {
        memset(per_length, 0, sizeof per_length);
        for (int i = 0; i < n; ++i) ++per_length[lengths[i]];
        int left = 1; // codes still free at this length; negative = over-subscribed
        for (int l = 1; l < 16; ++l) {
            left = left * 2 - per_length[l];
            if (left < 0) return false;
        }
        uint16_t first[16];
        first[1] = 0;
        for (int l = 1; l < 15; ++l) first[l + 1] = (uint16_t)(first[l] + per_length[l]);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbols[first[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(Bits &b) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= (int)b.take(1);
            if (b.overrun) return -1;
            const int count = per_length[l];
            if (code - count < first) return symbols[index + (code - first)];
            index += count;
            first = (first + count) << 1;
            code <<= 1;
        }
        return -1;
    }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// `limit`: the most the stream may expand to (a small file can expand a thousandfold: the caller knows what it expects)
bool inflate_block(Bits &b, const Code &lit, const Code &dist, std::vector<uint8_t> &out, size_t limit)
{
    for (;;) {
        const int s = lit.decode(b);
        if (s < 0) return false;
        if (s < 256) { if (out.size() >= limit) return false; out.push_back((uint8_t)s); continue; }
        if (s == 256) return true;
        if (s > 285) return false;
        const size_t len = kLenBase[s - 257] + b.take(kLenExtra[s - 257]);
        const int ds = dist.decode(b);
        if (ds < 0 || ds > 29) return false;
        const size_t back = kDistBase[ds] + b.take(kDistExtra[ds]);
        if (b.overrun || back > out.size() || len > limit - out.size()) return false;
        const size_t from = out.size() - back;
        for (size_t i = 0; i < len; ++i) out.push_back(out[from + i]); // may overlap what it writes
    }
}

bool inflate(Bits &b, std::vector<uint8_t> &out, size_t limit)
{
    for (;;) {
        const uint32_t last = b.take(1), type = b.take(2);
        if (b.overrun) return false;
        if (type == 0) {
            b.to_byte();
            if (b.pos + 4 > b.n) return false;
            const uint32_t len = b.p[b.pos] | (b.p[b.pos + 1] << 8), nlen = b.p[b.pos + 2] | (b.p[b.pos + 3] << 8);
            b.pos += 4;
            if ((len ^ 0xFFFFu) != nlen || b.pos + len > b.n || len > limit - out.size()) return false;
            out.insert(out.end(), b.p + b.pos, b.p + b.pos + len);
            b.pos += len;
        } else if (type == 1) {
            uint8_t l[288 + 30];
            for (int i = 0; i < 144; ++i) l[i] = 8;
            for (int i = 144; i < 256; ++i) l[i] = 9;
            for (int i = 256; i < 280; ++i) l[i] = 7;
            for (int i = 280; i < 288; ++i) l[i] = 8;
            for (int i = 0; i < 30; ++i) l[288 + i] = 5;
            Code lit, dist;
            lit.build(l, 288);
            dist.build(l + 288, 30);
            if (!inflate_block(b, lit, dist, out, limit)) return false;
        } else if (type == 2) {
            const int nlit = (int)b.take(5) + 257, ndist = (int)b.take(5) + 1, ncode = (int)b.take(4) + 4;
            if (b.overrun || nlit > 286 || ndist > 30) return false;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)b.take(3);
            Code meta;
            if (!meta.build(cl, 19)) return false;
            uint8_t l[286 + 30];
            int i = 0;
            while (i < nlit + ndist) {
                const int s = meta.decode(b);
                if (s < 0) return false;
                if (s < 16) { l[i++] = (uint8_t)s; continue; }
                uint8_t fill = 0;
                int rep;
                if (s == 16) { if (i == 0) return false; fill = l[i - 1]; rep = 3 + (int)b.take(2); }
                else if (s == 17) rep = 3 + (int)b.take(3);
                else rep = 11 + (int)b.take(7);
                if (b.overrun || i + rep > nlit + ndist) return false;
                while (rep--) l[i++] = fill;
            }
            if (l[256] == 0) return false; // no end-of-block code
            Code lit, dist;
            if (!lit.build(l, nlit)) return false;
            if (!dist.build(l + nlit, ndist)) return false;
            if (!inflate_block(b, lit, dist, out, limit)) return false;
        } else return false;
        if (last) return !b.overrun;
    }
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct CrcTable {
    uint32_t v[256];
    CrcTable()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            v[i] = c;
        }
    }
};

uint32_t crc32(const uint8_t *p, size_t n)
{
    static const CrcTable built; // initialised once, thread-safe (a function-local static)
    const uint32_t *table = built.v;
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// one (sub-)image of `rows` scanlines of `line` bytes, each preceded by its filter type; undone in place -> packed rows
bool unfilter(uint8_t *data, size_t rows, size_t line, size_t bpp, uint8_t *out)
{
    const uint8_t *prev = nullptr;
    for (size_t y = 0; y < rows; ++y) {
        const uint8_t type = data[y * (line + 1)];
        const uint8_t *in = data + y * (line + 1) + 1;
        uint8_t *o = out + y * line;
        for (size_t x = 0; x < line; ++x) {
            const int a = x >= bpp ? o[x - bpp] : 0, b = prev ? prev[x] : 0, c = (prev && x >= bpp) ? prev[x - bpp] : 0;
            int v;
            switch (type) {
            case 0: v = in[x]; break;
            case 1: v = in[x] + a; break;
            case 2: v = in[x] + b; break;
            case 3: v = in[x] + ((a + b) >> 1); break;
            case 4: v = in[x] + paeth(a, b, c); break;
            default: return false;
            }
            o[x] = (uint8_t)v;
        }
        prev = o;
    }
    return true;
}

struct Header { uint32_t w, h; int depth, ctype, interlace; };

int channels_of(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }

// pixel x of a packed row -> 8-bit RGB
void pixel_rgb(const Header &H, const uint8_t *row, size_t x, const uint8_t *palette, size_t palette_n, uint8_t *rgb)
{
    const int ch = channels_of(H.ctype);
    uint32_t s[4] = {0, 0, 0, 0};
    if (H.depth == 8) for (int c = 0; c < ch; ++c) s[c] = row[x * ch + c];
    else if (H.depth == 16) for (int c = 0; c < ch; ++c) s[c] = row[(x * ch + c) * 2]; // the high byte
    else { // 1, 2, 4 bits: one channel, first pixel in the high bits
        const size_t bit = x * H.depth;
        s[0] = (row[bit >> 3] >> (8 - H.depth - (bit & 7))) & ((1u << H.depth) - 1u);
        if (H.ctype == 0) s[0] = s[0] * 255u / ((1u << H.depth) - 1u);
    }
    if (H.ctype == 3) {
        if (s[0] < palette_n) { rgb[0] = palette[s[0] * 3]; rgb[1] = palette[s[0] * 3 + 1]; rgb[2] = palette[s[0] * 3 + 2]; }
        else rgb[0] = rgb[1] = rgb[2] = 0; // an index past the palette shows black
    } else if (H.ctype == 0 || H.ctype == 4) rgb[0] = rgb[1] = rgb[2] = (uint8_t)s[0];
    else { rgb[0] = (uint8_t)s[0]; rgb[1] = (uint8_t)s[1]; rgb[2] = (uint8_t)s[2]; }
}

} // namespace

bool zlib_inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &out, std::string &err, size_t limit)
{
    if (n < 6) { err = "zlib stream too short"; return false; }
    if ((src[0] & 0x0F) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) { err = "bad zlib header"; return false; }
    Bits b{src + 2, n - 2};
    out.clear();
    if (!inflate(b, out, limit)) { err = out.size() >= limit ? "deflate data expands beyond what the image needs" : "corrupt deflate data"; return false; }
    b.to_byte();
    if (b.pos + 4 > b.n) { err = "zlib stream ends before its checksum"; return false; }
    uint32_t s1 = 1, s2 = 0;
    for (size_t i = 0; i < out.size();) { // Adler-32, sums reduced before they can overflow
        const size_t stop = i + 5552 < out.size() ? i + 5552 : out.size();
        for (; i < stop; ++i) { s1 += out[i]; s2 += s1; }
        s1 %= 65521u; s2 %= 65521u;
    }
    if (((s2 << 16) | s1) != be32(b.p + b.pos)) { err = "zlib checksum mismatch"; return false; }
    return true;
}

bool decode_png_rgb8(const uint8_t *file, size_t size, int &width, int &height, std::vector<uint8_t> &rgb, std::string &err)
{
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (size < 8 + 25 || memcmp(file, sig, 8) != 0) { err = "not a PNG file"; return false; }
    Header H{};
    std::vector<uint8_t> idat, palette;
    bool have_header = false, ended = false;
    size_t pos = 8;
    while (!ended) {
        if (pos + 12 > size) { err = "PNG file ends inside a chunk"; return false; }
        const uint32_t len = be32(file + pos);
        if (len > 0x7FFFFFFFu || pos + 12 + (size_t)len > size) { err = "PNG chunk longer than the file"; return false; }
        const uint8_t *type = file + pos + 4, *body = file + pos + 8;
        if (crc32(type, 4 + (size_t)len) != be32(body + len)) { err = "PNG chunk checksum mismatch"; return false; }
        if (!have_header && memcmp(type, "IHDR", 4) != 0) { err = "PNG file does not start with its header chunk"; return false; }
        if (memcmp(type, "IHDR", 4) == 0) {
            if (len != 13) { err = "bad PNG header"; return false; }
            H.w = be32(body); H.h = be32(body + 4); H.depth = body[8]; H.ctype = body[9]; H.interlace = body[12];
            const bool depth_ok = H.ctype == 0 ? (H.depth == 1 || H.depth == 2 || H.depth == 4 || H.depth == 8 || H.depth == 16)
                                : H.ctype == 3 ? (H.depth == 1 || H.depth == 2 || H.depth == 4 || H.depth == 8)
                                : (H.ctype == 2 || H.ctype == 4 || H.ctype == 6) ? (H.depth == 8 || H.depth == 16) : false;
            if (!depth_ok || body[10] != 0 || body[11] != 0 || H.interlace > 1 || H.w == 0 || H.h == 0 || (uint64_t)H.w * H.h > (1u << 28)) {
                err = "unsupported PNG header (size, colour type, bit depth, compression, filter or interlace method)";
                return false;
            }
            have_header = true;
        } else if (memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768) { err = "bad PNG palette"; return false; }
            palette.assign(body, body + len);
        } else if (memcmp(type, "IDAT", 4) == 0) idat.insert(idat.end(), body, body + len);
        else if (memcmp(type, "IEND", 4) == 0) ended = true;
        else if (!(type[0] & 0x20)) { err = "unknown critical PNG chunk"; return false; }
        pos += 12 + (size_t)len;
    }
    if (H.ctype == 3 && palette.empty()) { err = "palette PNG without a palette"; return false; }
    const size_t bits = (size_t)channels_of(H.ctype) * H.depth;
    const size_t bpp = bits >= 8 ? bits / 8 : 1;
    // the passes of the image: one for a plain file, seven reduced images for an interlaced one (Adam7)
    static const uint32_t x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    // what the scanlines of all passes take, filter bytes included: the stream may not expand beyond it
    size_t expected = 0;
    for (int pass = 0; pass < (H.interlace ? 7 : 1); ++pass) {
        const uint32_t sx = H.interlace ? x0[pass] : 0, sy = H.interlace ? y0[pass] : 0, px = H.interlace ? dx[pass] : 1, py = H.interlace ? dy[pass] : 1;
        const size_t pw = H.w > sx ? (H.w - sx + px - 1) / px : 0, ph = H.h > sy ? (H.h - sy + py - 1) / py : 0;
        if (pw && ph) expected += ph * ((pw * bits + 7) / 8 + 1);
    }
    std::vector<uint8_t> raw;
    raw.reserve(expected);
    if (!zlib_inflate(idat.data(), idat.size(), raw, err, expected)) return false;

    width = (int)H.w; height = (int)H.h;
    rgb.assign((size_t)H.w * H.h * 3, 0);
    size_t at = 0;
    std::vector<uint8_t> rows;
    for (int pass = 0; pass < (H.interlace ? 7 : 1); ++pass) {
        const uint32_t sx = H.interlace ? x0[pass] : 0, sy = H.interlace ? y0[pass] : 0, px = H.interlace ? dx[pass] : 1, py = H.interlace ? dy[pass] : 1;
        const size_t pw = H.w > sx ? (H.w - sx + px - 1) / px : 0, ph = H.h > sy ? (H.h - sy + py - 1) / py : 0;
        if (pw == 0 || ph == 0) continue;
        const size_t line = (pw * bits + 7) / 8;
        if (at + ph * (line + 1) > raw.size()) { err = "PNG image data too short"; return false; }
        rows.resize(ph * line);
        if (!unfilter(raw.data() + at, ph, line, bpp, rows.data())) { err = "unknown PNG filter type"; return false; }
        at += ph * (line + 1);
        for (size_t y = 0; y < ph; ++y)
            for (size_t x = 0; x < pw; ++x)
                pixel_rgb(H, rows.data() + y * line, x, palette.data(), palette.size() / 3, &rgb[((sy + y * py) * (size_t)H.w + sx + x * px) * 3]);
    }
    return true;
}

bool load_png_rgb8(const std::string &path, int &width, int &height, std::vector<uint8_t> &rgb, std::string &err)
{
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { err = "cannot open " + path; return false; }
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fp)) > 0) file.insert(file.end(), buf, buf + got);
    fclose(fp);
    return decode_png_rgb8(file.data(), file.size(), width, height, rgb, err);
}

} // namespace rt
