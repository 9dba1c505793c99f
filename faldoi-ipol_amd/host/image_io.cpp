#include "image_io.hpp"

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace faldoi_host {
namespace {

std::vector<unsigned char> slurp(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return buf;
}

uint32_t be32(const unsigned char *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

Image decode_png(const std::vector<unsigned char> &buf, const std::string &path)
{
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) throw std::runtime_error(path + ": not a PNG");
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        const char *type = (const char *)&buf[pos + 4];
        if (pos + 12 + (size_t)len > buf.size()) throw std::runtime_error(path + ": truncated PNG chunk");
        const unsigned char *d = &buf[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) throw std::runtime_error(path + ": bad IHDR");
            w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(d, d + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + len);
        } else if (!memcmp(type, "IEND", 4)) {
            end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!w || !h || w > (1u << 15) || h > (1u << 15)) throw std::runtime_error(path + ": bad PNG size");
    if (interlace) throw std::runtime_error(path + ": interlaced PNG is not supported");
    if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16)
        throw std::runtime_error(path + ": bad PNG bit depth");
    int ch = 0;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 3: ch = 1; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: throw std::runtime_error(path + ": bad PNG colour type");
    }
    if (depth < 8 && ctype != 0 && ctype != 3) throw std::runtime_error(path + ": bad PNG depth/colour combination");
    if (ctype == 3 && depth == 16) throw std::runtime_error(path + ": bad PNG depth/colour combination");
    const size_t bpp = depth < 8 ? 1 : (size_t)ch * depth / 8;              // filter distance in bytes
    const size_t stride = ((size_t)w * ch * depth + 7) / 8;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || rawlen != raw.size()) throw std::runtime_error(path + ": PNG inflate failed");
    std::vector<unsigned char> pix(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const unsigned char *in = &raw[(stride + 1) * y];
        unsigned char *out = &pix[stride * y];
        const unsigned char *up = y ? &pix[stride * (y - 1)] : nullptr;
        const int ft = in[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = in[1 + i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += paeth(a, b, c); break;
                default: throw std::runtime_error(path + ": bad PNG filter");
            }
            out[i] = (unsigned char)v;
        }
    }
    if (depth < 8) {                                   // unpack 1/2/4-bit samples (MSB first) to bytes
        std::vector<unsigned char> wide((size_t)w * h);
        const int per = 8 / depth, mask = (1 << depth) - 1;
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                const unsigned char byte = pix[stride * y + x / per];
                wide[(size_t)y * w + x] = (unsigned char)((byte >> (depth * (per - 1 - (int)(x % per)))) & mask);
            }
        pix.swap(wide);
        depth = 8;
    }
    const size_t obpp = (size_t)ch * depth / 8;        // bytes per pixel of the unpacked samples
    Image im;
    im.w = (int)w; im.h = (int)h;
    const size_t n = (size_t)w * h;
    if (ctype == 3) {                                  // palette -> RGB
        if (plte.size() < 3) throw std::runtime_error(path + ": palette PNG without PLTE");
        im.pd = 3;
        im.data.resize(3 * n);
        for (size_t i = 0; i < n; ++i) {
            const size_t k = 3 * (size_t)pix[i];
            for (int c = 0; c < 3; ++c) im.data[c * n + i] = k + c < plte.size() ? plte[k + c] : 0;
        }
        return im;
    }
    im.pd = ch;
    im.data.resize((size_t)ch * n);
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < ch; ++c) {
            const unsigned char *s = &pix[i * obpp + (size_t)c * depth / 8];
            im.data[c * n + i] = depth == 8 ? (float)s[0] : (float)((s[0] << 8) | s[1]);
        }
    return im;
}

Image decode_pnm(const std::vector<unsigned char> &buf, const std::string &path)
{
    size_t pos = 2;
    auto next_int = [&]() {
        for (;;) {
            while (pos < buf.size() && isspace(buf[pos])) ++pos;
            if (pos < buf.size() && buf[pos] == '#') { while (pos < buf.size() && buf[pos] != '\n') ++pos; continue; }
            break;
        }
        long v = 0; bool any = false;
        while (pos < buf.size() && isdigit(buf[pos])) { v = v * 10 + (buf[pos++] - '0'); any = true; }
        if (!any) throw std::runtime_error(path + ": bad PNM header");
        return v;
    };
    const int ch = buf[1] == '5' ? 1 : 3;
    const long w = next_int(), h = next_int(), maxv = next_int();
    ++pos;                                             // single whitespace after maxval
    if (w <= 0 || h <= 0 || maxv <= 0 || maxv > 65535) throw std::runtime_error(path + ": bad PNM header");
    const int bps = maxv > 255 ? 2 : 1;
    const size_t n = (size_t)w * h;
    if (buf.size() < pos + n * ch * bps) throw std::runtime_error(path + ": truncated PNM");
    Image im; im.w = (int)w; im.h = (int)h; im.pd = ch; im.data.resize(n * ch);
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < ch; ++c) {
            const unsigned char *s = &buf[pos + (i * ch + c) * bps];
            im.data[c * n + i] = bps == 1 ? (float)s[0] : (float)((s[0] << 8) | s[1]);
        }
    return im;
}

Image decode_flo(const std::vector<unsigned char> &buf, const std::string &path)
{
    if (buf.size() < 12 || memcmp(buf.data(), "PIEH", 4) != 0) throw std::runtime_error(path + ": not a .flo file");
    int32_t w, h;
    memcpy(&w, &buf[4], 4); memcpy(&h, &buf[8], 4);
    if (w <= 0 || h <= 0 || buf.size() != 12 + (size_t)8 * w * h)
        throw std::runtime_error(path + ": .flo header does not match the file size");
    const size_t n = (size_t)w * h;
    Image im; im.w = w; im.h = h; im.pd = 2; im.data.resize(2 * n);
    const float *p = (const float *)&buf[12];
    for (size_t i = 0; i < n; ++i) { im.data[i] = p[2 * i]; im.data[n + i] = p[2 * i + 1]; }
    return im;
}

}  // namespace

Image read_image_split(const std::string &path)
{
    const std::vector<unsigned char> buf = slurp(path);
    if (buf.size() >= 4 && !memcmp(buf.data(), "PIEH", 4)) return decode_flo(buf, path);
    if (buf.size() >= 8 && buf[0] == 0x89 && buf[1] == 'P') return decode_png(buf, path);
    if (buf.size() >= 2 && buf[0] == 'P' && (buf[1] == '5' || buf[1] == '6')) return decode_pnm(buf, path);
    throw std::runtime_error(path + ": unsupported image format (PNG, PGM/PPM, .flo)");
}

Image read_flo(const std::string &path) { return decode_flo(slurp(path), path); }

void write_flo(const std::string &path, const float *u, int w, int h)
{
    const size_t n = (size_t)w * h;
    std::vector<float> inter(2 * n);
    for (size_t i = 0; i < n; ++i) { inter[2 * i] = u[i]; inter[2 * i + 1] = u[n + i]; }
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    const int32_t wh[2] = {w, h};
    const bool ok = fwrite("PIEH", 1, 4, f) == 4 && fwrite(wh, 4, 2, f) == 2 && fwrite(inter.data(), 4, 2 * n, f) == 2 * n;
    if (fclose(f) != 0 || !ok) throw std::runtime_error("short write to " + path);
}

}  // namespace faldoi_host
