// Depth-frame readers and array dumps for the headless driver.  The reference app reads 16-bit PNG
// depth maps with cv::imread(path, IMREAD_ANYDEPTH) (src/apps/demo.cpp:316-329) and its field writer is commented out
// (demo.cpp:252-283); this header replaces both without OpenCV / VTK:
//   read_depth     16-bit grayscale PNG (zlib inflate + the five PNG row filters; non-interlaced), binary PGM "P5" with
//                  maxval > 255 (big-endian samples), or raw little-endian uint16 of rows*cols pixels -- sniffed by magic
//   write_npy      NumPy .npy v1.0, little-endian float32, C order
// Link with -lz.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace sobfu_amd {

namespace detail {
inline uint32_t be32(const unsigned char* p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace detail

// 8- or 16-bit grayscale, non-interlaced PNG -> uint16 samples (8-bit values are passed through unscaled).
inline bool read_depth_png(const std::vector<unsigned char>& file, int rows, int cols, std::vector<uint16_t>& out, std::string* why = nullptr) {
    auto fail = [&](const char* m) { if (why) *why = m; return false; };
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) return fail("not a PNG file");
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, color = -1, interlace = 0;
    std::vector<unsigned char> idat;
    while (pos + 12 <= file.size()) {
        const uint32_t len = detail::be32(&file[pos]);
        const char* type = (const char*) &file[pos + 4];
        if (pos + 12 + (size_t) len > file.size()) return fail("truncated chunk");
        const unsigned char* body = &file[pos + 8];
        if (detail::be32(body + len) != (uint32_t) crc32(crc32(0L, Z_NULL, 0), &file[pos + 4], len + 4)) return fail("chunk CRC mismatch");
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = detail::be32(body); h = detail::be32(body + 4); depth = body[8]; color = body[9]; interlace = body[12];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t) len;
    }
    if (color != 0 || (depth != 16 && depth != 8)) return fail("only 8/16-bit grayscale PNG depth maps are supported");
    if (interlace != 0) return fail("interlaced PNG is not supported");
    if ((int) w != cols || (int) h != rows) return fail("image size differs from the configured rows x cols");
    const size_t bpp = depth / 8, stride = (size_t) w * bpp;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf got = (uLongf) raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong) idat.size()) != Z_OK || got != raw.size()) return fail("inflate failed");
    out.assign((size_t) rows * cols, 0);
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    for (uint32_t y = 0; y < h; ++y) {
        const unsigned char* line = &raw[(stride + 1) * y];
        const int filter = line[0];
        if (filter > 4) return fail("bad row filter");
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = line[1 + i];
            int v = x;
            if (filter == 1) v = x + a;
            else if (filter == 2) v = x + b;
            else if (filter == 3) v = x + ((a + b) >> 1);
            else if (filter == 4) v = x + detail::paeth(a, b, c);
            cur[i] = (unsigned char) v;
        }
        for (uint32_t x = 0; x < w; ++x)
            out[(size_t) y * cols + x] = bpp == 2 ? (uint16_t) ((cur[2 * x] << 8) | cur[2 * x + 1]) : (uint16_t) cur[x];
        prev.swap(cur);
    }
    return true;
}

inline bool read_depth(const std::string& path, int rows, int cols, std::vector<uint16_t>& out, std::string* why = nullptr) {
    auto fail = [&](const char* m) { if (why) *why = m; return false; };
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return fail("cannot open file");
    std::vector<unsigned char> file;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    const size_t npix = (size_t) rows * cols;
    if (file.size() >= 8 && file[0] == 0x89 && file[1] == 'P' && file[2] == 'N' && file[3] == 'G') return read_depth_png(file, rows, cols, out, why);
    if (file.size() >= 2 && file[0] == 'P' && file[1] == '5') {
        // header: "P5" <ws> width <ws> height <ws> maxval <one ws byte>, '#' comments run to the end of their line.  Parsed by
        // hand with bounds checks (the file buffer is not NUL-terminated).
        size_t pos = 2;
        int field[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k) {
            for (;;) {  // whitespace and comments before the number
                while (pos < file.size() && std::isspace(file[pos])) ++pos;
                if (pos < file.size() && file[pos] == '#') {
                    while (pos < file.size() && file[pos] != '\n') ++pos;
                    continue;
                }
                break;
            }
            size_t digits = 0;
            long v = 0;
            while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9' && digits < 9) {
                v = v * 10 + (file[pos] - '0');
                ++pos;
                ++digits;
            }
            if (digits == 0 || (pos < file.size() && file[pos] >= '0' && file[pos] <= '9')) return fail("bad PGM header");
            field[k] = (int) v;
        }
        if (pos >= file.size() || !std::isspace(file[pos])) return fail("bad PGM header");
        const int w = field[0], h = field[1], maxv = field[2];
        const size_t start = pos + 1;  // one whitespace byte after maxval
        if (w != cols || h != rows || maxv <= 255 || file.size() < start + 2 * npix) return fail("PGM is not a rows x cols 16-bit image");
        out.resize(npix);
        for (size_t i = 0; i < npix; ++i) out[i] = (uint16_t) ((file[start + 2 * i] << 8) | file[start + 2 * i + 1]);
        return true;
    }
    if (file.size() != 2 * npix) return fail("raw depth file is not rows*cols uint16 pixels");
    out.resize(npix);
    std::memcpy(out.data(), file.data(), 2 * npix);  // little-endian host
    return true;
}

// float32 array, C order, shape given outermost first (e.g. {Z, Y, X, 4}).
inline bool write_npy(const std::string& path, const float* data, const std::vector<size_t>& shape) {
    std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (";
    size_t n = 1;
    for (size_t i = 0; i < shape.size(); ++i) {
        dict += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
        if (i + 1 < shape.size()) dict += " ";
        n *= shape[i];
    }
    dict += "), }";
    while ((10 + dict.size() + 1) % 64 != 0) dict += ' ';
    dict += '\n';
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const unsigned char head[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char) (dict.size() & 255), (unsigned char) (dict.size() >> 8)};
    bool ok = std::fwrite(head, 1, 10, f) == 10 && std::fwrite(dict.data(), 1, dict.size(), f) == dict.size() &&
              std::fwrite(data, sizeof(float), n, f) == n;
    return std::fclose(f) == 0 && ok;
}

}  // namespace sobfu_amd
