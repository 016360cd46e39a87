#include "png.hpp"

#include <cstring>

namespace pthost {

namespace {

uint32_t crc_table[256];
bool crc_ready = false;
void crc_init() {
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}
uint32_t crc32(const uint8_t *p, size_t n, uint32_t c = 0xFFFFFFFFu) {
    if (!crc_ready) crc_init();
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c;
}
void be32(std::vector<uint8_t> &o, uint32_t v) {
    o.push_back((uint8_t)(v >> 24)); o.push_back((uint8_t)(v >> 16)); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v);
}
void chunk(std::vector<uint8_t> &o, const char *type, const std::vector<uint8_t> &data) {
    be32(o, (uint32_t)data.size());
    size_t start = o.size();
    o.insert(o.end(), type, type + 4);
    o.insert(o.end(), data.begin(), data.end());
    uint32_t c = crc32(o.data() + start, o.size() - start) ^ 0xFFFFFFFFu;
    be32(o, c);
}

}  // namespace

std::vector<uint8_t> EncodePNG(const uint8_t *pix, int width, int height, int stride) {
    bool opaque = true;
    for (int y = 0; y < height && opaque; y++)
        for (int x = 0; x < width; x++)
            if (pix[(size_t)y * stride + 4 * x + 3] != 255) { opaque = false; break; }
    const int bpp = opaque ? 3 : 4;
    // filtered scanlines (filter type 0)
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (1 + (size_t)width * bpp));
    for (int y = 0; y < height; y++) {
        raw.push_back(0);
        const uint8_t *row = pix + (size_t)y * stride;
        for (int x = 0; x < width; x++) raw.insert(raw.end(), row + 4 * x, row + 4 * x + bpp);
    }
    // zlib stream of stored (uncompressed) deflate blocks
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    size_t pos = 0;
    uint32_t a = 1, b = 0;
    do {
        size_t n = raw.size() - pos;
        if (n > 65535) n = 65535;
        bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        for (size_t i = pos; i < pos + n; i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    } while (pos < raw.size());
    be32(z, (b << 16) | a);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)width); be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(opaque ? 2 : 6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    return out;
}

}  // namespace pthost
