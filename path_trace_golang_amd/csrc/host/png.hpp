// png.hpp -- minimal PNG encoder (8-bit, truecolour with or without alpha), no external library.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pthost {
// Encodes `height` rows of `stride` bytes of RGBA8.  Like Go's image/png for an opaque *image.RGBA
// (every alpha = 255) the file is written as 8-bit RGB; otherwise as 8-bit RGBA.
std::vector<uint8_t> EncodePNG(const uint8_t *pix, int width, int height, int stride);
}  // namespace pthost
