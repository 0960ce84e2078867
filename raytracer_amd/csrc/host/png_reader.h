// PNG files as textures: decoded to 8-bit RGB rows, top row first - what the reference asks its
// vendored decoder for (src/texture.cpp:77-87: lodepng::decode(..., LCT_RGB), 8 bits per channel).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rt {

// Every colour type (grey, RGB, palette, grey + alpha, RGBA), every bit depth (1..16), interlaced or not.
// Alpha is dropped, 16-bit samples keep their high byte, grey samples below 8 bits are spread over 0..255.
bool decode_png_rgb8(const uint8_t *file, size_t size, int &width, int &height, std::vector<uint8_t> &rgb, std::string &err);
bool load_png_rgb8(const std::string &path, int &width, int &height, std::vector<uint8_t> &rgb, std::string &err);

// RFC 1950 / 1951: a zlib stream (header, deflate blocks, Adler-32) -> bytes; fails once the output would pass `limit`
bool zlib_inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &out, std::string &err, size_t limit = (size_t)-1);

} // namespace rt
