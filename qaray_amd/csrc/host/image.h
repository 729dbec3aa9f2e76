// image.h — minimal PNG (read: 8/16-bit grey/RGB/palette/alpha, non-interlaced; write: 8-bit
// grey/RGB) and binary PPM (P6) codecs for textures and framebuffer dumps.
//
// The reference uses LodePNG for both directions (src/textures/texture.cpp:58-93 decodes to
// RGB8 with lodepng::decode(..., LCT_RGB); src/fb/framebuffer.cpp:109-143 encodes 8-bit grey /
// RGB).  These are independent implementations of the same file formats (RFC 1950/1951, PNG 1.2).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace qaray_hip {

// Decodes to tightly packed RGB8 (alpha dropped, grey replicated, 16-bit -> high byte).
bool LoadPNG(const char *filename, int &width, int &height, std::vector<unsigned char> &rgb, std::string *err = nullptr);
bool LoadPPM(const char *filename, int &width, int &height, std::vector<unsigned char> &rgb);
// comps: 1 (grey) or 3 (RGB), 8 bits per component.
bool SavePNG(const char *filename, const unsigned char *data, int width, int height, int comps);

// Exposed for tests.
bool Inflate(const unsigned char *src, size_t n, std::vector<unsigned char> &out);
uint32_t Crc32(const unsigned char *p, size_t n, uint32_t crc = 0);

}  // namespace qaray_hip
