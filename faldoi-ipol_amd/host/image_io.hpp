// image_io.hpp -- the file formats on either side of the hot path.
//
// The reference reads and writes everything through the vendored iio library (src/iio.c); only
// two behaviours of it matter to the three executables and are reproduced here:
//   * images arrive as PLANAR float channels with the file's sample values (0..255 for 8-bit
//     PNG), iio_read_image_float_split, src/iio.c:2982;
//   * flows are Middlebury .flo: "PIEH", int32 w, int32 h, then h*w interleaved (u,v) float32
//     (writer src/iio.c:2539-2553, reader src/iio.c:1807), held in memory as two planes.
// Supported inputs: PNG (non-interlaced, 8/16-bit gray, gray+alpha, RGB, RGBA, palette), binary
// PGM/PPM (P5/P6), and .flo.  No libpng dependency: PNG is decoded with zlib's inflate.
#pragma once
#include <string>
#include <vector>

namespace faldoi_host {

struct Image {
    int w = 0, h = 0, pd = 0;          // pd planes of w*h floats, plane-major
    std::vector<float> data;
};

// Throws std::runtime_error with a message naming the file on any failure.
Image read_image_split(const std::string &path);
Image read_flo(const std::string &path);
void write_flo(const std::string &path, const float *u_planes /* 2*w*h */, int w, int h);

}  // namespace faldoi_host
