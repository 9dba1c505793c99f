// faldoi_imginfo -- prints "w h pd fnv1a64" of an image as the host readers decode it (planar float
// samples); used by the tests to pin the PNG/PNM/.flo decoders against an independent decoder.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "image_io.hpp"

int main(int argc, char *argv[])
{
    if (argc != 2) { fprintf(stderr, "usage: %s image\n", argv[0]); return 1; }
    try {
        const faldoi_host::Image im = faldoi_host::read_image_split(argv[1]);
        uint64_t h = 1469598103934665603ull;
        const unsigned char *p = (const unsigned char *)im.data.data();
        for (size_t i = 0; i < im.data.size() * sizeof(float); ++i) { h ^= p[i]; h *= 1099511628211ull; }
        printf("%d %d %d %016llx\n", im.w, im.h, im.pd, (unsigned long long)h);
    } catch (const std::exception &e) { fprintf(stderr, "ERROR: %s\n", e.what()); return 1; }
    return 0;
}
