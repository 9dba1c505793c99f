// sparse_flow -- drop-in for the reference executable (src/sparse_flow.cpp:1-73): turns a list of
// matches "x0 y0 x1 y1" (one per line) into a NaN-initialised two-plane flow with
// out[floor(y0)*w + floor(x0)] = (x1 - x0, y1 - y0).  CPU only, as in the reference.  The one
// deliberate difference: matches that fall outside the image are skipped (the reference writes
// out of bounds there, :35-39).
//   sparse_flow matches.txt W H out.flo
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "image_io.hpp"

int main(int argc, char *argv[])
{
    if (argc != 5) { fprintf(stderr, "usage %d :\n\t%s sift_matches.txt colums rows out.flo\n", argc, argv[0]); return 1; }
    try {
        const int w = atoi(argv[2]), h = atoi(argv[3]);
        if (w <= 0 || h <= 0) { fprintf(stderr, "ERROR: bad size\n"); return 1; }
        const size_t n = (size_t)w * h;
        std::vector<float> out(2 * n, std::numeric_limits<float>::quiet_NaN());
        std::ifstream in(argv[1]);
        if (!in) { fprintf(stderr, "ERROR: cannot open %s\n", argv[1]); return 1; }
        float x0, y0, x1, y1;
        for (std::string line; std::getline(in, line);) {        // one match per line, first four columns
            if (sscanf(line.c_str(), "%f %f %f %f", &x0, &y0, &x1, &y1) != 4) continue;
            const int i = (int)std::floor(x0), j = (int)std::floor(y0);
            if (i < 0 || j < 0 || i >= w || j >= h) continue;
            out[(size_t)j * w + i] = x1 - x0;
            out[n + (size_t)j * w + i] = y1 - y0;
        }
        faldoi_host::write_flo(argv[4], out.data(), w, h);
    } catch (const std::exception &e) { fprintf(stderr, "ERROR: %s\n", e.what()); return 1; }
    return 0;
}
