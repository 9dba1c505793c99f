// faldoi_flow -- the stage order of the reference's driver scripts without Python and without intermediate
// files (scripts_python/faldoi_sift.py:265-318, faldoi_deep.py:262-330): match lists -> sparse flows
// (src/sparse_flow.cpp) -> local step (local_faldoi, -m 0) -> global step (global_faldoi, -m 0), all through
// the C ABI of libfaldoi_hip.so.  The matchers themselves (sift_cli / match_cli / deepmatching) are external
// programs and stay outside: this program starts from their output.
//
//   faldoi_flow ims.txt matches_fwd.txt matches_bwd.txt out.flo
//               [-matcher sift|deep|cut]   column layout of the match lists:
//                    sift: a match_cli line; x0 y0 x1 y1 = columns 2,1,6,5 (cut_matching_list,
//                          scripts_python/auxiliar_faldoi_functions.py:19-30)
//                    deep: columns 1-4 (cut_deep_list :32-45);  cut: already "x0 y0 x1 y1"
//               [-local_out local.flo] [-sim_out sim_map]      keep the intermediate results
//               [-m 0|1] [-wr 5] [-loc_it 3] [-max_pch_it 4] [-fb_thresh 2] [-split_img 0|1|2] [-h_parts 3] [-v_parts 2]
//               [-w 5] [-verbose 0] [-gpu 0]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "faldoi_hip.h"
#include "image_io.hpp"
#include "options.hpp"

using namespace faldoi_host;

// src/sparse_flow.cpp:20-45 on an in-memory match list (out-of-image matches are skipped, as in our sparse_flow)
static std::vector<float> sparse_flow_from_matches(const std::string &path, const std::string &matcher, int w, int h)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    const size_t n = (size_t)w * h;
    std::vector<float> out(2 * n, std::numeric_limits<float>::quiet_NaN());
    for (std::string line; std::getline(in, line);) {
        std::istringstream ss(line);
        std::vector<std::string> tok;
        for (std::string t; ss >> t;) tok.push_back(t);
        int c[4] = { 0, 1, 2, 3 };
        if (matcher == "sift") { c[0] = 1; c[1] = 0; c[2] = 5; c[3] = 4; }
        if (tok.size() <= (size_t)std::max(std::max(c[0], c[1]), std::max(c[2], c[3]))) continue;
        float v[4];
        bool ok = true;
        for (int k = 0; k < 4 && ok; ++k) ok = sscanf(tok[c[k]].c_str(), "%f", &v[k]) == 1;
        if (!ok) continue;
        const int i = (int)std::floor(v[0]), j = (int)std::floor(v[1]);
        if (i < 0 || j < 0 || i >= w || j >= h) continue;
        out[(size_t)j * w + i] = v[2] - v[0];
        out[n + (size_t)j * w + i] = v[3] - v[1];
    }
    return out;
}

int main(int argc, char *argv[])
{
    try {
        std::vector<std::string> args(argv, argv + argc);
        const std::string matcher = pick_option(args, "matcher", "sift");
        const std::string local_out = pick_option(args, "local_out", "");
        const std::string sim_out = pick_option(args, "sim_out", "");
        const int w_radio = std::stoi(pick_option(args, "wr", "5"));
        const int loc_it = std::stoi(pick_option(args, "loc_it", "3"));
        const int max_pch_it = std::stoi(pick_option(args, "max_pch_it", "4"));
        const float fb_thresh = std::stof(pick_option(args, "fb_thresh", "2"));
        const int split_img = std::stoi(pick_option(args, "split_img", "0"));
        const int h_parts = std::stoi(pick_option(args, "h_parts", "3"));
        const int v_parts = std::stoi(pick_option(args, "v_parts", "2"));
        const int first_h_parts = std::stoi(pick_option(args, "first_h_parts", "0"));     // additive, with -split_img 2
        const int first_v_parts = std::stoi(pick_option(args, "first_v_parts", "0"));
        const int local_method = std::stoi(pick_option(args, "m", "0"));       // 0 / 1: local functional; the global step uses TV-l2 either way (:2132)
        const int warps = std::stoi(pick_option(args, "w", std::to_string(FALDOI_DEFAULT_WARPS_GLOBAL)));
        const int verbose = std::stoi(pick_option(args, "verbose", "0"));
        const int gpu = std::stoi(pick_option(args, "gpu", "0"));
        if (args.size() != 5 || (matcher != "sift" && matcher != "deep" && matcher != "cut")) {
            fprintf(stderr, "usage: %s ims.txt matches_fwd.txt matches_bwd.txt out.flo [-matcher sift|deep|cut] [-local_out f.flo] "
                            "[-sim_out f] [-m 0|1] [-wr 5] [-loc_it 3] [-max_pch_it 4] [-fb_thresh 2] [-split_img 0|1|2] [-h_parts 3] [-v_parts 2] "
                            "[-w 5] [-verbose 0] [-gpu 0]\n", argv[0]);
            return 1;
        }
        std::vector<std::string> files;
        {
            std::ifstream in(args[1]);
            if (!in) { fprintf(stderr, "ERROR: cannot open %s\n", args[1].c_str()); return 1; }
            for (std::string line; std::getline(in, line);) files.push_back(line);
        }
        if (files.size() != 2 && files.size() != 4) { fprintf(stderr, "ERROR: %s must list 2 or 4 images\n", args[1].c_str()); return 1; }
        const Image i0 = read_image_split(files[0]);
        const Image i1 = read_image_split(files[1]);
        const bool have_im1 = files.size() == 4;
        const Image im1 = have_im1 ? read_image_split(files[2]) : Image();
        if (i0.w != i1.w || i0.h != i1.h || i0.pd != i1.pd || i0.pd == 2 ||
            (have_im1 && (im1.w != i0.w || im1.h != i0.h || im1.pd != i0.pd))) {
            fprintf(stderr, "ERROR: input images size mismatch\n");
            return 1;
        }
        const int w = i0.w, h = i0.h, pd = i0.pd == 1 ? 1 : 3;
        const size_t n = (size_t)w * h;
        const std::vector<float> go = sparse_flow_from_matches(args[2], matcher, w, h);
        const std::vector<float> ba = sparse_flow_from_matches(args[3], matcher, w, h);

        faldoi_ctx *ctx = nullptr;
        if (faldoi_ctx_create(gpu, &ctx)) { fprintf(stderr, "ERROR: no usable GPU: %s\n", faldoi_last_error()); return 3; }
        auto die = [&](const char *what) {
            fprintf(stderr, "ERROR: %s: %s\n", what, faldoi_last_error());
            faldoi_ctx_destroy(ctx);
            return 3;
        };
        // local step
        faldoi_local_params lp;
        faldoi_local_default_params(&lp);
        lp.w_radio = w_radio; lp.iterations_of = loc_it; lp.max_iter_patch = max_pch_it; lp.epsilon = fb_thresh; lp.verbose = verbose;
        lp.split_img = split_img; lp.h_parts = h_parts; lp.v_parts = v_parts; lp.first_h_parts = first_h_parts; lp.first_v_parts = first_v_parts; lp.method = local_method;
        std::vector<float> u(2 * n), ene(n);
        faldoi_local_stats ls;
        if (faldoi_match_growing_variational(ctx, go.data(), ba.data(), i0.data.data(), i1.data.data(), nullptr, nullptr, w, h, pd,
                                             &lp, ene.data(), u.data(), &ls))
            return die("local step");
        if (!local_out.empty()) write_flo(local_out, u.data(), w, h);
        if (!sim_out.empty()) {
            FILE *f = fopen(sim_out.c_str(), "w");
            if (!f) throw std::runtime_error("cannot write " + sim_out);
            fprintf(f, "P2\n%d %d\n65535\n", w, h);
            for (size_t q = 0; q < n; ++q) fprintf(f, "%a\n", ene[q]);
            fclose(f);
        }
        // global step from the local flow (src/global_faldoi.cpp:2049-2136)
        std::vector<float> i0n(n), i1n(n);
        if (faldoi_preprocess_global(ctx, i0.data.data(), i1.data.data(), have_im1 ? im1.data.data() : nullptr, w, h, pd, i0n.data(),
                                     i1n.data()))
            return die("preprocessing");
        faldoi_tvl1_params gp;
        faldoi_tvl1_default_params(&gp);
        gp.warps = warps; gp.verbose = verbose;
        faldoi_tvl1_stats gs{};
        if (faldoi_tvl1_global(ctx, i0n.data(), i1n.data(), u.data(), nullptr, w, h, &gp, &gs)) return die("global step");
        write_flo(args[4], u.data(), w, h);
        if (verbose)
            std::cout << "faldoi_flow: local step " << ls.growing_ms * 1e-3 << " s (" << ls.solves[0] + ls.solves[1]
                      << " windows), global step " << gs.solve_ms * 1e-3 << " s (" << gs.total_iters << " iterations)" << std::endl;
        faldoi_ctx_destroy(ctx);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
}
