// local_faldoi -- drop-in for the reference executable of the same name
// (src/local_faldoi.cpp:1756-2110) with the match growing running on an MI355X through the C ABI of
// libfaldoi_hip.so (faldoi_match_growing_variational).
//
//   local_faldoi ims.txt in0.flo in1.flo out.flo sim_map [sal0 sal1]
//                [-m method] [-wr radius] [-p params.txt] [-loc_it n] [-max_pch_it n] [-split_img 0]
//                [-h_parts n] [-v_parts n] [-fb_thresh t] [-partial_res 0|1] [-verbose 0|1]
//   additive flags: [-gpu id] [-budget windows_per_launch]
//
// ims.txt: I0, I1 [, I-1, I2]; three lines are an error (:1852-1860); the extra frames only matter to
// the occlusion functional, which is not part of this build.  in0 / in1: forward / backward seed flows
// (NaN = no seed).  sim_map receives the energy map the way the reference's iio build writes a float
// image under a name it has no encoder for: ASCII "P2" with one C99 hex-float per line
// (src/iio.c:3505-3511).  Like the reference, the local step ignores the values of
// the -p file (src/energy_model.cpp:704-709).
// Built: -m 0..7 (TV-l2 coupled, NLTV-L1, TV-CSAD, NLTV-CSAD, each plain / weighted: of_estimation's switch,
// src/energy_model.cpp:691-802); image partitions with -m 0 / 1
// (-split_img 1 -h_parts H -v_parts V; not together with saliency maps; -split_img 2, additive: partitions from
// local iteration 0 on -- faster for one pair, deterministic, but not the reference's flow).  Other functionals are refused loudly.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "faldoi_hip.h"
#include "image_io.hpp"
#include "options.hpp"

using namespace faldoi_host;

static int usage(size_t n, const char *argv0)
{
    fprintf(stderr, "Without occlusions (n of params: 5 or 7 + 1 (own function name)):\n");
    fprintf(stderr, "usage %zu :\n\t%s ims.txt in0.flo in1.flo out.flo sim_map.tiff [sal0.tiff sal1.tiff]"
                    " [-m method_id] [-wr windows_radio] [-p file of parameters]"
                    " [-loc_it local_iters] [-max_pch_it max_iters_patch]"
                    " [-split_img split_image] [-h_parts horiz_parts]"
                    " [-v_parts vert_parts] [-fb_thresh thresh] [-partial_res val] [-verbose verbose]\n", n, argv0);
    fprintf(stderr, "  MI355X build extras: [-gpu id] [-budget windows_per_launch]\n");
    return 1;
}

static void write_sim_map(const std::string &path, const float *e, int w, int h)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) throw std::runtime_error("cannot write " + path);
    bool bytes = true;                                            // these_floats_are_actually_bytes, src/iio.c:3314-3331
    for (size_t i = 0; i < (size_t)w * h && bytes; ++i) bytes = (e[i] == std::floor(e[i])) && e[i] >= 0 && e[i] < 256;
    fprintf(f, "P2\n%d %d\n%d\n", w, h, bytes ? 255 : 65535);
    for (size_t i = 0; i < (size_t)w * h; ++i) fprintf(f, "%a\n", e[i]);
    fclose(f);
}

namespace {
struct PartialSink {
    std::string dir;
    bool split;
    static void write(void *user, int /*pair*/, int percent, int iteration, int part_idx, int w, int h, const float *flow)
    {
        const PartialSink &me = *static_cast<const PartialSink *>(user);
        std::string name = me.dir;
        if (!name.empty() && name.back() != '/') name += '/';
        name += "partial_results_fwd_" + std::to_string(percent) + "_iter_" + std::to_string(iteration);
        if (me.split) name += "_part_idx" + std::to_string(part_idx);
        name += ".flo";
        try {
            write_flo(name, flow, w, h);
        } catch (const std::exception &e) {                   // (the reference's writer aborts the program here; a partial result is
            fprintf(stderr, "warning: %s\n", e.what());       // a by-product: say so and go on)
        }
    }
};
}  // namespace

int main(int argc, char *argv[])
{
    try {
        std::vector<std::string> args(argv, argv + argc);
        const int w_radio = std::stoi(pick_option(args, "wr", "5"));
        const int method = std::stoi(pick_option(args, "m", "0"));
        (void)pick_option(args, "p", "");
        const int loc_it = std::stoi(pick_option(args, "loc_it", "3"));
        const int max_pch_it = std::stoi(pick_option(args, "max_pch_it", "4"));
        const int split_img = std::stoi(pick_option(args, "split_img", "0"));
        const int h_parts = std::stoi(pick_option(args, "h_parts", "3"));
        const int v_parts = std::stoi(pick_option(args, "v_parts", "2"));
        const int first_h_parts = std::stoi(pick_option(args, "first_h_parts", "0"));     // additive, with -split_img 2
        const int first_v_parts = std::stoi(pick_option(args, "first_v_parts", "0"));
        const float fb_thresh = std::stof(pick_option(args, "fb_thresh", "2"));
        const int partial_res = std::stoi(pick_option(args, "partial_res", "0"));
        // additive: where -partial_res 1 writes (the reference's fixed place, relative to the working directory, :950)
        const std::string partial_dir = pick_option(args, "partial_dir", "../Results/Partial_results/");
        const int verbose = std::stoi(pick_option(args, "verbose", "0"));
        const int gpu = std::stoi(pick_option(args, "gpu", "0"));
        const int budget = std::stoi(pick_option(args, "budget", "0"));
        if (args.size() < 6 || args.size() > 9) return usage(args.size(), args[0].c_str());
        if (args.size() == 7 || args.size() == 9) {
            fprintf(stderr, "ERROR: the occlusion outputs (TV-l1 with occlusions, -m 8) are not part of this build\n");
            return 2;
        }
        if (method < 0 || method > 7) {                       // of_estimation's switch, src/energy_model.cpp:711-800
            fprintf(stderr, "ERROR: method %d is not available in the local step of this build (functionals 0..7; the occlusion "
                            "model -m 8 is not built)\n", method);
            return 2;
        }

        std::vector<std::string> files;
        {
            std::ifstream in(args[1]);
            if (!in) { fprintf(stderr, "ERROR: cannot open %s\n", args[1].c_str()); return 1; }
            for (std::string line; std::getline(in, line);) files.push_back(line);
        }
        if (files.size() == 3) {                                  // :1852-1860
            fprintf(stderr, "ERROR: 3 images given as input\nWithout occlusions:\nUsage: 2 images in the following order: I0, I1\n"
                            "With occlusions:\nUsage: 4 images in the following order: I0, I1, I-1, I2\n");
            return 1;
        }
        if (files.size() < 2) { fprintf(stderr, "ERROR: %s must list at least two images\n", args[1].c_str()); return 1; }
        if (files.size() == 4) fprintf(stderr, "Only two of the four images given will be used, according to method selected\nMethod: TV-l2 coupled\n");

        const Image i0 = read_image_split(files[0]);
        const Image i1 = read_image_split(files[1]);
        const Image go = read_image_split(args[2]);
        const Image ba = read_image_split(args[3]);
        if (i0.w != i1.w || i0.h != i1.h || i0.pd != i1.pd) { fprintf(stderr, "ERROR: input images size mismatch\n"); return 1; }
        if (go.w != ba.w || go.h != ba.h || go.pd != 2 || ba.pd != 2 || go.w != i0.w || go.h != i0.h) {
            fprintf(stderr, "ERROR: input flow field size mismatch (global\n");
            return 1;
        }
        if (i0.pd == 2) { fprintf(stderr, "ERROR: two-channel images are not supported\n"); return 1; }
        const int pd = i0.pd == 1 ? 1 : 3;
        const size_t n = (size_t)i0.w * i0.h;
        Image sal0, sal1;
        if (args.size() == 8) {
            sal0 = read_image_split(args[6]);
            sal1 = read_image_split(args[7]);
            if (sal0.w != i0.w || sal0.h != i0.h || sal1.w != i0.w || sal1.h != i0.h || sal0.pd != 1 || sal1.pd != 1) {
                fprintf(stderr, "ERROR: saliency maps must be single-channel images of the frame size\n");
                return 1;
            }
            fprintf(stderr, "Reading given saliency values\n");
            for (size_t i = 0; i < n; ++i) {
                if (sal0.data[i] < 0.0) fprintf(stderr, "sal0 index: %zu\n", i);
                if (sal1.data[i] < 0.0) fprintf(stderr, "sal1 index: %zu\n", i);
            }
        } else {
            fprintf(stderr, "Saliency values not given\n");
        }

        faldoi_local_params prm;
        faldoi_local_default_params(&prm);
        prm.w_radio = w_radio; prm.iterations_of = loc_it; prm.max_iter_patch = max_pch_it; prm.epsilon = fb_thresh;
        prm.verbose = verbose; prm.method = method; prm.split_img = split_img; prm.h_parts = h_parts; prm.v_parts = v_parts;
        prm.first_h_parts = first_h_parts; prm.first_v_parts = first_v_parts;
        if (budget > 0) prm.budget = budget;
        // -partial_res 1 (src/local_faldoi.cpp:942-1036): the forward growings' flow at 30 / 70 / 80 / 95 / 100 % of their pixels,
        // under the reference's names -- "_part_idx<k>" is appended whenever -split_img is on, whole-image growings included
        // (k = -1 inside the iterations, the iteration number for the last growing, as the reference passes it)
        PartialSink sink{ partial_dir, split_img != 0 };
        if (partial_res == 1) { prm.partial_cb = &PartialSink::write; prm.partial_user = &sink; }
        faldoi_ctx *ctx = nullptr;
        if (faldoi_ctx_create(gpu, &ctx)) { fprintf(stderr, "ERROR: no usable GPU: %s\n", faldoi_last_error()); return 3; }
        std::vector<float> out(2 * n), ene(n);
        faldoi_local_stats st;
        const int rc = faldoi_match_growing_variational(ctx, go.data.data(), ba.data.data(), i0.data.data(), i1.data.data(),
                                                        sal0.data.empty() ? nullptr : sal0.data.data(),
                                                        sal1.data.empty() ? nullptr : sal1.data.data(), i0.w, i0.h, pd, &prm,
                                                        ene.data(), out.data(), &st);
        if (rc) fprintf(stderr, "ERROR: %s\n", faldoi_last_error());
        faldoi_ctx_destroy(ctx);
        if (rc) return rc == FALDOI_ERR_INVALID ? 1 : 3;
        for (int it = 0; it < loc_it && it < 8; ++it) {           // the reference prints these fractions (:187)
            printf("FB-Chosen: %f\n", (double)st.fb_kept[2 * it]);
            printf("FB-Chosen: %f\n", (double)st.fb_kept[2 * it + 1]);
        }
        std::cout << "(local_faldoi) Match growing variational took " << st.growing_ms * 1e-3 << " (device, "
                  << st.solves[0] + st.solves[1] << " windows)" << std::endl;
        write_flo(args[4], out.data(), i0.w, i0.h);
        write_sim_map(args[5], ene.data(), i0.w, i0.h);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
}
