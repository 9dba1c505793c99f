// global_faldoi -- drop-in for the reference executable of the same name
// (src/global_faldoi.cpp:1846-2215) with the variational solve running on an MI355X through the
// C ABI of libfaldoi_hip.so.
//
//   global_faldoi ims.txt in_flow.flo out.flo [occl_input.png occl_out.png] [-m method] [-w warps] [-p params.txt]
//                 [-glb_iters n] [-verbose 0|1]
//   additive flags (defaults = reference behaviour):
//                 [-max_inner n] [-tol_stop 0|1] [-chunk k] [-gpu id]
//
// ims.txt: one path per line: I0, I1 [, I-1, I2].  As in the reference, I-1 is only used when
// the file has four lines, otherwise I1 takes its place in the normalisation (:1933-1937).
// Methods: 0 (TV-l2 coupled) and 1 (its weighted id, which the reference's global step maps to the
// same solver, :2132); 2/3 (NLTV-L1, :2145), 4/5 (TV-CSAD, :2152) and 6/7 (NLTV-CSAD, :2137), likewise.
// The occlusion functional (8) is not part of this build and is refused loudly.
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

static int usage(size_t n)
{
    fprintf(stderr, "Without occlusions:\n");
    fprintf(stderr, "Usage: %zu  ims.txt in_flow.flo  out.flo [-m method_val] [-w num_warps] [-p file of parameters] "
                    "[-glb_iters global_iters] [-verbose verbose] \n", n);
    fprintf(stderr, "With occlusions:\n");
    fprintf(stderr, "Usage: %zu  ims.txt in_flow.flo  out.flo occl_input.png occl_out.png [-m method_val] [-w num_warps] "
                    "[-p file of parameters] [-glb_iters global_iters] [-verbose verbose]\n", n);
    fprintf(stderr, "  MI355X build extras: [-max_inner n] [-tol_stop 0|1] [-chunk k] [-gpu id] [-print_params 1]\n");
    return EXIT_FAILURE;
}

int main(int argc, char *argv[])
{
    try {
        std::vector<std::string> args(argv, argv + argc);
        const int nwarps = std::stoi(pick_option(args, "w", std::to_string(FALDOI_DEFAULT_WARPS_GLOBAL)));
        const int method = std::stoi(pick_option(args, "m", "0"));
        const std::string file_params = pick_option(args, "p", "");
        (void)pick_option(args, "glb_iters", "400");     // parsed and ignored for methods 0-7, like the reference (:684)
        const int verbose = std::stoi(pick_option(args, "verbose", "0"));
        const int max_inner = std::stoi(pick_option(args, "max_inner", std::to_string(FALDOI_DEFAULT_MAX_INNER)));
        const int tol_stop = std::stoi(pick_option(args, "tol_stop", "1"));
        const int chunk = std::stoi(pick_option(args, "chunk", "0"));
        const int gpu = std::stoi(pick_option(args, "gpu", "0"));
        const int print_params = std::stoi(pick_option(args, "print_params", "0"));   // additive: show what -p resolved to
        // 3 or 5 positional arguments (:1865-1876).  The occlusion paths of the 5-argument form are only touched by
        // method 8 in the reference (read :1944, written :2170-2180); for methods 0-7 they are accepted and ignored.
        if (args.size() != 6 && args.size() != 4) return usage(args.size());
        const bool tvl1 = method == 0 || method == 1, nltvl1 = method == 2 || method == 3;
        const bool nltvcsad = method == 6 || method == 7, tvcsad = method == 4 || method == 5;
        if (!tvl1 && !nltvcsad && !nltvl1 && !tvcsad) {
            fprintf(stderr, "ERROR: method %d is not available in this build (TV-l2 coupled -m 0/1, NLTV-L1 -m 2/3, "
                            "TV-CSAD -m 4/5, NLTV-CSAD -m 6/7)\n", method);
            return 2;
        }

        std::vector<std::string> files;
        {
            std::ifstream in(args[1]);
            if (!in) { fprintf(stderr, "ERROR: cannot open %s\n", args[1].c_str()); return EXIT_FAILURE; }
            for (std::string line; std::getline(in, line);) files.push_back(line);
        }
        if (files.size() < 2) { fprintf(stderr, "ERROR: %s must list at least two images\n", args[1].c_str()); return EXIT_FAILURE; }

        const Image i0 = read_image_split(files[0]);
        const Image i1 = read_image_split(files[1]);
        const bool have_im1 = files.size() == 4;
        const Image im1 = have_im1 ? read_image_split(files[2]) : Image();
        const Image flow = read_image_split(args[2]);
        if (i0.w != i1.w || i0.h != i1.h || i0.pd != i1.pd ||
            (have_im1 && (im1.w != i0.w || im1.h != i0.h || im1.pd != i0.pd))) {
            fprintf(stderr, "ERROR: input images and flow size mismatch\n");
            return EXIT_FAILURE;
        }
        if (flow.w != i0.w || flow.h != i0.h || flow.pd != 2) {
            fprintf(stderr, "ERROR: input flow field size mismatch\n");
            return EXIT_FAILURE;
        }
        const int pd = i0.pd == 1 ? 1 : 3;               // rgb2gray reads the first three planes (:1820)
        if (i0.pd == 2) { fprintf(stderr, "ERROR: two-channel images are not supported\n"); return EXIT_FAILURE; }

        const EnergyParams ep = read_energy_params(file_params);
        if (print_params) {
            printf("lambda %.9g theta %.9g tau %.9g tol %.9g\n", ep.lambda, ep.theta, ep.tau, ep.tol);
            return EXIT_SUCCESS;
        }
        faldoi_tvl1_params prm;
        faldoi_tvl1_default_params(&prm);
        prm.lambda = ep.lambda; prm.theta = ep.theta; prm.tau = ep.tau; prm.tol = ep.tol;
        prm.warps = nwarps; prm.max_inner = max_inner; prm.tol_stop = tol_stop; prm.chunk = chunk; prm.verbose = verbose;

        if (nltvcsad || nltvl1) {                         // src/global_faldoi.cpp:2042-2047,2137-2151
            if (pd != 3) { fprintf(stderr, "ERROR: the NLTV functionals need colour images (Lab weights)\n"); return EXIT_FAILURE; }
            faldoi_ctx *c = nullptr;
            if (faldoi_ctx_create(gpu, &c)) { fprintf(stderr, "ERROR: no usable GPU: %s\n", faldoi_last_error()); return 3; }
            const size_t n = (size_t)i0.w * i0.h;
            printf("W:%d H:%d Pd:%d\n", i0.w, i0.h, i0.pd);        // :2044
            std::vector<float> lab(3 * n), i0n(n), i1n(n), u(flow.data);
            int rc = faldoi_image_to_lab(i0.data.data(), i0.w, i0.h, lab.data());
            if (!rc) rc = faldoi_preprocess_global(c, i0.data.data(), i1.data.data(), have_im1 ? im1.data.data() : nullptr,
                                                   i0.w, i0.h, pd, i0n.data(), i1n.data());
            faldoi_nltvcsad_params np;                    // lambda/theta/tau are forced per method (:2138-2148)
            if (nltvl1) faldoi_nltvl1_default_params(&np); else faldoi_nltvcsad_default_params(&np);
            np.warps = nwarps; np.max_inner = max_inner; np.verbose = verbose;
            if (!rc) rc = nltvl1 ? faldoi_nltvl1_global(c, i0n.data(), i1n.data(), lab.data(), 3, i0.w, i0.h, &np, u.data(), nullptr)
                                 : faldoi_nltvcsad_global(c, i0n.data(), i1n.data(), lab.data(), 3, i0.w, i0.h, &np, u.data(), nullptr);
            if (rc) fprintf(stderr, "ERROR: %s\n", faldoi_last_error());
            faldoi_ctx_destroy(c);
            if (rc) return 3;
            write_flo(args[3], u.data(), i0.w, i0.h);
            return EXIT_SUCCESS;
        }

        if (tvcsad) {                                      // src/global_faldoi.cpp:2152-2158
            faldoi_ctx *c = nullptr;
            if (faldoi_ctx_create(gpu, &c)) { fprintf(stderr, "ERROR: no usable GPU: %s\n", faldoi_last_error()); return 3; }
            const size_t n = (size_t)i0.w * i0.h;
            std::vector<float> i0n(n), i1n(n), u(flow.data);
            int rc = faldoi_preprocess_global(c, i0.data.data(), i1.data.data(), have_im1 ? im1.data.data() : nullptr, i0.w, i0.h,
                                              pd, i0n.data(), i1n.data());
            faldoi_tvl1_params tp;                           // lambda/theta/tau are forced (:2153-2155); tol comes from the file
            faldoi_tvcsad_default_params(&tp);
            tp.tol = ep.tol; tp.warps = nwarps; tp.max_inner = max_inner; tp.verbose = verbose;
            if (!rc) rc = faldoi_tvcsad_global(c, i0n.data(), i1n.data(), u.data(), nullptr, i0.w, i0.h, &tp, nullptr);
            if (rc) fprintf(stderr, "ERROR: %s\n", faldoi_last_error());
            faldoi_ctx_destroy(c);
            if (rc) return 3;
            write_flo(args[3], u.data(), i0.w, i0.h);
            printf("Exits current level\n");               // :1634
            return EXIT_SUCCESS;
        }

        faldoi_ctx *ctx = nullptr;
        faldoi_tvl1_batch *batch = nullptr;
        auto die = [&](const char *what) {
            fprintf(stderr, "ERROR: %s: %s\n", what, faldoi_last_error());
            if (batch) faldoi_tvl1_batch_destroy(batch);
            if (ctx) faldoi_ctx_destroy(ctx);
            return 3;
        };
        if (faldoi_ctx_create(gpu, &ctx)) return die("no usable GPU");
        if (faldoi_tvl1_batch_create(ctx, 1, i0.w, i0.h, &batch)) return die("allocation");
        if (faldoi_tvl1_batch_upload_images(batch, 0, i0.data.data(), i1.data.data(), have_im1 ? im1.data.data() : nullptr,
                                            pd, flow.data.data(), nullptr))
            return die("upload");
        std::vector<int> iters(nwarps);
        std::vector<float> errs(nwarps);
        faldoi_tvl1_stats st{};
        st.iters_cap = nwarps; st.iters = iters.data(); st.err = errs.data();
        if (faldoi_tvl1_batch_solve(batch, &prm, &st)) return die("solve");
        std::vector<float> u(2 * (size_t)i0.w * i0.h);
        if (faldoi_tvl1_batch_download(batch, 0, u.data(), nullptr)) return die("download");
        if (verbose)
            std::cout << "(tvl2OF) All tasks took " << st.solve_ms * 1e-3 << " (device, " << st.total_iters
                      << " inner iterations)" << std::endl;
        write_flo(args[3], u.data(), i0.w, i0.h);
        faldoi_tvl1_batch_destroy(batch);
        faldoi_ctx_destroy(ctx);
        return EXIT_SUCCESS;
    } catch (const std::exception &e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return EXIT_FAILURE;
    }
}
