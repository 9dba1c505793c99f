// options.hpp -- the reference's flag convention (src/utils_preprocess.cpp:10-35): options are
// "-name value" pairs anywhere on the command line; pick_option removes the pair it consumed, and
// what is left are the positional arguments (argv[0] included), whose COUNT selects the usage.
#pragma once
#include <fstream>
#include <string>
#include <vector>

namespace faldoi_host {

inline std::string pick_option(std::vector<std::string> &args, const std::string &option, const std::string &def)
{
    const std::string flag = "-" + option;
    for (size_t i = 0; i < args.size(); ++i) {
        if (args[i] == flag) {
            if (i + 1 >= args.size()) continue;          // a trailing flag without a value is left in place
            const std::string value = args[i + 1];
            args.erase(args.begin() + (long)i, args.begin() + (long)i + 2);
            return value;
        }
    }
    return def;
}

// Energy parameter file (src/utils_preprocess.cpp:65-155): nine lines lambda, theta, tau, beta,
// alpha, tau_u, tau_eta, tau_chi, mu; a value <= 0 (or tau > 0.25) falls back to the default.
struct EnergyParams {
    float lambda = 40.f, theta = 0.3f, tau = 0.125f;     // src/parameters.h:20-22
    float tol = 0.01f;                                   // PAR_DEFAULT_TOL_D, :31
};

inline EnergyParams read_energy_params(const std::string &file)
{
    EnergyParams p;
    if (file.empty()) return p;
    std::ifstream in(file);
    std::string line;
    // all nine lines go through stof, which throws on a missing or malformed one exactly like the reference's
    // init_params (an unreadable file gives an empty first line, hence std::invalid_argument there too)
    float v[9];
    for (float &x : v) { std::getline(in, line); x = std::stof(line); }
    if (v[0] > 0) p.lambda = v[0];                       // :74-78
    if (v[1] > 0) p.theta = v[1];                        // :83-87
    if (v[2] > 0 && v[2] <= 0.25f) p.tau = v[2];         // :92-96 (0.25 is exact in both precisions)
    return p;                                            // beta, alpha, tau_u, tau_eta, tau_chi, mu: occlusion model only
}

}  // namespace faldoi_host
