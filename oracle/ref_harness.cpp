// ref_harness.cpp -- extern "C" doorways into the reference's OWN compiled objects.
//
// TEST INFRASTRUCTURE ONLY (see faldoi_oracle.h).  This file contains no algorithm: each
// function forwards to the reference implementation, compiled where its sources lie under
// /root/reference/src by oracle/Makefile, so ctypes can reach the C++-mangled symbols.
// It exists to pin oracle/faldoi_oracle.c bit-for-bit against the reference primitives.
// The outputs (oracle/_ref/) are git-ignored and are never shipped with the product.
#include <cstring>
#include "utils.h"                       // reference: src/utils.h
#include "tvl2_model.h"                  // reference: src/tvl2_model.h (guided_tvl2coupled)
#include "tvl2w_model.h"                 // reference: src/tvl2w_model.h (guided_tvl2coupled_w)
extern "C" {
#include "bicubic_interpolation.h"       // reference: src/bicubic_interpolation.h
#include "zoom.h"                        // reference: src/zoom.h (zoom.c is compiled with mask.c, as it includes mask.h)
#include "elap_recsep.h"                 // reference: src/elap_recsep.h
}

extern "C" {

void ref_centered_gradient(const float *in, float *dx, float *dy, int nx, int ny)
{ centered_gradient(in, dx, dy, nx, ny); }                       // src/utils.cpp:367

void ref_forward_gradient(const float *f, float *fx, float *fy, int nx, int ny)
{ forward_gradient(f, fx, fy, nx, ny); }                         // src/utils.cpp:285

void ref_divergence(const float *v1, const float *v2, float *d, int nx, int ny)
{ divergence(v1, v2, d, nx, ny); }                               // src/utils.cpp:239

void ref_gaussian(float *I, int xdim, int ydim, float sigma)
{ gaussian(I, xdim, ydim, sigma); }                              // src/utils.cpp:521

void ref_getminmax(float *mn, float *mx, const float *x, int n)
{ getminmax(mn, mx, x, n); }                                     // src/utils.cpp:679

void ref_image_normalization(const float *I0, const float *I1, float *I0n, float *I1n, int size)
{ image_normalization(I0, I1, I0n, I1n, size); }                 // src/utils.cpp:703

void ref_image_normalization_3(const float *I1, const float *I2, const float *I0,
                               float *I1n, float *I2n, float *I0n, int size)
{ image_normalization_3(I1, I2, I0, I1n, I2n, I0n, size); }      // src/utils.cpp:743

float ref_bicubic_at(const float *in, float uu, float vv, int nx, int ny, int border_out)
{ return bicubic_interpolation_at(in, uu, vv, nx, ny, border_out != 0); }   // src/bicubic_interpolation.c:138

void ref_bicubic_warp(const float *in, const float *u, const float *v, float *out,
                      int nx, int ny, int border_out)
{ bicubic_interpolation_warp(in, u, v, out, nx, ny, border_out != 0); }     // src/bicubic_interpolation.c:245

void ref_zoom_size(int nx, int ny, int *nxx, int *nyy, float factor)
{ zoom_size(nx, ny, nxx, nyy, factor); }                          // src/zoom.c:19

void ref_zoom_out(const float *I, float *Iout, int nx, int ny, float factor)
{ zoom_out(I, Iout, nx, ny, factor); }                            // src/zoom.c:38

void ref_zoom_in(const float *I, float *Iout, int nx, int ny, int nxx, int nyy)
{ zoom_in(I, Iout, nx, ny, nxx, nyy); }                           // src/zoom.c:82

void ref_elap_recursive_separable(float *out, float *in, int w, int h, int pd, float timestep, int niter, int scale)
{ elap_recursive_separable(out, in, w, h, pd, timestep, niter, scale); }     // src/elap_recsep.c:225

// ---- the local TV-L1 patch operator: the reference's own guided_tvl2coupled (src/tvl2_model.cpp:249)
// on the reference's own scratch structure, allocated by its initialize_stuff_tvl2coupled (:16) and
// zero-filled here so that "what an earlier call left behind" is well defined.
struct RefLocal {
    SpecificOFStuff stuff;
    OpticalFlowData ofd;
    int w, h;
};

void *ref_local_tvl1_create(int w, int h)
{
    RefLocal *r = new RefLocal();
    r->w = w; r->h = h;
    memset(&r->ofd, 0, sizeof(OpticalFlowData));
    initialize_stuff_tvl2coupled(&r->stuff, &r->ofd, w, h);
    Tvl2CoupledOFStuff &t = r->stuff.tvl2;
    float *planes[] = {t.xi11, t.xi12, t.xi21, t.xi22, t.u1x, t.u1y, t.u2x, t.u2y, t.v1, t.v2, t.rho_c, t.grad,
                       t.u1_, t.u2_, t.u1Aux, t.u2Aux, t.I1x, t.I1y, t.I1w, t.I1wx, t.I1wy, t.div_xi1, t.div_xi2, t.u_N};
    for (float *p : planes) memset(p, 0, sizeof(float) * (size_t)w * h);
    return r;
}

void ref_local_tvl1_destroy(void *h)
{
    RefLocal *r = (RefLocal *)h;
    free_stuff_tvl2coupled(&r->stuff);
    delete r;
}

// One of_estimation call for the TV-l2 coupled functional with of_estimation's hard-wired parameters
// (src/energy_model.cpp:704-709,796-800).  I1x/I1y are copied into the structure as prepare_stuff does.
float ref_local_tvl1_estimate(void *h, const float *I0, const float *I1, const float *I1x, const float *I1y,
                              float *u1, float *u2, int ii, int ij, int ei, int ej, int max_iter_patch)
{
    RefLocal *r = (RefLocal *)h;
    const size_t n = (size_t)r->w * r->h;
    memcpy(r->stuff.tvl2.I1x, I1x, sizeof(float) * n);
    memcpy(r->stuff.tvl2.I1y, I1y, sizeof(float) * n);
    r->ofd.u1 = u1;
    r->ofd.u2 = u2;
    r->ofd.params.max_iter_patch = max_iter_patch;
    float ener = 0.f;
    guided_tvl2coupled(I0, I1, &r->ofd, &r->stuff.tvl2, &ener, ii, ij, ei, ej, 40, 0.3, 0.125, 0.01, 1, false,
                       r->w, r->h);
    return ener;
}

const float *ref_local_tvl1_div(void *h, int which)
{
    RefLocal *r = (RefLocal *)h;
    return which ? r->stuff.tvl2.div_xi2 : r->stuff.tvl2.div_xi1;
}

// ---- the weighted functional: the reference's guided_tvl2coupled_w (src/tvl2w_model.cpp:254) the way
// of_estimation calls it (src/energy_model.cpp:742-749), weights from its gaussian1Dweight (src/utils.cpp:633)
struct RefLocalW {
    SpecificOFStuff stuff;
    OpticalFlowData ofd;
    int w, h, wr;
};

void ref_gaussian1d_weight(float *I, int r) { gaussian1Dweight(I, r); }     // the C++ copy, src/utils.cpp:633 (NOT what prepare_stuff binds)

void *ref_local_tvl1w_create(int w, int h, int wr)
{
    RefLocalW *r = new RefLocalW();
    r->w = w; r->h = h; r->wr = wr;
    memset(&r->ofd, 0, sizeof(OpticalFlowData));
    r->ofd.params.w_radio = wr;
    initialize_stuff_tvl2coupled_w(&r->stuff, &r->ofd, w, h);
    Tvl2CoupledOFStuff_W &t = r->stuff.tvl2w;
    float *planes[] = {t.xi11, t.xi12, t.xi21, t.xi22, t.u1x, t.u1y, t.u2x, t.u2y, t.v1, t.v2, t.rho_c, t.grad,
                       t.u1_, t.u2_, t.u1Aux, t.u2Aux, t.I1x, t.I1y, t.I1w, t.I1wx, t.I1wy, t.div_xi1, t.div_xi2, t.u_N};
    for (float *p : planes) memset(p, 0, sizeof(float) * (size_t)w * h);
    // prepare_stuff (src/energy_model.cpp:432-435) fills t.weight with gaussian1Dweight -- the C-linkage copy of src/mask.c,
    // which this translation unit cannot name next to utils.h's C++ declaration: the caller passes those weights in
    // (ref_gaussian1d_weight_c of ref_harness_models.cpp) through ref_local_tvl1w_set_weight.
    memset(t.weight, 0, sizeof(float) * (2 * wr + 1));
    return r;
}

void ref_local_tvl1w_set_weight(void *h, const float *weight)
{
    RefLocalW *r = (RefLocalW *)h;
    memcpy(r->stuff.tvl2w.weight, weight, sizeof(float) * (2 * r->wr + 1));
}

void ref_local_tvl1w_destroy(void *h)
{
    RefLocalW *r = (RefLocalW *)h;
    free_stuff_tvl2coupled_w(&r->stuff);
    delete r;
}

// wr_call: the radius in force during the call (1 for the seeds); iiw/ijw as get_index_weight sets them
float ref_local_tvl1w_estimate(void *h, const float *I0, const float *I1, const float *I1x, const float *I1y,
                               float *u1, float *u2, int ii, int ij, int ei, int ej, int max_iter_patch, int iiw, int ijw,
                               int wr_call)
{
    RefLocalW *r = (RefLocalW *)h;
    const size_t n = (size_t)r->w * r->h;
    Tvl2CoupledOFStuff_W &t = r->stuff.tvl2w;
    memcpy(t.I1x, I1x, sizeof(float) * n);
    memcpy(t.I1y, I1y, sizeof(float) * n);
    t.iiw = iiw;
    t.ijw = ijw;
    r->ofd.u1 = u1;
    r->ofd.u2 = u2;
    r->ofd.params.max_iter_patch = max_iter_patch;
    float lambda = 40;                                               // src/energy_model.cpp:704, :744-745
    const float central = t.weight[wr_call + 1];
    lambda = lambda / (central * central);
    float ener = 0.f;
    guided_tvl2coupled_w(I0, I1, &r->ofd, &t, &ener, ii, ij, ei, ej, lambda, 0.3, 0.125, 0.01, 1, false, r->w, r->h);
    return ener;
}

}  // extern "C"
