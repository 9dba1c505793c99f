// ref_harness.cpp -- extern "C" doorways into the reference's OWN compiled objects.
//
// TEST INFRASTRUCTURE ONLY (see faldoi_oracle.h).  This file contains no algorithm: each
// function forwards to the reference implementation, compiled where its sources lie under
// /root/reference/src by oracle/Makefile, so ctypes can reach the C++-mangled symbols.
// It exists to pin oracle/faldoi_oracle.c bit-for-bit against the reference primitives.
// The outputs (oracle/_ref/) are git-ignored and are never shipped with the product.
#include "utils.h"                       // reference: src/utils.h
extern "C" {
#include "bicubic_interpolation.h"       // reference: src/bicubic_interpolation.h
#include "zoom.h"                        // reference: src/zoom.h (zoom.c is compiled with mask.c, as it includes mask.h)
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

}  // extern "C"
