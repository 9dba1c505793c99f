/*
 * faldoi_hip.h -- C ABI of libfaldoi_hip.so: the MI355X (gfx950) implementation of FALDOI's
 * dense variational minimisation hot path.
 *
 * The reference (fperezgamonal/faldoi-ipol) has no FFI/plugin ABI; its de-facto operator
 * surface is a set of C++ functions called in-process.  Each entry point
 * below names the reference function it replaces (file:line in the upstream repo) and keeps
 * its argument meaning: planar row-major fp32 images, caller-owned host buffers, flow stored
 * as two contiguous planes (u1 then u2), dual variables xi caller-zeroed and updated in place.
 *
 * Conventions
 *   - plain C types only; every function returns an int status (FALDOI_OK == 0) and never
 *     throws or aborts (the reference's void functions abort()/assert on failure);
 *   - a faldoi_ctx owns one device + one HIP stream + workspaces; distinct contexts may be used
 *     from distinct host threads at the same time, also on one device (tested: tests/test_gpu_round3.py,
 *     "threads"); one context is not thread-safe (mirrors the reference's per-thread
 *     SpecificOFStuff/OpticalFlowData ownership, src/local_faldoi.cpp:1192-1219,1305-1347);
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point returns
 *     FALDOI_ERR_NO_DEVICE.
 */
#ifndef FALDOI_HIP_H
#define FALDOI_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FALDOI_ABI_VERSION 1

enum {
    FALDOI_OK = 0,
    FALDOI_ERR_INVALID = 1,      /* bad argument (null pointer, size, parameter range) */
    FALDOI_ERR_NO_DEVICE = 2,    /* no HIP device / device index out of range */
    FALDOI_ERR_HIP = 3,          /* a HIP runtime call failed; see faldoi_last_error() */
    FALDOI_ERR_OOM = 4,          /* device or host allocation failed */
    FALDOI_ERR_STATE = 5         /* call sequence error (e.g. solve before upload) */
};

/* Reference constants, src/parameters.h:20-22,31,43,51 */
#define FALDOI_DEFAULT_LAMBDA 40.0f
#define FALDOI_DEFAULT_THETA 0.3f
#define FALDOI_DEFAULT_TAU 0.125f
#define FALDOI_DEFAULT_TOL 0.01f
#define FALDOI_DEFAULT_MAX_INNER 400
#define FALDOI_DEFAULT_WARPS_GLOBAL 5
#define FALDOI_MAX_CHUNK 16        /* inner iterations fused into one launch (halo width) */

typedef struct faldoi_ctx faldoi_ctx;
typedef struct faldoi_tvl1_batch faldoi_tvl1_batch;
typedef struct faldoi_local_tvl1 faldoi_local_tvl1;

/* Solver parameters: the scalar arguments of tvl2OF (src/global_faldoi.cpp:565-572) plus the
 * two knobs the reference hard-codes (loop bound MAX_ITERATIONS_GLOBAL and the tolerance stop,
 * src/global_faldoi.cpp:684) and one scheduling knob of this implementation. */
typedef struct faldoi_tvl1_params {
    float lambda;      /* data weight                               (ref default 40)    */
    float theta;       /* coupling                                  (ref default 0.3)   */
    float tau;         /* time step                                 (ref default 0.125) */
    float tol;         /* stop when max|du|^2 <= tol*tol            (ref default 0.01)  */
    int warps;         /* outer warps                               (ref default 5)     */
    int max_inner;     /* inner iteration bound per warp            (ref: 400)          */
    int tol_stop;      /* 1 = reference behaviour; 0 = always run max_inner iterations  */
    int chunk;         /* inner iterations per kernel launch, 1..FALDOI_MAX_CHUNK; 0 = auto.
                          Affects speed only: results are bit-identical for every value. */
    int verbose;       /* 1 = print the reference's "Warping: w,Iter: n Error: e" lines  */
} faldoi_tvl1_params;

/* What the reference prints per warp (src/global_faldoi.cpp:803) plus timing. */
typedef struct faldoi_tvl1_stats {
    long total_iters;        /* sum over warps of inner iterations executed              */
    int warps;               /* number of valid entries below (min(params.warps, cap))   */
    int iters_cap;           /* capacity of iters/err given by the caller                */
    int *iters;              /* caller-owned, may be NULL                                */
    float *err;              /* caller-owned, may be NULL                                */
    double solve_ms;         /* device time of the solve (HIP events on the ctx stream)  */
    int chunk_launches;      /* launches of the fused iteration kernel                   */
} faldoi_tvl1_stats;

/* ---------------------------------------------------------------- context */
int faldoi_abi_version(void);
int faldoi_device_count(int *count);
int faldoi_ctx_create(int device, faldoi_ctx **out);
void faldoi_ctx_destroy(faldoi_ctx *ctx);
int faldoi_ctx_synchronize(faldoi_ctx *ctx);
/* The persistent TV-L1 kernel (one launch per solve, see faldoi_tvl1_batch_last_persistent) needs all its workgroups resident at
 * once, so a solve takes it only when nothing else of this process is running on the device (any context, any thread); a second
 * thread's solve runs on the per-launch path meanwhile (`busy`).  Work of OTHER processes on the device is invisible to the
 * library: a launch that waits ~50 ms for a workgroup gives up, the solve is redone per launch (`fallbacks`, a line on stderr),
 * and the context stays on the per-launch path for the next 16 solves (doubling per consecutive give-up up to 1024: `backed_off`).
 * Counters are per context, since its creation; any pointer may be NULL.  faldoi_ctx_persist_retry ends a back-off now. */
int faldoi_ctx_persist_stats(faldoi_ctx *ctx, long *launches, long *fallbacks, long *busy, long *backed_off);
int faldoi_ctx_persist_retry(faldoi_ctx *ctx);
/* TV-L1 solves of this context that were repeated with the general form of the quotients by theta and |grad I|^2, because a divisor
 * turned out to be an even integer >= 6 -- possible only for images that were not normalised to [0, 1] (faldoi_device.hpp: div_once;
 * the result is the reference's either way, the repeat costs one solve and happens once per upload). */
int faldoi_ctx_exact_redos(faldoi_ctx *ctx, long *redos);
/* Host wall time (ms) of this context's last faldoi_tvl1_batch_upload (also inside faldoi_tvl1_global) and faldoi_tvl1_batch_download:
 * the copies from / to the caller's buffers and their synchronisation -- the H2D / D2H figures SURVEY 8d asks to be reported beside
 * the solver's device time (faldoi_tvl1_stats.solve_ms never includes them). */
int faldoi_ctx_last_transfer_ms(faldoi_ctx *ctx, double *h2d_ms, double *d2h_ms);
/* Diagnostic of the fused NLTV kernel's range-checked build (libfaldoi_hip_check.so, compiled with -DFALDOI_NL_CHECK=1; see
 * csrc/nltv_fused_kernel.hpp, nf_check): out[0] = bit per kind of index that left its extent, out[1] = number of violations,
 * out[2] = largest offending value, out[3] = launches checked; reset != 0 clears the words afterwards.  *compiled_in = 1 only in
 * the check build -- in the product library the checks do not exist and the words stay 0. */
int faldoi_debug_nl_check(faldoi_ctx *ctx, unsigned out[4], int reset, int *compiled_in);
/* Thread-local text of the last failure (HIP error string, argument that was rejected). */
const char *faldoi_last_error(void);
void faldoi_tvl1_default_params(faldoi_tvl1_params *p);

/* ---------------------------------------------------------------- one-call solver
 * Replaces  void tvl2OF(const float *I0, float *I1, float *u1, float *u2, float *xi11, float *xi12,
 *                       float *xi21, float *xi22, lambda, theta, tau, tol_OF, nx, ny, warps, verbose)
 *           src/global_faldoi.cpp:556-882.
 * I0,I1: preprocessed gray planes (nx*ny).  u: 2*nx*ny, in = initial flow, out = refined flow.
 * xi: 4*nx*ny (xi11,xi12,xi21,xi22), in/out; NULL means "all zero, do not return".
 * Host pointers.  H2D/D2H copies are outside stats->solve_ms. */
int faldoi_tvl1_global(faldoi_ctx *ctx, const float *I0, const float *I1, float *u, float *xi,
                       int nx, int ny, const faldoi_tvl1_params *prm, faldoi_tvl1_stats *stats);

/* ---------------------------------------------------------------- device-resident batch
 * The same solver on `npairs` independent image pairs of one size, all planes resident in HBM
 * between calls (one batched launch sequence; pairs may stop at different iterations).
 * This is the form bench.py times and the multi-GPU driver shards (pairs are independent,
 * SURVEY.md section 8e).  */
int faldoi_tvl1_batch_create(faldoi_ctx *ctx, int npairs, int nx, int ny, faldoi_tvl1_batch **out);
void faldoi_tvl1_batch_destroy(faldoi_tvl1_batch *b);
/* Host -> HBM.  xi may be NULL (zero).  Resets the pair's state. */
int faldoi_tvl1_batch_upload(faldoi_tvl1_batch *b, int pair, const float *I0, const float *I1,
                             const float *u, const float *xi);
/* Invalidates the last result.  Every batch_solve starts from the flow/duals that were uploaded
 * (kept in their own planes), so a solve can be repeated without uploading again; this call only
 * makes a following download fail until the next solve. */
int faldoi_tvl1_batch_reset(faldoi_tvl1_batch *b);
/* Runs the whole solve for every pair, starting from the uploaded state; stats may be NULL or an
 * array of npairs entries. */
int faldoi_tvl1_batch_solve(faldoi_tvl1_batch *b, const faldoi_tvl1_params *prm,
                            faldoi_tvl1_stats *stats);
/* HBM -> host.  u: 2*nx*ny; xi: 4*nx*ny or NULL. */
int faldoi_tvl1_batch_download(faldoi_tvl1_batch *b, int pair, float *u, float *xi);
/* Timing of the last batch_solve, from HIP events recorded on the context's own stream:
 * solve_ms = the whole solve; chunk_launches / px_iters = launches of the fused iteration kernel
 * and pixel-iterations they advanced; chunk_kernel_ms = sum over those launches of the time
 * between an event pair placed around each one (measured only when the environment variable
 * FALDOI_EVENT_TIMING=1 is set and tol_stop == 0, and always for the one launch of the persistent kernel;
 * otherwise 0).  Any pointer may be NULL. */
int faldoi_tvl1_batch_last_timing(faldoi_tvl1_batch *b, double *solve_ms, int *chunk_launches,
                                  long *px_iters, double *chunk_kernel_ms);

/* Launch schedule the last batch_solve chose: tile = 64 columns x rows_per_thread*waves rows, `chunk` inner
 * iterations per launch of the fused kernel (reporting only; results do not depend on it). */
int faldoi_tvl1_batch_last_schedule(faldoi_tvl1_batch *b, int *rows_per_thread, int *waves, int *chunk);
/* The schedule batch_solve chooses for `npairs` pairs of nx x ny with these parameters on a device with `compute_units`
 * compute units: tile = 64 columns x rows_per_thread*waves rows, `chunk` iterations per launch (or, *persistent = 1, between
 * two halo exchanges of the one persistent launch), `tiles` workgroups.  Host logic only -- no device is touched, so it also
 * answers in a CPU-only build container; results never depend on the schedule.  Any out pointer may be NULL. */
int faldoi_tvl1_schedule_query(int npairs, int nx, int ny, const faldoi_tvl1_params *prm, int compute_units,
                               int *persistent, int *rows_per_thread, int *waves, int *chunk, int *tiles);
/* *persistent = 1 when the last batch_solve ran as ONE launch of the persistent kernel (one pair whose tiles fit the chip one
 * per compute unit: every tile's state stays in registers for the whole solve and only the halo is exchanged between
 * workgroups; with tol_stop != 0 the workgroups also take the reference's stop decision themselves, every `chunk` iterations); chunk_launches is then 1, `chunk` the iterations between two exchanges and
 * chunk_kernel_ms that launch's duration.  The environment variable FALDOI_PERSIST=0 selects the per-launch path; results are
 * identical on both.
 * Which solves take it: ONE pair whose 64 x 48 tiles (54 x 36 owned pixels each) number at most one per compute unit and at most 256
 * -- up to about 0.5 Mpixel on a 256-CU device: 1024 x 436 is 247 tiles; a 1280 x 720 frame (480 tiles) and every batch of several
 * pairs run on the per-launch path, with the same results (faldoi_tvl1_schedule_query answers for any size without a device). */
int faldoi_tvl1_batch_last_persistent(faldoi_tvl1_batch *b, int *persistent);

/* ---------------------------------------------------------------- primitives (host buffers)
 * Stand-alone launches of the device functions the fused kernel is made of, for stage-level
 * parity tests and for callers on either side of the hot path. */
/* src/utils.cpp:367 centered_gradient */
int faldoi_centered_gradient(faldoi_ctx *ctx, const float *in, float *dx, float *dy, int nx, int ny);
/* src/utils.cpp:285 forward_gradient */
int faldoi_forward_gradient(faldoi_ctx *ctx, const float *f, float *fx, float *fy, int nx, int ny);
/* src/utils.cpp:239 divergence */
int faldoi_divergence(faldoi_ctx *ctx, const float *v1, const float *v2, float *div, int nx, int ny);
/* src/bicubic_interpolation.c:245 bicubic_interpolation_warp */
int faldoi_bicubic_warp(faldoi_ctx *ctx, const float *in, const float *u, const float *v, float *out,
                        int nx, int ny, int border_out);

/* ---------------------------------------------------------------- preprocessing
 * Replaces the image preparation in global_faldoi's main, src/global_faldoi.cpp:2049-2068:
 * rgb2gray (:1820) -> image_normalization_3 (src/utils.cpp:743, call-site binding of :2065)
 * -> gaussian(sigma = PRESMOOTHING_SIGMA = 0.9) (src/utils.cpp:521), on the device.
 * i0, i1, im1: planar images with pd channels (1 = gray, 3 = R,G,B planes; iio's "split" layout),
 * values as read from file (0..255).  im1 = I(-1); NULL means "use i1", which is what the
 * reference does for a 2-line list file (:1933-1937).  Outputs: nx*ny planes the solver takes. */
int faldoi_preprocess_global(faldoi_ctx *ctx, const float *i0, const float *i1, const float *im1,
                             int nx, int ny, int pd, float *i0n, float *i1n);
/* Same preparation straight into a batch slot (images never return to the host) plus the
 * initial flow u (2 planes) and duals xi (4 planes or NULL = zero), as faldoi_tvl1_batch_upload. */
int faldoi_tvl1_batch_upload_images(faldoi_tvl1_batch *b, int pair, const float *i0, const float *i1,
                                    const float *im1, int pd, const float *u, const float *xi);

/* ---------------------------------------------------------------- local patch operator (TV-l2 coupled)
 * The operator local_faldoi calls ~3 M times per pair:
 *   initialize_auxiliar_stuff + prepare_stuff   src/energy_model.cpp:202,276   -> faldoi_local_tvl1_create
 *   of_estimation(ofS, ofD, &ener, i0, i1, i_1, index, w, h)   src/energy_model.cpp:691 (default branch:
 *     guided_tvl2coupled src/tvl2_model.cpp:249 + eval_tvl2coupled :174)     -> faldoi_local_tvl1_estimate
 *   free_auxiliar_stuff                         src/energy_model.cpp:240       -> faldoi_local_tvl1_destroy
 * I0, I1: the planes prepare_stuff produces (gray, 2-image normalisation, Gaussian 0.9); the centred
 * gradient of I1 is computed here.  The flow (OpticalFlowData::u1|u2) lives on the device between
 * calls: set_flow / get_flow move all of it.  A patch is get_index_patch's window
 * {ii, ij, ei, ej} (src/utils.cpp:36), 2..11 pixels wide and high; parameters are the ones
 * of_estimation hard-wires (lambda 40, theta 0.3, tau 0.125, tol 0.01, 1 warp).
 * Like the reference operator this one is STATEFUL: duals and both divergence planes persist between
 * calls, and divergence_patch's behaviour of leaving the window's first row / first and last column
 * stale while writing the image's row 0 / columns 0 and nx-1 (src/utils.cpp:88-103) is kept, so a
 * sequence of single-patch calls reproduces the reference bit for bit.  Several patches in one call
 * run concurrently; that equals the sequential order only if their windows (and those writes) do not
 * interact.  ener_out[npatches] = *ener_N; iters_out may be NULL. */
int faldoi_local_tvl1_create(faldoi_ctx *ctx, const float *I0, const float *I1, int nx, int ny,
                             faldoi_local_tvl1 **out);
void faldoi_local_tvl1_destroy(faldoi_local_tvl1 *h);
int faldoi_local_tvl1_set_flow(faldoi_local_tvl1 *h, const float *u);
int faldoi_local_tvl1_get_flow(faldoi_local_tvl1 *h, float *u);
int faldoi_local_tvl1_estimate(faldoi_local_tvl1 *h, int npatches, const int *patches, int max_iter_patch,
                               float *ener_out, int *iters_out);
/* interpolate_poisson(ofD, patch, w), src/local_faldoi.cpp:326-372, = elap_recursive_separable(out, in,
 * w, h, 2, 0.4, 3, 7), src/elap_recsep.c:193-231: inside each window the pixels whose fixed_points
 * entry is not 1 are replaced by a multiscale harmonic extension of the fixed ones (every window
 * pixel is rewritten).  fixed_points: OpticalFlowData::fixed_points, nx*ny ints, uploaded by set_fixed.
 * The relaxation is sequential in the reference, so each (window, channel) runs on one thread, in the
 * reference's order; windows of one call run concurrently (same caveat as estimate). */
int faldoi_local_tvl1_set_fixed(faldoi_local_tvl1 *h, const int *fixed_points);
int faldoi_local_tvl1_interpolate(faldoi_local_tvl1 *h, int npatches, const int *patches);

/* ---------------------------------------------------------------- pyramid resampling
 * src/zoom.c:19-106 (zoom_size, zoom_out, zoom_in).  The reference never compiles or calls
 * zoom.c (src/CMakeLists.txt:45); it is provided because the resampler belongs to the same family
 * of primitives (Gaussian of src/mask.c + bicubic with clamped taps).  Host buffers.
 * zoom_out: Iout has the size faldoi_zoom_size returns; 0 < factor < 1. */
int faldoi_zoom_size(int nx, int ny, float factor, int *nxx, int *nyy);
int faldoi_zoom_out(faldoi_ctx *ctx, const float *I, float *Iout, int nx, int ny, float factor);
int faldoi_zoom_in(faldoi_ctx *ctx, const float *I, float *Iout, int nx, int ny, int nxx, int nyy);

/* ---------------------------------------------------------------- NLTV-CSAD functional (method 6)
 * Replaces  void nltvcsad_PD(const float *I0, float *I1, float *a, int pd, lambda, theta, tau, w, h,
 *                            warps, verbose, float *u1, float *u2)      src/global_faldoi.cpp:1642-1808
 * I0, I1: preprocessed gray planes; a: Lab image of the RAW first frame (pd planes, what
 * image_to_lab returns, :2042-2047); u: 2 contiguous planes, in/out.  400 inner iterations per warp
 * without a tolerance test, as in the reference (:1729); max_inner makes that bound explicit.
 * The neighbour weights (exp/hypot/sqrt of Lab differences) are evaluated on the host with the
 * C library for bit parity and uploaded; all warps and iterations run on the device. */
typedef struct faldoi_nltvcsad_params {
    float lambda;      /* 0.85: main overrides the parameter file for this method (:2138-2140) */
    float theta;       /* 0.3  */
    float tau;         /* 0.1  */
    int warps;         /* -w, default 5 */
    int max_inner;     /* reference: MAX_ITERATIONS_GLOBAL = 400 */
    int verbose;
} faldoi_nltvcsad_params;
void faldoi_nltvcsad_default_params(faldoi_nltvcsad_params *p);
int faldoi_nltvcsad_global(faldoi_ctx *ctx, const float *I0, const float *I1, const float *a, int pd,
                           int nx, int ny, const faldoi_nltvcsad_params *prm, float *u, double *solve_ms);
/* NLTV-L1 (method 2/3): replaces nltvl1_PD, src/global_faldoi.cpp:1177-1327 -- the same non-local
 * regulariser with TV-L1's data term.  Same argument meaning as faldoi_nltvcsad_global; the default
 * parameters are the ones main forces for this method (lambda 2.0, theta 0.3, tau 0.1, :2146-2148). */
void faldoi_nltvl1_default_params(faldoi_nltvcsad_params *p);
int faldoi_nltvl1_global(faldoi_ctx *ctx, const float *I0, const float *I1, const float *a, int pd,
                         int nx, int ny, const faldoi_nltvcsad_params *prm, float *u, double *solve_ms);

/* TV-CSAD, tvcsad_PD, src/global_faldoi.cpp:1449-1636 (global_faldoi -m 4 / 5; main forces lambda 0.85, theta 0.3,
 * tau 0.125, :2153-2156).  Same buffers as tvl2OF: u = 2 planes in/out, xi = 4 planes in/out or NULL (zero).
 * params: lambda/theta/tau/tol/warps/max_inner/verbose are used (tol_stop, chunk are ignored: the tolerance test
 * is always on, as in the reference).  The reference accumulates the stop test's mean |du|^2 racily across its
 * OpenMP threads; this build returns what the reference computes on ONE thread (terms added in pixel order). */
void faldoi_tvcsad_default_params(faldoi_tvl1_params *p);
int faldoi_tvcsad_global(faldoi_ctx *ctx, const float *I0, const float *I1, float *u, float *xi, int nx, int ny,
                         const faldoi_tvl1_params *params, faldoi_tvl1_stats *stats);
/* src/global_faldoi.cpp:906-932 image_to_lab: planar R,G,B (0..255) -> planar L,a*,b* (host, libm). */
int faldoi_image_to_lab(const float *rgb, int nx, int ny, float *lab);

/* ---- the local step: match_growing_variational, src/local_faldoi.cpp:1060-1741 (-m 0 / 1, -split_img 0 / 1 / 2) ----
 * Seeds -> priority-queue growing (forward and backward) -> forward-backward pruning -> re-seeding,
 * `iterations_of` times, then a last forward growing.  Argument order follows the reference function:
 * go / ba = seed flows (2 planes each, NaN = no seed), i0 / i1 = frames (planar, pd channels, 0..255; or
 * already normalised gray planes when params->preprocessed), sal_* = saliency planes or NULL,
 * ene_val (may be NULL) and out_flow (2 planes) = the forward energy map and flow.
 * Every candidate is processed on the device in the reference's order (one workgroup per growing);
 * the batch form runs the growings of several pairs of one size concurrently. */
typedef struct faldoi_local_params {
    int w_radio;          /* -wr,          PAR_DEFAULT_WINSIZE 5   (src/parameters.h:64) */
    int iterations_of;    /* -loc_it,      LOCAL_ITER 3            (:61) */
    int max_iter_patch;   /* -max_pch_it,  MAX_ITERATIONS_LOCAL 4  (:42) */
    float epsilon;        /* -fb_thresh,   FB_TOL 2                (:63) */
    int preprocessed;     /* extension: 1 = i0/i1 are normalised gray planes (skip prepare_stuff's image part) */
    int budget;           /* windows per growing per kernel launch (host relaunches until the queue is empty) */
    int verbose;
    int method;           /* -m: 0 = TV-l2 coupled (M_TVL1), 1 = with weights (M_TVL1_W, src/tvl2w_model.cpp); 2 / 3 = NLTV-L1 (src/nltv_model.cpp) /
                           * weighted; 4 / 5 = TV-CSAD (src/tvcsad_model.cpp) / weighted; 6 / 7 = NLTV-CSAD (src/nltvcsad_model.cpp) / weighted
                           * -- of_estimation's switch, src/energy_model.cpp:691-802.  2..7: whole image only, raw RGB frames for 2, 3, 6, 7 */
    int split_img;        /* -split_img: 1 = grow on image partitions from local iteration 1 on (src/aux_partitions.cpp; every pair of a
                           * batch on its own partitions, all of them in one launch).
                           * 2 (additive, not in the reference) = from iteration 0 on: 2*H*V growings in flight all the way */
    int h_parts;          /* -h_parts, HOR_PARTS 3 (src/parameters.h:69) */
    int v_parts;          /* -v_parts, VER_PARTS 2 (:70) */
    int first_h_parts;    /* split_img 2 only (additive): grid of local iteration 0, first_h_parts x first_v_parts; */
    int first_v_parts;    /* 0 = the transposed grid, like every even iteration.  A coarse first grid keeps most of the accuracy */
    /* -partial_res 1 (src/local_faldoi.cpp:942-1036): every FORWARD growing hands out its flow the first time the share of pixels it
     * has fixed passes 30 / 70 / 80 / 95 % (the reference's test, `percent > P[k] && percent < P[k+1]` on a float percentage, so a
     * threshold can be skipped on tiny domains) and once more when its queue is empty (percent = 100).  flow: 2 planes of w*h, valid
     * during the call; part_idx as the reference passes it to local_growing (-1 whole image inside the iterations, the partition's
     * index, the iteration number for the last whole-image growing).  The executable writes the reference's file names from it.
     * NULL = off.  The growing stops at exactly those counts to copy the flow out, so the result is unchanged. */
    void (*partial_cb)(void *user, int pair, int percent, int iteration, int part_idx, int w, int h, const float *flow);
    void *partial_user;
} faldoi_local_params;
typedef struct faldoi_local_stats {
    long long pops[2];    /* candidates popped, forward / backward (summed over the batch) */
    long long solves[2];  /* windows minimised (of_estimation calls) */
    long long fills[2];   /* windows filled by interpolate_poisson */
    float fb_kept[16];    /* "FB-Chosen" fraction per local iteration: [2*it] forward, [2*it+1] backward */
    float growing_ms;     /* device time from the first seed to the last pop, incl. the host re-seeding */
    int launches;
    long long cycles[3];  /* shader-clock cycles summed over growings: queue (pop + push), Poisson fill, minimisation */
} faldoi_local_stats;
void faldoi_local_default_params(faldoi_local_params *p);
/* Host-only replay of the candidate queue (pq_cand, src/energy_structures.h:13-30), for tests of the pop
 * order among equal energies: ops[k] = 1 pushes energy sims[k] with id k (k < 65536), 0 pops. */
int faldoi_queue_replay(int nops, const int *ops, const float *sims, int *popped, int *npopped);
int faldoi_queue_replay_device(faldoi_ctx *ctx, int nops, const int *ops, const float *sims, int *popped, int *npopped);
/* prepare_stuff's image part for TV-l2 coupled (src/energy_model.cpp:660-684): gray, image_normalization
 * over the two frames (src/utils.cpp:703-734), Gaussian 0.9. */
int faldoi_preprocess_local(faldoi_ctx *ctx, const float *i0, const float *i1, int nx, int ny, int pd, float *i0n,
                            float *i1n);
/* fb_consistency_check(in0, in1, trust_in0, w, h, epsilon), src/local_faldoi.cpp:167-190: in0 / in1 = forward /
 * backward flow (2 planes each); trust[p] = 0 where |in0(x) + in1(x + in0(x))| > epsilon (bicubic, 0 outside),
 * 1 elsewhere; *kept (may be NULL) = number of ones ("FB-Chosen"). */
int faldoi_fb_consistency_check(faldoi_ctx *ctx, const float *in0, const float *in1, int nx, int ny, float epsilon,
                                int *trust, int *kept);
int faldoi_match_growing_variational(faldoi_ctx *ctx, const float *go, const float *ba, const float *i0, const float *i1,
                                     const float *sal_go, const float *sal_ba, int nx, int ny, int pd,
                                     const faldoi_local_params *params, float *ene_val, float *out_flow,
                                     faldoi_local_stats *stats);
int faldoi_match_growing_batch(faldoi_ctx *ctx, int npairs, const float *const *go, const float *const *ba,
                               const float *const *i0, const float *const *i1, const float *const *sal_go,
                               const float *const *sal_ba, int nx, int ny, int pd, const faldoi_local_params *params,
                               float *const *ene_val, float *const *out_flow, faldoi_local_stats *stats);

/* ---------------------------------------------------------------- self-test
 * The kernels replace the IEEE fp32 divide on two call sites by an fp64-assisted sequence that
 * is proven to return the same bits (csrc/faldoi_device.hpp).  This entry point checks the claim
 * on the device: `count` hashed samples over all finite bit patterns; *mismatches must be 0. */
int faldoi_selftest_division(faldoi_ctx *ctx, unsigned seed, unsigned long long count,
                             unsigned long long *mismatches);

#ifdef __cplusplus
}
#endif
#endif /* FALDOI_HIP_H */
