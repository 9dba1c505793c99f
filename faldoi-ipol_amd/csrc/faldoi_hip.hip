// faldoi_hip.hip -- host side of libfaldoi_hip.so (C ABI declared in include/faldoi_hip.h).
//
// Owns: device selection, one HIP stream per context, the HBM slabs of a batch, and the launch
// schedule of the fused TV-L1 kernel.  There is no CPU compute path in this library: every
// entry point either launches HIP kernels or fails with a status code.
#include "faldoi_hip.h"
#include "tvl1_kernels.hpp"
#include "tvl1_persist_kernel.hpp"
#include "tvl1_persist_dyn_kernel.hpp"
#include "preprocess_kernels.hpp"
#include "nltv_kernels.hpp"
#include "nltv_fused_kernel.hpp"
#include "local_kernels.hpp"
#include "grow_kernels.hpp"
#include "grow_twin_kernel.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

using namespace faldoi;

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? FALDOI_ERR_OOM : FALDOI_ERR_HIP,           \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------------ ABI hygiene
namespace {
// Host containers may throw; nothing crosses the C ABI.
// Development hooks.  Everything the library reads from the environment that is NOT product behaviour -- tuning aids (FALDOI_TILE,
// FALDOI_PERSIST_CFG, FALDOI_NL_ROWS, FALDOI_HEAP_TOP_LEVELS, FALDOI_WARP_LDS, FALDOI_WARP_WPS, FALDOI_SYNC_GROUP), measurement aids
// (FALDOI_STAMPS, FALDOI_GROW_DETAIL, FALDOI_REPLAY_CYCLES, FALDOI_EVENT_TIMING, FALDOI_CHUNK_REPEAT, FALDOI_WARP_REPEAT: bench.py times
// kernels with the last two), and test
// hooks that force a rare path (FALDOI_PERSIST_SPINS, FALDOI_GROW_DENSE, FALDOI_RESEED, FALDOI_TVCSAD_SERIAL_SUM, FALDOI_CSAD_SORT,
// FALDOI_NL_CHECK_SELFTEST, FALDOI_SELFTEST_DUMP) -- goes through dev_env(), and ONE compile-time switch removes them all:
// -DFALDOI_DEV_HOOKS=0 builds a library that reads only the three documented switches (FALDOI_PERSIST, FALDOI_EXACT_QUOTIENTS,
// FALDOI_NL_FUSED).  The default build keeps the hooks: the suites and bench.py use them.  (The timing-only what-if
// variants of the kernels are compile-time already: FALDOI_WHATIF, FALDOI_NFWHATIF, FALDOI_HEAP_STAMPS, never defined in a product build.)
#ifndef FALDOI_DEV_HOOKS
#define FALDOI_DEV_HOOKS 1
#endif
static inline const char *dev_env(const char *name) { return FALDOI_DEV_HOOKS ? getenv(name) : nullptr; }

template <typename Fn> int abi_guard(const char *what, Fn fn)
{
    try {
        return fn();
    } catch (const std::bad_alloc &) {
        return fail(FALDOI_ERR_OOM, "out of host memory in %s", what);
    } catch (const std::exception &e) {
        return fail(FALDOI_ERR_STATE, "%s: %s", what, e.what());
    }
}


// Whatever a function enqueued on the stream may still read or write its local host containers when it
// returns early on an error: drain the stream on every exit path (a no-op when it is already idle).
struct StreamDrain {
    hipStream_t s;
    explicit StreamDrain(hipStream_t st) : s(st) {}
    ~StreamDrain() { (void)hipStreamSynchronize(s); }
    StreamDrain(const StreamDrain &) = delete;
    StreamDrain &operator=(const StreamDrain &) = delete;
};

// Runs fn(0 .. ntasks-1) on the host's threads (at most 16: the CPU share of one GPU).  Returns false if a task threw
// (out of host memory): nothing propagates out of a worker thread or across the C ABI.
template <typename Fn> bool grow_host_parallel(int ntasks, Fn fn)
{
    std::atomic<bool> failed(false);
    auto guarded = [&](int task) {
        try { fn(task); } catch (...) { failed = true; }
    };
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads > 16) nthreads = 16;
    if (nthreads < 2 || ntasks < 2) {
        for (int t = 0; t < ntasks; ++t) guarded(t);
        return !failed;
    }
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    try {
        for (unsigned t = 0; t < nthreads && t < (unsigned)ntasks; ++t)
            pool.emplace_back([&] { for (int task; (task = next.fetch_add(1)) < ntasks;) guarded(task); });
    } catch (...) {                                      // could not start (more) threads: the caller's thread finishes the list
        for (int task; (task = next.fetch_add(1)) < ntasks;) guarded(task);
    }
    for (std::thread &t : pool) t.join();
    return !failed;
}

}  // namespace

// ------------------------------------------------------------------------------------------ types
namespace {
struct DevBuf {
    float *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

struct faldoi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int *h_done = nullptr;            // pinned
    int *h_stats = nullptr;           // pinned staging of a solve's per-warp iteration counts and errors (one copy queued behind the
    size_t h_stats_cap = 0;           // solve, one synchronisation for everything), in ints
    int n_cu = 256;
    // What this context has seen of the persistent TV-L1 kernel (tvl1_persist_kernel.hpp), faldoi_ctx_persist_stats:
    long persist_launches = 0;        // solves that ran as one persistent launch
    long exact_redos = 0;             // solves repeated with the general quotients (faldoi_device.hpp: div_once_ok)
    long persist_fallbacks = 0;       // launches that gave up waiting (workgroups not all resident) and were redone per launch
    long persist_busy = 0;            // solves kept off the persistent path because the device had other work of this process
    long persist_backed_off = 0;      // solves kept off it while backing off after a give-up
    int persist_backoff = 0;          // such solves still to come before the persistent path is tried again
    int persist_backoff_len = 0;      // length of the current back-off: 16, doubling per consecutive give-up up to 1024; 0 = none
    // Host wall time of the last batch upload / download on this context (copies from / to the caller's pageable buffers plus the
    // synchronisation), faldoi_ctx_last_transfer_ms: what SURVEY 8d asks to be reported beside the solver's device time
    double last_h2d_ms = 0.0, last_d2h_ms = 0.0;
};

// ------------------------------------------------------------------------------------------ who else is on the device
// The persistent kernel needs every one of its workgroups resident at once (one per CU); whatever else holds CUs while it
// starts can keep some of them out, and two persistent grids can each hold half the chip and wait for the other half.  Inside
// one process the library knows what it has in flight: every compute entry point counts itself in and out per device, and a
// solve takes the persistent path only if it is alone on the device at that moment -- the contract of include/faldoi_hip.h
// ("distinct contexts may be used from distinct host threads") then costs the second thread nothing but the per-launch
// path.  Other processes (or another library) on the same device are not visible here: the bounded spin of the kernel and the
// per-context back-off above cover them.
namespace {
constexpr int kMaxDevices = 64;
struct DeviceGate {
    std::mutex m;
    int active = 0;                   // compute entry points of this library running on the device right now (any context / thread)
    bool persist = false;             // one of them is a persistent TV-L1 grid
};
DeviceGate g_gate[kMaxDevices];

class OpScope {                       // RAII: an entry point's stay on a device
public:
    explicit OpScope(const faldoi_ctx *c) : g_(c ? &g_gate[(unsigned)c->device % kMaxDevices] : nullptr)
    {
        if (!g_) return;
        std::lock_guard<std::mutex> lk(g_->m);
        ++g_->active;
    }
    ~OpScope()
    {
        if (!g_) return;
        std::lock_guard<std::mutex> lk(g_->m);
        --g_->active;
        if (persist_) g_->persist = false;
    }
    bool try_persist()                // true: the caller is alone on the device and now holds the persistent slot
    {
        if (!g_) return false;
        std::lock_guard<std::mutex> lk(g_->m);
        if (!persist_ && g_->active == 1 && !g_->persist) g_->persist = persist_ = true;
        return persist_;
    }
    void release_persist()
    {
        if (!g_ || !persist_) return;
        std::lock_guard<std::mutex> lk(g_->m);
        g_->persist = persist_ = false;
    }
    OpScope(const OpScope &) = delete;
    OpScope &operator=(const OpScope &) = delete;
private:
    DeviceGate *g_;
    bool persist_ = false;
};
}  // namespace

struct faldoi_tvl1_batch {
    faldoi_ctx *ctx = nullptr;
    int npairs = 0, nx = 0, ny = 0;
    size_t plane_elems = 0;
    float *slabs = nullptr;           // [npairs][P_COUNT][plane_elems]
    PairCtl *ctl = nullptr;           // [npairs]
    float *err_part = nullptr;        // [slots][npairs*ntiles][kMaxChunk]
    size_t err_part_elems = 0;
    int *d_iters = nullptr;           // [npairs][warps]
    float *d_err = nullptr;
    int stats_cap = 0;
    int *d_done = nullptr;
    std::vector<int> final_cur;       // which state buffer holds each pair's result
    std::vector<char> uploaded;
    bool exact_quot = false;          // a solve of these images met a divisor the one-product quotient must not be used with
    bool solved = false;
    double last_ms = 0.0;
    double last_chunk_ms = 0.0;       // sum of per-launch event times of the fused kernel (0 = not measured)
    int last_launches = 0;
    int last_tile_r = 0, last_tile_nw = 0, last_chunk = 0;   // schedule of the last solve
    long last_px_iters = 0;
    std::vector<hipEvent_t> ev;       // per-launch event pairs (only when FALDOI_EVENT_TIMING=1)
    float *xslab = nullptr;           // halo exchange slabs of the persistent kernel (tvl1_persist_kernel.hpp)
    size_t xslab_bytes = 0;
    unsigned *xflags = nullptr;       // its epoch flags + abort word, then the error boards of the tolerance-stop form
    size_t xflags_bytes = 0;
    int last_persistent = 0;
    bool persist_off = false;         // the persistent kernel cannot be launched for this batch (e.g. the device's LDS cannot hold the snapshot)
};

static int check_device(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        if (count) *count = 0;
        return fail(FALDOI_ERR_NO_DEVICE, "no HIP device available (%s)",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (count) *count = n;
    return FALDOI_OK;
}

// ------------------------------------------------------------------------------------------ context
extern "C" int faldoi_abi_version(void) { return FALDOI_ABI_VERSION; }

extern "C" const char *faldoi_last_error(void) { return g_err; }

extern "C" int faldoi_device_count(int *count)
{
    if (!count) return fail(FALDOI_ERR_INVALID, "count is NULL");
    return check_device(count);
}

extern "C" void faldoi_tvl1_default_params(faldoi_tvl1_params *p)
{
    if (!p) return;
    p->lambda = FALDOI_DEFAULT_LAMBDA;
    p->theta = FALDOI_DEFAULT_THETA;
    p->tau = FALDOI_DEFAULT_TAU;
    p->tol = FALDOI_DEFAULT_TOL;
    p->warps = FALDOI_DEFAULT_WARPS_GLOBAL;
    p->max_inner = FALDOI_DEFAULT_MAX_INNER;
    p->tol_stop = 1;
    p->chunk = 0;
    p->verbose = 0;
}

extern "C" int faldoi_ctx_create(int device, faldoi_ctx **out)
{
    if (!out) return fail(FALDOI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    int rc = check_device(&n);
    if (rc) return rc;
    if (device < 0 || device >= n) return fail(FALDOI_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    faldoi_ctx *c = new (std::nothrow) faldoi_ctx;
    if (!c) return fail(FALDOI_ERR_OOM, "host allocation failed");
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_done, sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) {
        faldoi_ctx_destroy(c);
        return fail(FALDOI_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return FALDOI_OK;
}

extern "C" void faldoi_ctx_destroy(faldoi_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->h_done) (void)hipHostFree(c->h_done);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    delete c;
}

extern "C" int faldoi_ctx_synchronize(faldoi_ctx *c)
{
    if (!c) return fail(FALDOI_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FALDOI_OK;
}

extern "C" int faldoi_ctx_persist_stats(faldoi_ctx *c, long *launches, long *fallbacks, long *busy, long *backed_off)
{
    if (!c) return fail(FALDOI_ERR_INVALID, "ctx is NULL");
    if (launches) *launches = c->persist_launches;
    if (fallbacks) *fallbacks = c->persist_fallbacks;
    if (busy) *busy = c->persist_busy;
    if (backed_off) *backed_off = c->persist_backed_off;
    return FALDOI_OK;
}

extern "C" int faldoi_ctx_persist_retry(faldoi_ctx *c)
{
    if (!c) return fail(FALDOI_ERR_INVALID, "ctx is NULL");
    c->persist_backoff = 0;
    c->persist_backoff_len = 0;
    return FALDOI_OK;
}

extern "C" int faldoi_ctx_exact_redos(faldoi_ctx *c, long *redos)
{
    if (!c || !redos) return fail(FALDOI_ERR_INVALID, "NULL argument");
    *redos = c->exact_redos;
    return FALDOI_OK;
}

static float *plane_ptr(faldoi_tvl1_batch *b, int pair, int plane);
static int prim_check(faldoi_ctx *ctx, int nx, int ny);

#include "tvl1_probe_host.inc"

// ------------------------------------------------------------------------------------------ tiling
struct TileCfg {
    int R, NW;           // rows per thread, waves per workgroup (tile = 64 x R*NW)
};

static bool tile_supported(int r, int w)
{
    return (r == 4 && (w == 16 || w == 12 || w == 8)) || (r == 3 && (w == 16 || w == 12 || w == 8)) ||
           (r == 8 && w == 8) || (r == 2 && w == 16);
}

// Tile shape and iterations per launch.  Measured on MI355X (tests/tools/sweep.py, 1024x436 pairs):
//   * while the 64x64 / k=10 tiling yields at most one tile per CU (a single pair of this size:
//     240 tiles on 256 CUs) it is the fastest of the launch-per-chunk schedules -- fewest launches, all CUs busy once
//     (a static schedule of such a job runs on the persistent kernel instead, pick_persist);
//   * with more tiles than CUs (batches, large frames) occupancy is no longer the constraint, the halo redundancy and
//     the overlap between workgroups are: two workgroups per CU, each with its own barriers, advanced 5 iterations per launch.
//     Round 2 chose 64x32 tiles of 512 threads (114 / 121 / 135 / 144 Gpx*iter/s at 2 / 3 / 4 / 8 pairs per GPU; 64x36 tiles of
//     768 threads: 96 / 108 / 106 / 114).  Round 3: with the one-product quotients the 64x36 kernel needs 73 registers instead of
//     81, so two of its workgroups fit a CU (6 waves per SIMD), and it is the faster one again: 115 / 136 / 137 / 146 / 154 / 155
//     Gpx*iter/s at 2 / 3 / 4 / 5 / 6 / 8 pairs against 116 / 123 / 136 / 138 / 145 / 145 for 64x32 (tools, same run, mixed pairs).
// An explicit params.chunk or FALDOI_TILE=R,NW (tuning aid) overrides the choice.
static void pick_schedule(const faldoi_tvl1_batch *b, const faldoi_tvl1_params *prm, TileCfg *tc, int *chunk)
{
    const int S = kTileW - 20;                                  // owned pixels per 64-wide tile at k = 10
    const long tiles_big = (long)((b->nx + S - 1) / S) * ((b->ny + S - 1) / S) * b->npairs;
    const bool small_job = tiles_big <= b->ctx->n_cu;
    *tc = small_job ? TileCfg{4, 16} : TileCfg{3, 12};
    int k = prm->chunk ? prm->chunk : (small_job ? 10 : 5);
    if (const char *s = dev_env("FALDOI_TILE")) {
        int r = 0, w = 0;
        if (sscanf(s, "%d,%d", &r, &w) == 2 && tile_supported(r, w)) *tc = TileCfg{r, w};
    }
    if (k > prm->max_inner) k = prm->max_inner;
    while (k > 1 && tc->R * tc->NW - 2 * k < 8) --k;            // keep a useful tile interior
    *chunk = k;
}

template <int R, int NW, bool ONCE>
static hipError_t launch_chunk_t(dim3 grid, hipStream_t s, float *base, size_t pe, int nx, int ny, int ntx, int nty,
                                 int npairs, int halo, TvConst prm, const PairCtl *ctl, float *err, int st, int cur,
                                 int first, int iters, int slot, unsigned long long *stamps)
{
    hipLaunchKernelGGL((tvl1_chunk_kernel<R, NW, ONCE>), grid, dim3(64 * NW), 0, s, base, pe, nx, ny, ntx, nty, npairs, halo,
                       prm, ctl, err, st, cur, first, iters, slot, stamps);
    return hipGetLastError();
}

static hipError_t launch_chunk(TileCfg tc, dim3 grid, hipStream_t s, float *base, size_t pe, int nx, int ny, int ntx,
                               int nty, int npairs, int halo, TvConst prm, const PairCtl *ctl, float *err, int st,
                               int cur, int first, int iters, int slot, unsigned long long *stamps = nullptr)
{
#define FALDOI_CASE(r, w)                                                                                                    \
    if (tc.R == r && tc.NW == w)                                                                                             \
        return prm.once ? launch_chunk_t<r, w, true>(grid, s, base, pe, nx, ny, ntx, nty, npairs, halo, prm, ctl, err, st, cur,  \
                                                     first, iters, slot, stamps)                                             \
                        : launch_chunk_t<r, w, false>(grid, s, base, pe, nx, ny, ntx, nty, npairs, halo, prm, ctl, err, st, cur, \
                                                      first, iters, slot, stamps);
    FALDOI_CASE(4, 16)
    FALDOI_CASE(3, 12)
    FALDOI_CASE(4, 12)
    FALDOI_CASE(3, 16)
    FALDOI_CASE(4, 8)
    FALDOI_CASE(3, 8)
    FALDOI_CASE(2, 16)
    FALDOI_CASE(8, 8)
#undef FALDOI_CASE
    return hipErrorInvalidValue;
}

static void launch_warp(int variant, dim3 grid, hipStream_t s, float *base, size_t pe, int nx, int ny, int npairs,
                        PairCtl *ctl, int st, int cur)
{
    // FALDOI_WARP_LDS=bytes (tuning aid): unused dynamic LDS per workgroup, which caps how many workgroups a CU keeps resident
    // (the kernel's waves all gather, then all compute; fewer resident waves stagger the two phases)
    static const int lds = [] { const char *v = dev_env("FALDOI_WARP_LDS"); const int b = v ? atoi(v) : 0; return b > 0 && b <= 65536 ? b : 0; }();
    if (variant == 8)
        hipLaunchKernelGGL(tvl1_warp_kernel<8>, grid, dim3(256), lds, s, base, pe, nx, ny, npairs, ctl, st, cur);
    else if (variant == 6)
        hipLaunchKernelGGL(tvl1_warp_kernel<6>, grid, dim3(256), lds, s, base, pe, nx, ny, npairs, ctl, st, cur);
    else
        hipLaunchKernelGGL(tvl1_warp_kernel<4>, grid, dim3(256), lds, s, base, pe, nx, ny, npairs, ctl, st, cur);
}

// The persistent kernel (tvl1_persist_kernel.hpp) runs a static schedule of ONE pair whose tiles fit the chip one per CU.
// Geometry: tile = 64 x R*NW, k iterations between two halo exchanges, halo k columns and hy = k rounded up to a multiple of R
// rows (a thread's R rows travel as one 16-byte vector and must share their owner).
struct PersistCfg {
    int R, NW, k, hy;
};

static bool persist_supported(int r, int w)
{
    return (r == 4 && (w == 16 || w == 13 || w == 11)) || (r == 3 && (w == 16 || w == 14));
}

static bool persist_fits(const faldoi_tvl1_batch *b, int R, int NW, int k, PersistCfg *out)
{
    if (k < 1 || k > kMaxChunk) return false;
    const int hy = ((k + R - 1) / R) * R;
    const int Sx = kTileW - 2 * k, Sy = R * NW - 2 * hy;
    if (Sx < k || Sy < hy || Sx < 8 || Sy < R) return false;            // a halo reaches the adjacent tile only
    const long tiles = (long)((b->nx + Sx - 1) / Sx) * ((b->ny + Sy - 1) / Sy);
    if (tiles > b->ctx->n_cu || tiles > 256) return false;                // every workgroup resident, one per CU (256: a wave reads the
                                                                          // tolerance-stop kernel's error board four tiles per lane)
    *out = PersistCfg{ R, NW, k, hy };
    return true;
}

static bool pick_persist(const faldoi_tvl1_batch *b, const faldoi_tvl1_params *prm, PersistCfg *out)
{
    if (const char *on = getenv("FALDOI_PERSIST"))                        // FALDOI_PERSIST=0: always the per-launch path
        if (atoi(on) <= 0) return false;
    if (b->persist_off) return false;                                     // a launch for this batch was refused by the runtime
    if (b->npairs != 1) return false;
    const bool dynamic = prm->tol_stop != 0;                              // tvl1_persist_dyn_kernel.hpp: built for 64x48 tiles only
    if (const char *s = dev_env("FALDOI_PERSIST_CFG")) {                   // tuning aid: R,NW,k
        int r = 0, w = 0, k = 0;
        if (sscanf(s, "%d,%d,%d", &r, &w, &k) == 3 && persist_supported(r, w) && (!dynamic || (r == 3 && w == 16))) {
            if (k > prm->max_inner) k = prm->max_inner;
            if (dynamic && k > 8) return false;
            return persist_fits(b, r, w, k, out);
        }
        return false;
    }
    // 64 x 48 tiles (three rows per thread), five iterations between exchanges: the fastest of the geometries tried at
    // 1024x436 (tools/persist_probe.py sweep), and the only one faster than the per-launch path there; an explicit
    // params.chunk is honoured where a geometry fits it
    static const int cand[][3] = { { 3, 16, 5 }, { 4, 16, 8 } };
    for (const auto &c : cand) {
        int k = prm->chunk ? prm->chunk : c[2];
        if (k > prm->max_inner) k = prm->max_inner;
        if (dynamic && k > 8) return false;                                // its error board is read eight entries per tile
        if (persist_fits(b, c[0], c[1], k, out)) return true;
        if (!prm->chunk || dynamic) break;
    }
    return false;
}

static hipError_t launch_persist(const PersistCfg &pc, dim3 grid, hipStream_t s, const PersistArgs &a)
{
#define FALDOI_CASE(r, w)                                                                          \
    if (pc.R == r && pc.NW == w) {                                                                 \
        if (a.prm.once) hipLaunchKernelGGL((tvl1_persist_kernel<r, w, true>), grid, dim3(64 * w), 0, s, a);   \
        else hipLaunchKernelGGL((tvl1_persist_kernel<r, w, false>), grid, dim3(64 * w), 0, s, a);  \
        return hipGetLastError();                                                                  \
    }
    FALDOI_CASE(3, 16)
    FALDOI_CASE(3, 14)
    FALDOI_CASE(4, 16)
    FALDOI_CASE(4, 13)
    FALDOI_CASE(4, 11)
#undef FALDOI_CASE
    return hipErrorInvalidValue;
}

static hipError_t launch_persist_dyn(const PersistCfg &pc, dim3 grid, hipStream_t s, const PersistArgs &a)
{
    if (pc.R != 3 || pc.NW != 16) return hipErrorInvalidValue;
    const size_t lds = 8ull * 3 * 64 * 16 * sizeof(float);             // the snapshot: 96 KB beside 21 KB static (160 KB per CU)
    // per device, so not cached in a static: a process may hold contexts on several devices
    const void *fn = a.prm.once ? reinterpret_cast<const void *>(&tvl1_persist_dyn_kernel<3, 16, true>)
                                : reinterpret_cast<const void *>(&tvl1_persist_dyn_kernel<3, 16, false>);
    const hipError_t attr = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return attr;
    if (a.prm.once) hipLaunchKernelGGL((tvl1_persist_dyn_kernel<3, 16, true>), grid, dim3(64 * 16), lds, s, a);
    else hipLaunchKernelGGL((tvl1_persist_dyn_kernel<3, 16, false>), grid, dim3(64 * 16), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ batch
static int faldoi_tvl1_batch_create_impl(faldoi_ctx *ctx, int npairs, int nx, int ny, faldoi_tvl1_batch **out)
{
    if (!ctx || !out) return fail(FALDOI_ERR_INVALID, "ctx/out is NULL");
    *out = nullptr;
    if (npairs < 1 || npairs > 256) return fail(FALDOI_ERR_INVALID, "npairs=%d outside 1..256", npairs);
    if (nx < 2 || ny < 2 || (long)nx * ny > (1L << 28))
        return fail(FALDOI_ERR_INVALID, "image size %dx%d unsupported (need >= 2x2, <= 2^28 px)", nx, ny);
    HIP_TRY(hipSetDevice(ctx->device));
    faldoi_tvl1_batch *b = new (std::nothrow) faldoi_tvl1_batch;
    if (!b) return fail(FALDOI_ERR_OOM, "host allocation failed");
    b->ctx = ctx;
    b->npairs = npairs;
    b->nx = nx;
    b->ny = ny;
    b->plane_elems = (((size_t)nx * ny + 63) / 64) * 64;
    b->final_cur.assign(npairs, 0);
    b->uploaded.assign(npairs, 0);
    const size_t bytes = (size_t)npairs * P_COUNT * b->plane_elems * sizeof(float);
    hipError_t e = hipMalloc((void **)&b->slabs, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(b->slabs, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&b->ctl, sizeof(PairCtl) * npairs);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_done, sizeof(int));
    if (e != hipSuccess) {
        faldoi_tvl1_batch_destroy(b);
        return fail(e == hipErrorOutOfMemory ? FALDOI_ERR_OOM : FALDOI_ERR_HIP, "batch allocation of %zu bytes failed: %s",
                    bytes, hipGetErrorString(e));
    }
    *out = b;
    return FALDOI_OK;
}

extern "C" int faldoi_tvl1_batch_create(faldoi_ctx *ctx, int npairs, int nx, int ny, faldoi_tvl1_batch **out)
{
    return abi_guard("the batch allocation", [&] { return faldoi_tvl1_batch_create_impl(ctx, npairs, nx, ny, out); });
}

extern "C" void faldoi_tvl1_batch_destroy(faldoi_tvl1_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
    if (b->slabs) (void)hipFree(b->slabs);
    if (b->ctl) (void)hipFree(b->ctl);
    if (b->err_part) (void)hipFree(b->err_part);
    if (b->d_iters) (void)hipFree(b->d_iters);
    if (b->d_err) (void)hipFree(b->d_err);
    if (b->d_done) (void)hipFree(b->d_done);
    if (b->xslab) (void)hipFree(b->xslab);
    if (b->xflags) (void)hipFree(b->xflags);
    delete b;
}

static float *plane_ptr(faldoi_tvl1_batch *b, int pair, int plane)
{
    return b->slabs + ((size_t)pair * P_COUNT + plane) * b->plane_elems;
}

static int faldoi_tvl1_batch_upload_impl(faldoi_tvl1_batch *b, int pair, const float *I0, const float *I1,
                                        const float *u, const float *xi)
{
    if (!b || !I0 || !I1 || !u) return fail(FALDOI_ERR_INVALID, "NULL argument");
    if (pair < 0 || pair >= b->npairs) return fail(FALDOI_ERR_INVALID, "pair %d out of range", pair);
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t s = b->ctx->stream;
    const size_t n = (size_t)b->nx * b->ny, nb = n * sizeof(float);
    const auto t_up = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(plane_ptr(b, pair, P_I0), I0, nb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(plane_ptr(b, pair, P_I1), I1, nb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(plane_ptr(b, pair, P_INIT + 0), u, nb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(plane_ptr(b, pair, P_INIT + 1), u + n, nb, hipMemcpyHostToDevice, s));
    for (int k = 0; k < 4; ++k) {
        if (xi)
            HIP_TRY(hipMemcpyAsync(plane_ptr(b, pair, P_INIT + 2 + k), xi + k * n, nb, hipMemcpyHostToDevice, s));
        else
            HIP_TRY(hipMemsetAsync(plane_ptr(b, pair, P_INIT + 2 + k), 0, nb, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    b->ctx->last_h2d_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up).count();
    b->uploaded[pair] = 1;
    b->exact_quot = false;
    b->solved = false;
    return FALDOI_OK;
}

extern "C" int faldoi_tvl1_batch_upload(faldoi_tvl1_batch *b, int pair, const float *I0, const float *I1,
                                        const float *u, const float *xi)
{
    return abi_guard("the batch upload", [&] { return faldoi_tvl1_batch_upload_impl(b, pair, I0, I1, u, xi); });
}

extern "C" int faldoi_tvl1_batch_reset(faldoi_tvl1_batch *b)
{
    if (!b) return fail(FALDOI_ERR_INVALID, "batch is NULL");
    b->solved = false;      // batch_solve always restarts from the uploaded state
    return FALDOI_OK;
}

static int ensure(void **p, size_t *have, size_t want_bytes)
{
    if (*p && *have >= want_bytes) return FALDOI_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(p, want_bytes));
    *have = want_bytes;
    return FALDOI_OK;
}

static int faldoi_tvl1_batch_solve_impl(faldoi_tvl1_batch *b, const faldoi_tvl1_params *prm, faldoi_tvl1_stats *stats, OpScope &scope)
{
    if (!b || !prm) return fail(FALDOI_ERR_INVALID, "NULL argument");
    for (int i = 0; i < b->npairs; ++i)
        if (!b->uploaded[i]) return fail(FALDOI_ERR_STATE, "pair %d was never uploaded", i);
    if (!(prm->lambda > 0) || !(prm->theta > 0) || !(prm->tau > 0) || !(prm->tol >= 0))
        return fail(FALDOI_ERR_INVALID, "lambda/theta/tau must be > 0 and tol >= 0");
    if (prm->warps < 1 || prm->warps > 100000) return fail(FALDOI_ERR_INVALID, "warps=%d outside 1..100000", prm->warps);
    if (prm->max_inner < 1 || prm->max_inner > 1000000)
        return fail(FALDOI_ERR_INVALID, "max_inner=%d outside 1..1000000", prm->max_inner);
    if (prm->chunk < 0 || prm->chunk > FALDOI_MAX_CHUNK)
        return fail(FALDOI_ERR_INVALID, "chunk=%d outside 0..%d", prm->chunk, FALDOI_MAX_CHUNK);

    faldoi_ctx *c = b->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    b->solved = false;                          // a solve that fails midway leaves nothing to download
    StreamDrain drain(s);                       // no early return leaves work in flight behind local host buffers
    TileCfg tc;
    int chunk;
    pick_schedule(b, prm, &tc, &chunk);
    PersistCfg pc{ 0, 0, 0, 0 };
    bool persist = pick_persist(b, prm, &pc);
    if (persist && c->persist_backoff > 0) {    // a recent launch gave up (its workgroups were not all resident): per-launch path for a while
        --c->persist_backoff;
        ++c->persist_backed_off;
        persist = false;
    }
    if (persist && !scope.try_persist()) {      // something else of this process is on the device (another thread's solve or growing)
        ++c->persist_busy;
        persist = false;
    }
    if (persist) {
        tc = TileCfg{ pc.R, pc.NW };
        chunk = pc.k;
    }
    const int TH = tc.R * tc.NW;
    const int halo = chunk;
    const int Sx = kTileW - 2 * halo, Sy = TH - 2 * (persist ? pc.hy : halo);
    const int ntx = (b->nx + Sx - 1) / Sx, nty = (b->ny + Sy - 1) / Sy;
    const int ntiles = ntx * nty;
    const int total_tiles = ntiles * b->npairs;
    const dim3 grid(8 * ((total_tiles + 7) / 8));
    const int warps = prm->warps;
    const bool dynamic = prm->tol_stop != 0;

    TvConst k;
    k.rtheta = 1.0 / (double)prm->theta;
    // The quotients by theta and |grad I|^2 as one fp64 product each (faldoi_device.hpp: div_once) -- unless a divisor is an even
    // integer >= 6 (div_once_ok).  theta is tested here; the kernels that compute warp constants report such a |grad I|^2 (images
    // that were never normalised) in PairCtl::unsafe, and the solve is then repeated below in the general form, which this batch
    // keeps until its next upload.  FALDOI_EXACT_QUOTIENTS=1 selects the general form from the start (tests).
    static const bool env_exact = [] { const char *v = getenv("FALDOI_EXACT_QUOTIENTS"); return v && atoi(v) > 0; }();
    k.once = !(prm->theta >= 6.0f && std::fmod(0.5f * prm->theta, 1.0f) == 0.0f) && !b->exact_quot && !env_exact;
    k.l_t = prm->lambda * prm->theta;          // src/global_faldoi.cpp:577
    k.theta = prm->theta;
    k.tau = prm->tau;
    k.tol2 = prm->tol * prm->tol;              // :684
    k.max_inner = prm->max_inner;
    k.tol_stop = prm->tol_stop;
    k.warps = warps;
    k.chunk = chunk;

    // workspaces
    {
        const size_t slots = dynamic ? 1 : (size_t)warps;
        size_t have = b->err_part_elems * sizeof(float);
        int rc = ensure((void **)&b->err_part, &have, slots * total_tiles * kMaxChunk * sizeof(float));
        if (rc) return rc;
        b->err_part_elems = have / sizeof(float);
        if (b->stats_cap < warps * b->npairs) {
            if (b->d_iters) (void)hipFree(b->d_iters);
            if (b->d_err) (void)hipFree(b->d_err);
            b->d_iters = nullptr;
            b->d_err = nullptr;
            b->stats_cap = 0;
            HIP_TRY(hipMalloc((void **)&b->d_iters, sizeof(int) * warps * b->npairs));
            HIP_TRY(hipMalloc((void **)&b->d_err, sizeof(float) * warps * b->npairs));
            b->stats_cap = warps * b->npairs;
        }
    }

    const int npix = b->nx * b->ny;
    const dim3 wgrid(((npix + 255) / 256) * b->npairs);
    int wvariant = 4;                                  // waves/SIMD the per-warp kernel is compiled for
    if (const char *v = dev_env("FALDOI_WARP_WPS")) wvariant = atoi(v);
    int chunk_repeat = 1;
    if (const char *v = dev_env("FALDOI_CHUNK_REPEAT")) { const int r = atoi(v); if (r >= 1 && r <= 8) chunk_repeat = r; }
    int warp_repeat = 1;
    if (const char *v = dev_env("FALDOI_WARP_REPEAT")) { const int r = atoi(v); if (r >= 1 && r <= 8) warp_repeat = r; }
    int launches = 0;
    HIP_TRY(hipEventRecord(c->ev0, s));
    HIP_TRY(hipMemsetAsync(b->d_done, 0, sizeof(int), s));
    {
        const int blocks = (int)(((size_t)npix * b->npairs + 255) / 256);
        hipLaunchKernelGGL(reset_state_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, s, b->slabs,
                           b->plane_elems, npix, b->npairs, b->ctl, chunk);
        HIP_TRY(hipGetLastError());
        // src/global_faldoi.cpp:611: centred gradient of I1, once per solve
        hipLaunchKernelGGL(centered_gradient_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, s,
                           plane_ptr(b, 0, P_I1), plane_ptr(b, 0, P_I1X), plane_ptr(b, 0, P_I1Y), b->nx, b->ny,
                           (size_t)P_COUNT * b->plane_elems, (size_t)P_COUNT * b->plane_elems, b->npairs);
        HIP_TRY(hipGetLastError());
    }

    // Optional per-launch timing of the fused kernel: one HIP event pair around every launch on the
    // context's stream (FALDOI_EVENT_TIMING=1; static schedule only).  Off by default: the events
    // add a little stream traffic, so the headline wall-clock is taken without them.
    const bool ev_timing = !dynamic && dev_env("FALDOI_EVENT_TIMING") && atoi(dev_env("FALDOI_EVENT_TIMING")) > 0;
    size_t ev_used = 0;
    if (ev_timing) {
        const size_t need = 2 * (size_t)warps * ((prm->max_inner + chunk - 1) / chunk);
        while (b->ev.size() < need) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            b->ev.push_back(e);
        }
    }
    // FALDOI_STAMPS=1 (measurement aid): one launch in the middle of the solve records, per workgroup, when it started, when
    // its loads had landed, when its iterations ended and when its stores had drained (constant 100 MHz clock)
    unsigned long long *d_stamps = nullptr;
    DevBuf stamp_buf;
    if (!dynamic && !persist && dev_env("FALDOI_STAMPS") && atoi(dev_env("FALDOI_STAMPS")) > 0) {
        HIP_TRY(hipMalloc((void **)&stamp_buf.p, sizeof(unsigned long long) * 4 * grid.x));
        d_stamps = reinterpret_cast<unsigned long long *>(stamp_buf.p);
        HIP_TRY(hipMemsetAsync(d_stamps, 0, sizeof(unsigned long long) * 4 * grid.x, s));
    }
    unsigned long long *p_stamps_out = nullptr;
    if (persist) {
        const size_t slab_bytes = 2ull * ntiles * kXPlanes * tc.NW * kTileW * 16ull;
        const size_t flag_words = ((size_t)ntiles + 1) * kFlagStride;    // epoch flags, abort word
        const size_t flag_bytes = (flag_words + (dynamic ? 2ull * ntiles * kMaxChunk : 0)) * sizeof(unsigned);   // + two error boards
        if (slab_bytes >= (1ull << 32)) return fail(FALDOI_ERR_STATE, "exchange slab of %zu bytes", slab_bytes);
        int rc = ensure((void **)&b->xslab, &b->xslab_bytes, slab_bytes);
        if (rc) return rc;
        rc = ensure((void **)&b->xflags, &b->xflags_bytes, flag_bytes);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(b->xflags, 0, flag_bytes, s));      // epochs count from 0 in every launch
        unsigned long long *p_stamps = nullptr;
        if (dev_env("FALDOI_STAMPS") && atoi(dev_env("FALDOI_STAMPS")) > 0) {
            HIP_TRY(hipMalloc((void **)&stamp_buf.p, sizeof(unsigned long long) * kPersistStamps * ntiles));
            p_stamps = reinterpret_cast<unsigned long long *>(stamp_buf.p);
            HIP_TRY(hipMemsetAsync(p_stamps, 0, sizeof(unsigned long long) * kPersistStamps * ntiles, s));
        }
        PersistArgs a;
        a.base = b->slabs;
        a.plane_elems = b->plane_elems;
        a.nx = b->nx; a.ny = b->ny; a.ntx = ntx; a.nty = nty;
        a.hx = chunk; a.hy = pc.hy;
        // the rows the grid has to spare go half to the first tile row, half to the last (PersistArgs::yshift); FALDOI_PERSIST_YSHIFT=0
        // (development hook) keeps the grid flush with the image's first row, as in rounds 2 and 3
        a.yshift = (((nty * Sy - b->ny) / 2) / pc.R) * pc.R;
        if (const char *v = dev_env("FALDOI_PERSIST_YSHIFT")) {
            const int q = atoi(v);
            if (q >= 0 && q % pc.R == 0 && q <= nty * Sy - b->ny) a.yshift = q;
        }
        a.prm = k;
        a.err_out = b->err_part;
        a.xslab = b->xslab;
        a.xslab_bytes = (unsigned)slab_bytes;
        a.flags = b->xflags;
        // ~1.2 us per poll: 40 000 polls = ~50 ms.  A neighbour lags by one chunk of iterations at most (tens of microseconds), so this
        // long a wait means a workgroup that is not resident, not a slow one.  FALDOI_PERSIST_SPINS=0 (test hook): every tile gives
        // up at its first exchange, before it polls
        a.spin_limit = 40000u;
        if (const char *v = dev_env("FALDOI_PERSIST_SPINS")) { const long q = atol(v); if (q >= 0 && q < (1L << 31)) a.spin_limit = (unsigned)q; }
        a.stamps = p_stamps;
        a.stamp_warp = warps / 2;
        a.unsafe = &b->ctl[0].unsafe;
        a.err_board = b->xflags + flag_words;
        a.iters_out = b->d_iters;
        a.werr_out = b->d_err;
        while (b->ev.size() < 2) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            b->ev.push_back(e);
        }
        HIP_TRY(hipEventRecord(b->ev[0], s));
        if ((dynamic ? launch_persist_dyn(pc, grid, s, a) : launch_persist(pc, grid, s, a)) != hipSuccess) {
            (void)hipGetLastError();                               // e.g. a device whose LDS cannot hold the snapshot: per-launch path
            b->persist_off = true;
            scope.release_persist();
            return faldoi_tvl1_batch_solve_impl(b, prm, stats, scope);
        }
        HIP_TRY(hipEventRecord(b->ev[1], s));
        ev_used = 2;
        launches = 1;
        if (!dynamic) {                                            // (the tolerance-stop kernel writes iterations and errors itself)
            const int rem = prm->max_inner % chunk;
            hipLaunchKernelGGL(tvl1_static_stats_kernel, dim3(warps * b->npairs), dim3(256), 0, s, b->err_part, ntiles,
                               b->npairs, warps, prm->max_inner, rem ? rem : chunk, b->d_iters, b->d_err);
            HIP_TRY(hipGetLastError());
        }
        b->final_cur[0] = 1;
        HIP_TRY(hipMemcpyAsync(c->h_done, b->xflags + (size_t)ntiles * kFlagStride, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        p_stamps_out = p_stamps;
    } else if (!dynamic) {
        int cur = 0;
        int last_iters = chunk;
        for (int w = 0; w < warps; ++w) {
            int n = 0;
            while (n < prm->max_inner) {
                const int it = (prm->max_inner - n) < chunk ? (prm->max_inner - n) : chunk;
                if (n == 0) {
                    // FALDOI_WARP_REPEAT=r (measurement aid): the per-warp kernel is idempotent -- it reads the flow and
                    // writes the warp constants only -- so launching it r times changes no result; the difference of two
                    // whole-solve times isolates its duration (bench.py)
                    for (int r = 0; r < warp_repeat; ++r)
                        launch_warp(wvariant, wgrid, s, b->slabs, b->plane_elems, b->nx, b->ny, b->npairs, b->ctl, 1, cur);
                    HIP_TRY(hipGetLastError());
                }
                if (ev_timing) HIP_TRY(hipEventRecord(b->ev[ev_used++], s));
                // FALDOI_CHUNK_REPEAT=r (measurement aid, like FALDOI_WARP_REPEAT): a launch reads one state buffer and writes
                // the other, so repeating it with the same arguments rewrites the same values
                for (int r = 0; r < chunk_repeat; ++r)
                    HIP_TRY(launch_chunk(tc, grid, s, b->slabs, b->plane_elems, b->nx, b->ny, ntx, nty, b->npairs, halo, k,
                                         b->ctl, b->err_part, 1, cur, (n == 0 ? 1 : 0) | (n + it >= prm->max_inner ? 2 : 0), it, w,
                                         (d_stamps && w == warps / 2 && n == 0) ? d_stamps : nullptr));
                if (ev_timing) HIP_TRY(hipEventRecord(b->ev[ev_used++], s));
                ++launches;
                cur ^= 1;
                n += it;
                last_iters = it;
            }
        }
        hipLaunchKernelGGL(tvl1_static_stats_kernel, dim3(warps * b->npairs), dim3(256), 0, s, b->err_part, ntiles,
                           b->npairs, warps, prm->max_inner, last_iters, b->d_iters, b->d_err);
        HIP_TRY(hipGetLastError());
        for (int i = 0; i < b->npairs; ++i) b->final_cur[i] = cur;
    } else {
        int group = 8;
        if (const char *g = dev_env("FALDOI_SYNC_GROUP")) { int v = atoi(g); if (v >= 1 && v <= 1024) group = v; }
        const long max_launches = 2L * warps * ((prm->max_inner + chunk - 1) / chunk + 1) + group;
        for (;;) {
            for (int g = 0; g < group; ++g) {
                launch_warp(wvariant, wgrid, s, b->slabs, b->plane_elems, b->nx, b->ny, b->npairs, b->ctl, 0, 0);
                HIP_TRY(hipGetLastError());
                HIP_TRY(launch_chunk(tc, grid, s, b->slabs, b->plane_elems, b->nx, b->ny, ntx, nty, b->npairs, halo, k,
                                     b->ctl, b->err_part, 0, 0, 0, 0, 0));
                hipLaunchKernelGGL(tvl1_ctl_kernel, dim3(b->npairs), dim3(256), 0, s, b->ctl, b->err_part, ntiles, k,
                                   b->d_iters, b->d_err, b->d_done);
                HIP_TRY(hipGetLastError());
                ++launches;
            }
            HIP_TRY(hipMemcpyAsync(c->h_done, b->d_done, sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (*c->h_done >= b->npairs) break;
            if (launches > max_launches) return fail(FALDOI_ERR_STATE, "solver state machine did not terminate");
        }
        std::vector<PairCtl> hc(b->npairs);
        HIP_TRY(hipMemcpyAsync(hc.data(), b->ctl, sizeof(PairCtl) * b->npairs, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int i = 0; i < b->npairs; ++i) b->final_cur[i] = hc[i].cur;
    }
    HIP_TRY(hipEventRecord(c->ev1, s));
    // the per-warp iteration counts and errors follow the solve down the same stream into pinned memory: one wait for everything
    const size_t nstat = (size_t)warps * b->npairs;
    if (c->h_stats_cap < 2 * nstat + b->npairs) {
        if (c->h_stats) (void)hipHostFree(c->h_stats);
        c->h_stats = nullptr;
        c->h_stats_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&c->h_stats, sizeof(int) * (2 * nstat + b->npairs), hipHostMallocDefault));
        c->h_stats_cap = 2 * nstat + b->npairs;
    }
    HIP_TRY(hipMemcpyAsync(c->h_stats, b->d_iters, sizeof(int) * nstat, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(c->h_stats + nstat, b->d_err, sizeof(float) * nstat, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpy2DAsync(c->h_stats + 2 * nstat, sizeof(int), &b->ctl[0].unsafe, sizeof(PairCtl), sizeof(int), b->npairs,
                             hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    double chunk_ms = 0.0;
    for (size_t i = 0; i + 1 < ev_used; i += 2) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, b->ev[i], b->ev[i + 1]));
        chunk_ms += t;
    }

    if (persist) scope.release_persist();       // the grid has drained (the event above): the device is free for the next one
    if (persist && *c->h_done != 0) {
        // Bounded spins: a tile that waited ~50 ms for a neighbour raised the abort word and every workgroup left.  The cause is
        // residency (something else held CUs this grid needed -- another process, a CU mask), not the data: the solve is redone on
        // the per-launch path, and this context stays there for the next 16 solves (doubling per consecutive give-up, up to 1024)
        // before it tries again.  Counted in faldoi_ctx_persist_stats; said on stderr every time (the back-off bounds how often).
        ++c->persist_fallbacks;
        c->persist_backoff_len = c->persist_backoff_len ? (c->persist_backoff_len < 1024 ? 2 * c->persist_backoff_len : 1024) : 16;
        c->persist_backoff = c->persist_backoff_len + 1;           // + the redo below ...
        --c->persist_backed_off;                                   // ... which is this same solve, not one more that was kept off the path
        fprintf(stderr, "faldoi: the persistent TV-L1 kernel gave up waiting for a neighbouring tile (first at tile %d): its workgroups "
                        "were not all resident; using the per-launch path for this and the next %d solves of this context (give-up %ld)\n",
                *c->h_done - 1, c->persist_backoff_len, c->persist_fallbacks);
        return faldoi_tvl1_batch_solve_impl(b, prm, stats, scope);
    }
    if (persist) {
        ++c->persist_launches;
        c->persist_backoff_len = 0;
    }
    if (k.once) {
        bool unsafe = false;
        for (int i = 0; i < b->npairs; ++i) unsafe = unsafe || c->h_stats[2 * nstat + i] != 0;
        if (unsafe) {                             // a |grad I|^2 that is an even integer >= 6: the general quotient, from the start
            b->exact_quot = true;
            ++c->exact_redos;
            return faldoi_tvl1_batch_solve_impl(b, prm, stats, scope);
        }
    }
    if (p_stamps_out) { const int rc = probe_report_persistent(p_stamps_out, ntiles, ntx, nty, warps); if (rc) return rc; }   // tvl1_probe_host.inc
    if (d_stamps) { const int rc = probe_report_chunk(d_stamps, total_tiles, ntx, nty); if (rc) return rc; }
    const int *h_it = c->h_stats;
    const float *h_er = reinterpret_cast<const float *>(c->h_stats + nstat);
    long px_iters = 0;
    for (int i = 0; i < b->npairs; ++i) {
        long tot = 0;
        for (int w = 0; w < warps; ++w) tot += h_it[(size_t)i * warps + w];
        px_iters += tot * (long)npix;
        if (stats) {
            faldoi_tvl1_stats *st = &stats[i];
            st->total_iters = tot;
            st->solve_ms = ms;
            st->chunk_launches = launches;
            const int nrep = warps < st->iters_cap ? warps : st->iters_cap;
            st->warps = nrep;
            for (int w = 0; w < nrep; ++w) {
                if (st->iters) st->iters[w] = h_it[(size_t)i * warps + w];
                if (st->err) st->err[w] = h_er[(size_t)i * warps + w];
            }
        }
        if (prm->verbose)
            for (int w = 0; w < warps; ++w)      // same text as src/global_faldoi.cpp:803
                fprintf(stderr, "Warping: %d,Iter: %d Error: %f\n", w, h_it[(size_t)i * warps + w],
                        h_er[(size_t)i * warps + w]);
    }
    b->solved = true;
    b->last_ms = ms;
    b->last_chunk_ms = chunk_ms;
    b->last_launches = launches;
    b->last_tile_r = tc.R;
    b->last_tile_nw = tc.NW;
    b->last_chunk = chunk;
    b->last_persistent = persist ? 1 : 0;
    b->last_px_iters = px_iters;
    return FALDOI_OK;
}

extern "C" int faldoi_tvl1_batch_solve(faldoi_tvl1_batch *b, const faldoi_tvl1_params *prm, faldoi_tvl1_stats *stats)
{
    return abi_guard("the TV-L1 solve", [&] {
        OpScope scope(b ? b->ctx : nullptr);
        return faldoi_tvl1_batch_solve_impl(b, prm, stats, scope);
    });
}

extern "C" int faldoi_tvl1_batch_last_timing(faldoi_tvl1_batch *b, double *solve_ms, int *chunk_launches, long *px_iters,
                                             double *chunk_kernel_ms)
{
    if (!b) return fail(FALDOI_ERR_INVALID, "batch is NULL");
    if (!b->solved) return fail(FALDOI_ERR_STATE, "no solve has completed");
    if (solve_ms) *solve_ms = b->last_ms;
    if (chunk_launches) *chunk_launches = b->last_launches;
    if (px_iters) *px_iters = b->last_px_iters;
    if (chunk_kernel_ms) *chunk_kernel_ms = b->last_chunk_ms;
    return FALDOI_OK;
}

extern "C" int faldoi_tvl1_schedule_query(int npairs, int nx, int ny, const faldoi_tvl1_params *prm, int compute_units,
                                          int *persistent, int *rows_per_thread, int *waves, int *chunk, int *tiles)
{
    return abi_guard("the schedule query", [&] {
        if (!prm) return fail(FALDOI_ERR_INVALID, "params is NULL");
        if (npairs < 1 || nx < 2 || ny < 2 || compute_units < 1) return fail(FALDOI_ERR_INVALID, "npairs, size and compute units must be positive");
        faldoi_ctx c;                                   // host-only stand-ins: the choice depends on sizes and the CU count alone
        c.n_cu = compute_units;
        faldoi_tvl1_batch b;
        b.ctx = &c;
        b.npairs = npairs;
        b.nx = nx;
        b.ny = ny;
        TileCfg tc;
        int k = 0;
        pick_schedule(&b, prm, &tc, &k);
        PersistCfg pc{ 0, 0, 0, 0 };
        const bool persist = pick_persist(&b, prm, &pc);
        if (persist) { tc = TileCfg{ pc.R, pc.NW }; k = pc.k; }
        const int Sx = kTileW - 2 * k, Sy = tc.R * tc.NW - 2 * (persist ? pc.hy : k);
        if (persistent) *persistent = persist ? 1 : 0;
        if (rows_per_thread) *rows_per_thread = tc.R;
        if (waves) *waves = tc.NW;
        if (chunk) *chunk = k;
        if (tiles) *tiles = ((nx + Sx - 1) / Sx) * ((ny + Sy - 1) / Sy) * npairs;
        return (int)FALDOI_OK;
    });
}

extern "C" int faldoi_tvl1_batch_last_persistent(faldoi_tvl1_batch *b, int *persistent)
{
    if (!b || !persistent) return fail(FALDOI_ERR_INVALID, "NULL argument");
    if (!b->solved) return fail(FALDOI_ERR_STATE, "no solve has completed");
    *persistent = b->last_persistent;
    return FALDOI_OK;
}

extern "C" int faldoi_tvl1_batch_last_schedule(faldoi_tvl1_batch *b, int *rows_per_thread, int *waves, int *chunk)
{
    if (!b) return fail(FALDOI_ERR_INVALID, "batch is NULL");
    if (!b->solved) return fail(FALDOI_ERR_STATE, "no solve has completed");
    if (rows_per_thread) *rows_per_thread = b->last_tile_r;
    if (waves) *waves = b->last_tile_nw;
    if (chunk) *chunk = b->last_chunk;
    return FALDOI_OK;
}

extern "C" int faldoi_tvl1_batch_download(faldoi_tvl1_batch *b, int pair, float *u, float *xi)
{
    if (!b || !u) return fail(FALDOI_ERR_INVALID, "NULL argument");
    if (pair < 0 || pair >= b->npairs) return fail(FALDOI_ERR_INVALID, "pair %d out of range", pair);
    if (!b->solved) return fail(FALDOI_ERR_STATE, "download before solve");
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t s = b->ctx->stream;
    const size_t n = (size_t)b->nx * b->ny, nb = n * sizeof(float);
    const int st = b->final_cur[pair] ? P_STATE1 : P_STATE0;
    const auto t_down = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(u, plane_ptr(b, pair, st + 0), nb, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(u + n, plane_ptr(b, pair, st + 1), nb, hipMemcpyDeviceToHost, s));
    if (xi)
        for (int k = 0; k < 4; ++k)
            HIP_TRY(hipMemcpyAsync(xi + k * n, plane_ptr(b, pair, st + 4 + k), nb, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    b->ctx->last_d2h_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_down).count();
    return FALDOI_OK;
}

extern "C" int faldoi_ctx_last_transfer_ms(faldoi_ctx *c, double *h2d_ms, double *d2h_ms)
{
    if (!c) return fail(FALDOI_ERR_INVALID, "ctx is NULL");
    if (h2d_ms) *h2d_ms = c->last_h2d_ms;
    if (d2h_ms) *d2h_ms = c->last_d2h_ms;
    return FALDOI_OK;
}

// ------------------------------------------------------------------------------------------ one-call solver
extern "C" int faldoi_tvl1_global(faldoi_ctx *ctx, const float *I0, const float *I1, float *u, float *xi, int nx, int ny,
                                  const faldoi_tvl1_params *prm, faldoi_tvl1_stats *stats)
{
    if (!ctx || !I0 || !I1 || !u || !prm) return fail(FALDOI_ERR_INVALID, "NULL argument");
    faldoi_tvl1_batch *b = nullptr;
    int rc = faldoi_tvl1_batch_create(ctx, 1, nx, ny, &b);
    if (rc) return rc;
    rc = faldoi_tvl1_batch_upload(b, 0, I0, I1, u, xi);
    if (!rc) rc = faldoi_tvl1_batch_solve(b, prm, stats);
    if (!rc) rc = faldoi_tvl1_batch_download(b, 0, u, xi);
    faldoi_tvl1_batch_destroy(b);
    return rc;
}

// ------------------------------------------------------------------------------------------ preprocessing
// Normalised Gaussian half-kernel exactly as src/utils.cpp:529-552 builds it in its C++ translation
// unit: the exponent argument is fp32 and exp() resolves to the fp32 overload, the prefactor and
// the product are fp64, the normalisation accumulates in fp32.
static int gaussian_taps(float sigma, GaussTaps *t)
{
    const float den = 2 * sigma * sigma;
    const int size = (int)(5 * sigma) + 1;
    if (!(sigma > 0) || size > kMaxGaussTaps) return -1;
    for (int i = 0; i < size; i++) {
        const float arg = -i * i / den;
        t->B[i] = (float)(1 / (sigma * std::sqrt(2.0 * 3.1415926)) * (double)expf(arg));
    }
    float norm = 0;
    for (int i = 0; i < size; i++) norm += t->B[i];
    norm *= 2;
    norm -= t->B[0];
    for (int i = 0; i < size; i++) t->B[i] /= norm;
    for (int i = size; i < kMaxGaussTaps; i++) t->B[i] = 0.f;
    t->size = size;
    return size;
}

// The C copy (src/mask.c:248-340, what src/zoom.c links): same text, but `exp` is the double function.
static int gaussian_taps_c(float sigma, GaussTaps *t)
{
    const float den = 2 * sigma * sigma;
    const int size = (int)(5 * sigma) + 1;
    if (!(sigma > 0) || size > kMaxGaussTaps) return -1;
    for (int i = 0; i < size; i++)
        t->B[i] = (float)(1 / (sigma * std::sqrt(2.0 * 3.1415926)) * std::exp((double)(-i * i / den)));
    float norm = 0;
    for (int i = 0; i < size; i++) norm += t->B[i];
    norm *= 2;
    norm -= t->B[0];
    for (int i = 0; i < size; i++) t->B[i] /= norm;
    for (int i = size; i < kMaxGaussTaps; i++) t->B[i] = 0.f;
    t->size = size;
    return size;
}

// ---- pyramid resampling, src/zoom.c ----
extern "C" int faldoi_zoom_size(int nx, int ny, float factor, int *nxx, int *nyy)
{
    if (!nxx || !nyy) return fail(FALDOI_ERR_INVALID, "NULL argument");
    *nxx = (int)((float)nx * factor + 0.5);          // src/zoom.c:29-30
    *nyy = (int)((float)ny * factor + 0.5);
    return FALDOI_OK;
}

static int zoom_run(faldoi_ctx *ctx, const float *I, float *Iout, int nx, int ny, int nxx, int nyy, float fx, float fy,
                    const GaussTaps *taps)
{
    const size_t n = (size_t)nx * ny, m = (size_t)nxx * nyy;
    DevBuf buf;
    HIP_TRY(hipMalloc((void **)&buf.p, sizeof(float) * (2 * n + m)));
    float *src = buf.p, *tmp = buf.p + n, *dst = buf.p + 2 * n;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(src, I, sizeof(float) * n, hipMemcpyHostToDevice, s));
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (taps) {
        hipLaunchKernelGGL(gaussian_pass_kernel<false>, dim3(blocks), dim3(256), 0, s, src, tmp, nx, ny, *taps);
        hipLaunchKernelGGL(gaussian_pass_kernel<true>, dim3(blocks), dim3(256), 0, s, tmp, src, nx, ny, *taps);
    }
    const int oblocks = (int)((m + 255) / 256 < 2048 ? (m + 255) / 256 : 2048);
    hipLaunchKernelGGL(zoom_resample_kernel, dim3(oblocks), dim3(256), 0, s, src, dst, nx, ny, nxx, nyy, fx, fy);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(Iout, dst, sizeof(float) * m, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return FALDOI_OK;
}

static int faldoi_zoom_out_impl(faldoi_ctx *ctx, const float *I, float *Iout, int nx, int ny, float factor)
{
    if (!ctx || !I || !Iout) return fail(FALDOI_ERR_INVALID, "NULL argument");
    if (!(factor > 0.f) || !(factor < 1.f)) return fail(FALDOI_ERR_INVALID, "zoom factor %g outside (0,1)", (double)factor);
    int rc = prim_check(ctx, nx, ny);
    if (rc) return rc;
    int nxx, nyy;
    faldoi_zoom_size(nx, ny, factor, &nxx, &nyy);
    if (nxx < 1 || nyy < 1) return fail(FALDOI_ERR_INVALID, "zoomed size %dx%d is empty", nxx, nyy);
    const float sigma = (float)(0.6 * std::sqrt(1.0 / (factor * factor) - 1.0));       // src/zoom.c:54 (ZOOM_SIGMA_ZERO)
    GaussTaps taps;
    const int size = gaussian_taps_c(sigma, &taps);
    if (size < 0 || size > nx || size > ny)
        return fail(FALDOI_ERR_INVALID, "image %dx%d too small (or factor %g too small) for the %d-tap pre-smoothing", nx, ny,
                    (double)factor, size);
    return zoom_run(ctx, I, Iout, nx, ny, nxx, nyy, factor, factor, &taps);
}

extern "C" int faldoi_zoom_out(faldoi_ctx *ctx, const float *I, float *Iout, int nx, int ny, float factor)
{
    return abi_guard("zoom_out", [&] {
        OpScope scope(ctx);                       // (kernels of this call count as work on the device: DeviceGate)
        return faldoi_zoom_out_impl(ctx, I, Iout, nx, ny, factor);
    });
}

static int faldoi_zoom_in_impl(faldoi_ctx *ctx, const float *I, float *Iout, int nx, int ny, int nxx, int nyy)
{
    if (!ctx || !I || !Iout) return fail(FALDOI_ERR_INVALID, "NULL argument");
    int rc = prim_check(ctx, nx, ny);
    if (rc) return rc;
    if (nxx < 1 || nyy < 1 || (long)nxx * nyy > (1L << 28)) return fail(FALDOI_ERR_INVALID, "bad zoomed size %dx%d", nxx, nyy);
    const float fx = ((float)nxx / nx), fy = ((float)nyy / ny);                         // src/zoom.c:92-93
    return zoom_run(ctx, I, Iout, nx, ny, nxx, nyy, fx, fy, nullptr);
}

extern "C" int faldoi_zoom_in(faldoi_ctx *ctx, const float *I, float *Iout, int nx, int ny, int nxx, int nyy)
{
    return abi_guard("zoom_in", [&] {
        OpScope scope(ctx);                       // (kernels of this call count as work on the device: DeviceGate)
        return faldoi_zoom_in_impl(ctx, I, Iout, nx, ny, nxx, nyy);
    });
}

// Device-side pipeline on three gray planes already in HBM (A = i0, B = i1, C = i_1), in place.
// tmp: one scratch plane, part: >= 6*(blocks+1) floats.
static int preprocess_device(faldoi_ctx *c, float *A, float *B, float *C, float *tmp, float *part, int nx, int ny,
                             float sigma)
{
    GaussTaps taps;
    const int size = gaussian_taps(sigma, &taps);
    if (size < 0) return fail(FALDOI_ERR_INVALID, "sigma=%g unsupported", (double)sigma);
    if (size > nx || size > ny) return fail(FALDOI_ERR_INVALID, "image %dx%d too small for the %d-tap Gaussian", nx, ny, size);
    const int n = nx * ny;
    const int blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
    hipStream_t s = c->stream;
    // C == nullptr: the two-image form of the local step (image_normalization, src/utils.cpp:703)
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(blocks), dim3(256), 0, s, A, B, C ? C : A, n, part + 6);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, s, part + 6, blocks, part);
    if (C) hipLaunchKernelGGL(normalize3_kernel, dim3(blocks), dim3(256), 0, s, A, B, C, n, part);
    else hipLaunchKernelGGL(normalize2_kernel, dim3(blocks), dim3(256), 0, s, A, B, n, part);
    float *planes[3] = {A, B, C};
    for (int k = 0; k < (C ? 3 : 2); ++k) {           // src/global_faldoi.cpp:2066-2068
        hipLaunchKernelGGL(gaussian_pass_kernel<false>, dim3(blocks), dim3(256), 0, s, planes[k], tmp, nx, ny, taps);
        hipLaunchKernelGGL(gaussian_pass_kernel<true>, dim3(blocks), dim3(256), 0, s, tmp, planes[k], nx, ny, taps);
    }
    HIP_TRY(hipGetLastError());
    return FALDOI_OK;
}

// Host planar image (pd channels) -> gray plane in HBM.  staging: >= 3*n floats of device memory.
static int upload_gray(faldoi_ctx *c, const float *img, int n, int pd, float *staging, float *dst)
{
    if (pd == 1) {
        HIP_TRY(hipMemcpyAsync(dst, img, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(staging, img, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
        const int blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
        hipLaunchKernelGGL(rgb2gray_kernel, dim3(blocks), dim3(256), 0, c->stream, staging, dst, n);
        HIP_TRY(hipGetLastError());
    }
    return FALDOI_OK;
}

static int preprocess_check(faldoi_ctx *ctx, const float *i0, const float *i1, int nx, int ny, int pd)
{
    if (!ctx || !i0 || !i1) return fail(FALDOI_ERR_INVALID, "NULL argument");
    if (pd != 1 && pd != 3) return fail(FALDOI_ERR_INVALID, "pd=%d: only gray (1) or RGB (3) planar images", pd);
    if (nx < 6 || ny < 6 || (long)nx * ny > (1L << 28)) return fail(FALDOI_ERR_INVALID, "image size %dx%d unsupported", nx, ny);
    HIP_TRY(hipSetDevice(ctx->device));
    return FALDOI_OK;
}

static int faldoi_preprocess_global_impl(faldoi_ctx *ctx, const float *i0, const float *i1, const float *im1, int nx,
                                        int ny, int pd, float *i0n, float *i1n)
{
    int rc = preprocess_check(ctx, i0, i1, nx, ny, pd);
    if (rc) return rc;
    if (!i0n || !i1n) return fail(FALDOI_ERR_INVALID, "NULL output");
    const size_t n = (size_t)nx * ny;
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof(float) * (8 * n + 6 * 1026)));
    float *A = d, *B = d + n, *C = d + 2 * n, *tmp = d + 3 * n, *staging = d + 4 * n, *part = d + 8 * n;
    rc = upload_gray(ctx, i0, (int)n, pd, staging, A);
    if (!rc) rc = upload_gray(ctx, i1, (int)n, pd, staging, B);
    if (!rc) rc = upload_gray(ctx, im1 ? im1 : i1, (int)n, pd, staging, C);     // 2-line list file: I-1 := I1
    if (!rc) rc = preprocess_device(ctx, A, B, C, tmp, part, nx, ny, 0.90f);  // PRESMOOTHING_SIGMA
    if (!rc) {
        hipError_t e = hipMemcpyAsync(i0n, A, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(i1n, B, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(FALDOI_ERR_HIP, "copy back failed: %s", hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(d);
    return rc;
}

extern "C" int faldoi_preprocess_global(faldoi_ctx *ctx, const float *i0, const float *i1, const float *im1, int nx,
                                        int ny, int pd, float *i0n, float *i1n)
{
    return abi_guard("the preprocessing", [&] {
        OpScope scope(ctx);                       // (kernels of this call count as work on the device: DeviceGate)
        return faldoi_preprocess_global_impl(ctx, i0, i1, im1, nx, ny, pd, i0n, i1n);
    });
}

static int faldoi_tvl1_batch_upload_images_impl(faldoi_tvl1_batch *b, int pair, const float *i0, const float *i1,
                                               const float *im1, int pd, const float *u, const float *xi)
{
    if (!b || !u) return fail(FALDOI_ERR_INVALID, "NULL argument");
    if (pair < 0 || pair >= b->npairs) return fail(FALDOI_ERR_INVALID, "pair %d out of range", pair);
    faldoi_ctx *c = b->ctx;
    int rc = preprocess_check(c, i0, i1, b->nx, b->ny, pd);
    if (rc) return rc;
    const size_t n = (size_t)b->nx * b->ny, nb = n * sizeof(float);
    const auto t_up = std::chrono::steady_clock::now();                 // (here "H2D" includes the device-side preprocessing)
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof(float) * (5 * n + 6 * 1026)));
    float *C = d, *tmp = d + n, *staging = d + 2 * n, *part = d + 5 * n;
    float *A = plane_ptr(b, pair, P_I0), *B = plane_ptr(b, pair, P_I1);
    rc = upload_gray(c, i0, (int)n, pd, staging, A);
    if (!rc) rc = upload_gray(c, i1, (int)n, pd, staging, B);
    if (!rc) rc = upload_gray(c, im1 ? im1 : i1, (int)n, pd, staging, C);
    if (!rc) rc = preprocess_device(c, A, B, C, tmp, part, b->nx, b->ny, 0.90f);
    hipStream_t s = c->stream;
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpyAsync(plane_ptr(b, pair, P_INIT + 0), u, nb, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(plane_ptr(b, pair, P_INIT + 1), u + n, nb, hipMemcpyHostToDevice, s);
        for (int k = 0; k < 4 && e == hipSuccess; ++k)
            e = xi ? hipMemcpyAsync(plane_ptr(b, pair, P_INIT + 2 + k), xi + k * n, nb, hipMemcpyHostToDevice, s)
                   : hipMemsetAsync(plane_ptr(b, pair, P_INIT + 2 + k), 0, nb, s);
    }
    hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(FALDOI_ERR_HIP, "upload failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    c->last_h2d_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up).count();
    b->uploaded[pair] = 1;
    b->exact_quot = false;
    b->solved = false;
    return FALDOI_OK;
}

extern "C" int faldoi_tvl1_batch_upload_images(faldoi_tvl1_batch *b, int pair, const float *i0, const float *i1,
                                               const float *im1, int pd, const float *u, const float *xi)
{
    return abi_guard("the batch upload", [&] {
        OpScope scope(b ? b->ctx : nullptr);                       // (kernels of this call count as work on the device: DeviceGate)
        return faldoi_tvl1_batch_upload_images_impl(b, pair, i0, i1, im1, pd, u, xi);
    });
}

// ------------------------------------------------------------------------------------------ NLTV-CSAD
// Host-side tables.  They involve pow/exp/hypot, whose last bit is a property of the C library; to
// stay bit-identical with the reference they are evaluated here with the same libm calls the C++
// reference resolves to (pow in double, exp of a float argument = expf, hypot in double, sqrtf)
// and uploaded.  They are computed once per solve; every iteration runs on the device.
extern "C" void faldoi_nltvcsad_default_params(faldoi_nltvcsad_params *p)
{
    if (!p) return;
    p->lambda = 0.85f;      // src/global_faldoi.cpp:2138-2140: main overrides the -p file for -m 6/7
    p->theta = 0.3f;
    p->tau = 0.1f;
    p->warps = FALDOI_DEFAULT_WARPS_GLOBAL;
    p->max_inner = FALDOI_DEFAULT_MAX_INNER;
    p->verbose = 0;
}

static inline float sqf(float f) { return f * f; }      // aux_pow2, src/global_faldoi.cpp:904

// src/global_faldoi.cpp:906-932
static int faldoi_image_to_lab_impl(const float *in, int nx, int ny, float *out)
{
    if (!in || !out || nx < 1 || ny < 1) return fail(FALDOI_ERR_INVALID, "bad argument");
    const int size = nx * ny;
    const float T = 0.008856f;
    const float color_attenuation = 1.5f;
    for (int i = 0; i < size; i++) {
        const float r = in[i] / 255.f, g = in[i + size] / 255.f, b = in[i + 2 * size] / 255.f;
        float X = (float)(0.412453 * r + 0.357580 * g + 0.180423 * b);
        float Y = (float)(0.212671 * r + 0.715160 * g + 0.072169 * b);
        float Z = (float)(0.019334 * r + 0.119193 * g + 0.950227 * b);
        X = (float)(X / 0.950456);
        Z = (float)(Z / 1.088754);
        const float Y3 = (float)std::pow((double)Y, 1. / 3);
        const float fX = (float)(X > T ? std::pow((double)X, 1. / 3) : 7.787 * X + 16 / 116.);
        const float fY = (float)(Y > T ? (double)Y3 : 7.787 * Y + 16 / 116.);
        const float fZ = (float)(Z > T ? std::pow((double)Z, 1. / 3) : 7.787 * Z + 16 / 116.);
        const float L = (float)(Y > T ? 116 * Y3 - 16.0 : 903.3 * Y);
        const float A = 500 * (fX - fY);
        const float B = 200 * (fY - fZ);
        const float inner = (float)(sqf(L / 100) - 0.6);            // aux_pow2 takes a float
        const float correct_lab = expf(-color_attenuation * sqf(inner));
        out[i] = L;
        out[i + size] = A * correct_lab;
        out[i + 2 * size] = B * correct_lab;
    }
    return FALDOI_OK;
}

extern "C" int faldoi_image_to_lab(const float *in, int nx, int ny, float *out)
{
    return abi_guard("image_to_lab", [&] { return faldoi_image_to_lab_impl(in, nx, ny, out); });
}

// Weights of the 24 dual slots and their sum, src/global_faldoi.cpp:935-1054.  wts: [24][n], wt: [n].
static void nltv_weights_rows(const float *a, int pd, int w, int h, int y0, int y1, float *wts, float *wt)
{
    const size_t n = (size_t)w * h;
    for (int j = y0; j < y1; j++)
        for (int i = 0; i < w; i++) {
            const size_t pos = (size_t)j * w + i;
            int it = 0;
            float ne = 0.0f;
            for (int k = -kNlR; k <= kNlR; k++)
                for (int l = -kNlR; l <= kNlR; l++) {
                    if (k == 0 && l == 0) continue;
                    float wp = 0.0f;
                    if (i + l >= 0 && i + l < w && j + k >= 0 && j + k < h) {
                        float difI = 0.0f;
                        for (int m = 0; m < pd; m++) {
                            const float aux = a[n * m + pos] - a[n * m + (size_t)(j + k) * w + (i + l)];
                            difI += aux * aux;
                        }
                        difI = sqrtf(difI);
                        const float wc = expf(-difI / 5.0f);                          // MAX_INTENSITY
                        const float difS = (float)hypot((double)l, (double)k);
                        const float ws = expf(-difS / 2.0f);                          // MAX_BETA
                        wp = sqrtf(wc * ws);
                        ne += wp;
                    }
                    wts[(size_t)it * n + pos] = wp;
                    it++;
                }
            wt[pos] = ne;
        }
}

static int nl_solve(bool data_l1, faldoi_ctx *ctx, const float *I0, const float *I1, const float *a, int pd, int nx,
                    int ny, const faldoi_nltvcsad_params *prm, float *u, double *solve_ms);

static int faldoi_nltvcsad_global_impl(faldoi_ctx *ctx, const float *I0, const float *I1, const float *a, int pd, int nx,
                                      int ny, const faldoi_nltvcsad_params *prm, float *u, double *solve_ms)
{
    return nl_solve(false, ctx, I0, I1, a, pd, nx, ny, prm, u, solve_ms);
}

extern "C" int faldoi_nltvcsad_global(faldoi_ctx *ctx, const float *I0, const float *I1, const float *a, int pd, int nx,
                                      int ny, const faldoi_nltvcsad_params *prm, float *u, double *solve_ms)
{
    return abi_guard("the NLTV-CSAD solver", [&] {
        OpScope scope(ctx);
        return faldoi_nltvcsad_global_impl(ctx, I0, I1, a, pd, nx, ny, prm, u, solve_ms);
    });
}

extern "C" void faldoi_nltvl1_default_params(faldoi_nltvcsad_params *p)
{
    if (!p) return;
    faldoi_nltvcsad_default_params(p);
    p->lambda = 2.0f;      // src/global_faldoi.cpp:2146-2148: main overrides the -p file for -m 2/3
}

static int faldoi_nltvl1_global_impl(faldoi_ctx *ctx, const float *I0, const float *I1, const float *a, int pd, int nx,
                                    int ny, const faldoi_nltvcsad_params *prm, float *u, double *solve_ms)
{
    return nl_solve(true, ctx, I0, I1, a, pd, nx, ny, prm, u, solve_ms);
}

extern "C" int faldoi_nltvl1_global(faldoi_ctx *ctx, const float *I0, const float *I1, const float *a, int pd, int nx,
                                    int ny, const faldoi_nltvcsad_params *prm, float *u, double *solve_ms)
{
    return abi_guard("the NLTV-L1 solver", [&] {
        OpScope scope(ctx);
        return faldoi_nltvl1_global_impl(ctx, I0, I1, a, pd, nx, ny, prm, u, solve_ms);
    });
}

// Rows per tile of the fused NLTV kernel (64 x rows pixels, one workgroup per CU at a time): the grid runs in
// ceil(tiles / CUs) rounds of one tile each.  Fewest rounds first, then the taller tile (fewer crossing edges, and a last round
// that leaves CUs idle runs faster for the others): 1024x436 on 256 CUs -> 16 rows (26.6 ms per 400 iterations; 14 rows, two
// exact rounds, 26.9; 12 rows, three rounds, 32.0).
static int nl_pick_rows(int nx, int ny, int n_cu)
{
    static const int cand[] = { 16, 14, 12, 8 };
    int best = 16;
    long best_rounds = -1;
    for (int ty : cand) {
        const long tiles = (long)((nx + kNfW - 1) / kNfW) * ((ny + ty - 1) / ty);
        const long rounds = (tiles + n_cu - 1) / n_cu;
        if (best_rounds < 0 || rounds < best_rounds) { best = ty; best_rounds = rounds; }
    }
    return best;
}

template <bool L1, int TY> static hipError_t nl_fused_prepare_t()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&nl_fused_kernel<L1, TY>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)sizeof(NfLds<TY>));
}
template <bool L1, int TY> static hipError_t nl_fused_launch_t(hipStream_t s, const NfArgs &a)
{
    const int tiles = a.ntx * a.nty;
    hipLaunchKernelGGL((nl_fused_kernel<L1, TY>), dim3(8 * ((tiles + 7) / 8)), dim3(kNfW * TY), sizeof(NfLds<TY>), s, a);
    return hipGetLastError();
}
#define FALDOI_NL_DISPATCH(fn, ...)                                                       \
    do {                                                                                  \
        if (data_l1) {                                                                    \
            if (ty == 16) return fn<true, 16>(__VA_ARGS__);                               \
            if (ty == 14) return fn<true, 14>(__VA_ARGS__);                               \
            if (ty == 12) return fn<true, 12>(__VA_ARGS__);                               \
            if (ty == 8) return fn<true, 8>(__VA_ARGS__);                                 \
        } else {                                                                          \
            if (ty == 16) return fn<false, 16>(__VA_ARGS__);                              \
            if (ty == 14) return fn<false, 14>(__VA_ARGS__);                              \
            if (ty == 12) return fn<false, 12>(__VA_ARGS__);                              \
            if (ty == 8) return fn<false, 8>(__VA_ARGS__);                                \
        }                                                                                 \
        return hipErrorInvalidValue;                                                      \
    } while (0)
static hipError_t nl_fused_prepare(bool data_l1, int ty) { FALDOI_NL_DISPATCH(nl_fused_prepare_t); }
static hipError_t nl_fused_launch(bool data_l1, int ty, hipStream_t s, const NfArgs &a) { FALDOI_NL_DISPATCH(nl_fused_launch_t, s, a); }
#undef FALDOI_NL_DISPATCH

extern "C" int faldoi_debug_nl_check(faldoi_ctx *ctx, unsigned out[4], int reset, int *compiled_in)
{
    return abi_guard("the NLTV range-check report", [&] {
        if (!ctx || !out) return fail(FALDOI_ERR_INVALID, "NULL argument");
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nf_check), 4 * sizeof(unsigned)));
        if (reset) {
            const unsigned zero[4] = { 0u, 0u, 0u, 0u };
            HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_nf_check), zero, sizeof zero));
        }
        if (compiled_in) *compiled_in = kNfCheck ? 1 : 0;
        return (int)FALDOI_OK;
    });
}

static bool csad_sort_requested()
{
    const char *v = dev_env("FALDOI_CSAD_SORT");
    return v && atoi(v) > 0;
}

static int nl_solve(bool data_l1, faldoi_ctx *ctx, const float *I0, const float *I1, const float *a, int pd, int nx,
                    int ny, const faldoi_nltvcsad_params *prm, float *u, double *solve_ms)
{
    if (!ctx || !I0 || !I1 || !a || !prm || !u) return fail(FALDOI_ERR_INVALID, "NULL argument");
    if (pd < 1 || pd > 4) return fail(FALDOI_ERR_INVALID, "pd=%d unsupported", pd);
    if (nx < 2 || ny < 2 || (long)nx * ny > (1L << 26)) return fail(FALDOI_ERR_INVALID, "image size %dx%d unsupported", nx, ny);
    if (!(prm->lambda > 0) || !(prm->theta > 0) || !(prm->tau > 0) || prm->warps < 1 || prm->max_inner < 1)
        return fail(FALDOI_ERR_INVALID, "bad NLTV-CSAD parameters");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = (size_t)nx * ny, nb = n * sizeof(float);
    const size_t pe = ((n + 63) / 64) * 64;     // (padding the plane stride by odd multiples of 256 B was measured: no effect)

    // host tables (threads over row bands)
    std::vector<float> wts((size_t)kNlSlots * n), wt(n);
    {
        const int bands = ny < 64 ? ny : 64;
        const bool ok = grow_host_parallel(bands, [&](int t) {
            nltv_weights_rows(a, pd, nx, ny, (int)((long)ny * t / bands), (int)((long)ny * (t + 1) / bands), wts.data(),
                              wt.data());
        });
        if (!ok) return fail(FALDOI_ERR_OOM, "out of host memory while building the NLTV weights");
    }
    StreamDrain drain(s);                       // wts / wt are read by asynchronous copies below

    DevBuf buf;
    HIP_TRY(hipMalloc((void **)&buf.p, sizeof(float) * pe * N_COUNT));
    float *base = buf.p;
    HIP_TRY(hipMemsetAsync(base, 0, sizeof(float) * pe * N_COUNT, s));          // duals start at 0 (:1031)
    HIP_TRY(hipMemcpyAsync(base + N_I0 * pe, I0, nb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(base + N_I1 * pe, I1, nb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(base + N_U1 * pe, u, nb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(base + N_U2 * pe, u + n, nb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(base + N_WT * pe, wt.data(), nb, hipMemcpyHostToDevice, s));
    for (int k = 0; k < kNlSlots; ++k)
        HIP_TRY(hipMemcpyAsync(base + (size_t)(N_W + k) * pe, wts.data() + (size_t)k * n, nb, hipMemcpyHostToDevice, s));

    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const float l_t = prm->lambda * prm->theta;                                 // :1659
    // One fused launch per iteration (nltv_fused_kernel.hpp); FALDOI_NL_FUSED=0 selects the two streaming kernels (same bits).
    // (the fused kernel addresses the slab through one buffer resource: below 4 GB, i.e. up to ~7.8 Mpx)
    bool fused = !(getenv("FALDOI_NL_FUSED") && atoi(getenv("FALDOI_NL_FUSED")) <= 0) &&
                       (unsigned long long)pe * N_COUNT * sizeof(float) < (1ull << 32) &&
                       prm->tau < 0x1p35f;                                      // nl_dual_update's argument for |g| below 2^-60
    // FALDOI_CSAD_SORT=1: the CSAD selection by the 48-input sorting network per iteration (the round-3 form; streaming kernels only) --
    // a device-side cross-check of the selection by the warp's rank table (nltv_kernels.hpp), which is the product path.
    const bool ranked = !csad_sort_requested();
    if (!ranked) fused = false;
    NfArgs fa;
    int ty = 0;
    if (fused) {
        ty = nl_pick_rows(nx, ny, ctx->n_cu);
        if (const char *v = dev_env("FALDOI_NL_ROWS")) { const int q = atoi(v); if (q == 8 || q == 12 || q == 14 || q == 16) ty = q; }   // tuning aid
        fa.base = base; fa.pe = pe; fa.nx = nx; fa.ny = ny;
        fa.ntx = (nx + kNfW - 1) / kNfW; fa.nty = (ny + ty - 1) / ty;
        fa.cur = 0; fa.tau = prm->tau; fa.l_t = l_t; fa.theta = prm->theta; fa.rtheta = 1.0 / (double)prm->theta;
        fa.check_selftest = (dev_env("FALDOI_NL_CHECK_SELFTEST") && atoi(dev_env("FALDOI_NL_CHECK_SELFTEST")) > 0) ? 1 : 0;
        if (nl_fused_prepare(data_l1, ty) != hipSuccess) {              // (less LDS than the kernel asks for: the streaming kernels instead)
            (void)hipGetLastError();
            fused = false;
        }
    }
    HIP_TRY(hipEventRecord(ctx->ev0, s));
    hipLaunchKernelGGL(centered_gradient_kernel, grid, block, 0, s, base + N_I1 * pe, base + N_I1X * pe, base + N_I1Y * pe,
                       nx, ny, (size_t)0, (size_t)0, 1);                        // :1690
    for (int w = 0; w < prm->warps; ++w) {
        hipLaunchKernelGGL(nl_warp_kernel, grid, block, 0, s, base, pe, nx, ny, data_l1 ? 1 : 0, fused ? fa.cur : 0);
        if (!data_l1 && ranked)                                                 // the order of this warp's 48 CSAD terms per pixel
            hipLaunchKernelGGL(csad_rank_kernel, grid, block, 0, s, base + N_I0 * pe, base + N_I1W * pe, base + N_T1 * pe, base + N_T2 * pe,
                               reinterpret_cast<unsigned *>(base + N_RK * pe), pe, nx, ny);
        for (int it = 0; it < prm->max_inner; ++it) {                           // :1729 / :1240, no tolerance test
            if (fused) {
                const hipError_t e = nl_fused_launch(data_l1, ty, s, fa);
                if (e == hipSuccess) {
                    fa.cur ^= 1;                                                // duals and ubar: read one buffer, write the other
                    continue;
                }
                // a refused FIRST launch (nothing has been written yet; the per-warp kernel put ubar where both paths read it) sends
                // the whole solve to the two streaming kernels, which return the same bits; anything later is an error
                if (w != 0 || it != 0) HIP_TRY(e);
                (void)hipGetLastError();
                fprintf(stderr, "faldoi: the fused NLTV kernel could not be launched (%s); using the two-kernel path\n", hipGetErrorString(e));
                fused = false;
            }
            hipLaunchKernelGGL(nl_dual_kernel, grid, block, 0, s, base, pe, nx, ny, prm->tau);
            if (data_l1)
                hipLaunchKernelGGL((nl_primal_kernel<true, true>), grid, block, 0, s, base, pe, nx, ny, l_t, prm->theta, prm->tau);
            else if (ranked)
                hipLaunchKernelGGL((nl_primal_kernel<false, true>), grid, block, 0, s, base, pe, nx, ny, l_t, prm->theta, prm->tau);
            else
                hipLaunchKernelGGL((nl_primal_kernel<false, false>), grid, block, 0, s, base, pe, nx, ny, l_t, prm->theta, prm->tau);
        }
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ctx->ev1, s));
    HIP_TRY(hipMemcpyAsync(u, base + N_U1 * pe, nb, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(u + n, base + N_U2 * pe, nb, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (solve_ms) *solve_ms = ms;
    if (prm->verbose) fprintf(stderr, "NLTV (%s data term): %d warps x %d iterations, %.3f ms on the device\n", data_l1 ? "L1" : "CSAD", prm->warps, prm->max_inner, ms);
    return FALDOI_OK;
}

// ------------------------------------------------------------------------------------------ local patch operator
struct faldoi_local_tvl1 {
    faldoi_ctx *ctx = nullptr;
    int nx = 0, ny = 0;
    size_t pe = 0;
    float *planes = nullptr;          // [L_COUNT][pe]
    PatchDesc *d_patches = nullptr;
    float *d_ener = nullptr;
    int *d_iters = nullptr;
    int *d_fixed = nullptr;           // OpticalFlowData::fixed_points (1 = fixed), nx*ny
    int cap = 0;
};

extern "C" void faldoi_local_tvl1_destroy(faldoi_local_tvl1 *h)
{
    if (!h) return;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->planes) (void)hipFree(h->planes);
    if (h->d_patches) (void)hipFree(h->d_patches);
    if (h->d_ener) (void)hipFree(h->d_ener);
    if (h->d_iters) (void)hipFree(h->d_iters);
    if (h->d_fixed) (void)hipFree(h->d_fixed);
    delete h;
}

static int faldoi_local_tvl1_create_impl(faldoi_ctx *ctx, const float *I0, const float *I1, int nx, int ny,
                                        faldoi_local_tvl1 **out)
{
    if (!ctx || !I0 || !I1 || !out) return fail(FALDOI_ERR_INVALID, "NULL argument");
    *out = nullptr;
    int rc = prim_check(ctx, nx, ny);
    if (rc) return rc;
    faldoi_local_tvl1 *h = new (std::nothrow) faldoi_local_tvl1;
    if (!h) return fail(FALDOI_ERR_OOM, "host allocation failed");
    h->ctx = ctx; h->nx = nx; h->ny = ny;
    const size_t n = (size_t)nx * ny;
    h->pe = ((n + 63) / 64) * 64;
    hipStream_t s = ctx->stream;
    hipError_t e = hipMalloc((void **)&h->planes, sizeof(float) * h->pe * L_COUNT);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_fixed, sizeof(int) * n);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_fixed, 0, sizeof(int) * n, s);
    if (e == hipSuccess) e = hipMemsetAsync(h->planes, 0, sizeof(float) * h->pe * L_COUNT, s);   // duals, div, flow start at 0
    if (e == hipSuccess) e = hipMemcpyAsync(h->planes + L_I0 * h->pe, I0, sizeof(float) * n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(h->planes + L_I1 * h->pe, I1, sizeof(float) * n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        // prepare_stuff (src/energy_model.cpp:660-686): centred gradient of the target frame
        hipLaunchKernelGGL(centered_gradient_kernel, dim3(blocks), dim3(256), 0, s, h->planes + L_I1 * h->pe,
                           h->planes + L_I1X * h->pe, h->planes + L_I1Y * h->pe, nx, ny, (size_t)0, (size_t)0, 1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        faldoi_local_tvl1_destroy(h);
        return fail(e == hipErrorOutOfMemory ? FALDOI_ERR_OOM : FALDOI_ERR_HIP, "local context setup failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return FALDOI_OK;
}

extern "C" int faldoi_local_tvl1_create(faldoi_ctx *ctx, const float *I0, const float *I1, int nx, int ny,
                                        faldoi_local_tvl1 **out)
{
    return abi_guard("the local operator", [&] {
        OpScope scope(ctx);                       // (kernels of this call count as work on the device: DeviceGate)
        return faldoi_local_tvl1_create_impl(ctx, I0, I1, nx, ny, out);
    });
}

extern "C" int faldoi_local_tvl1_set_flow(faldoi_local_tvl1 *h, const float *u)
{
    if (!h || !u) return fail(FALDOI_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->ctx->device));
    const size_t n = (size_t)h->nx * h->ny;
    HIP_TRY(hipMemcpyAsync(h->planes + L_U1 * h->pe, u, sizeof(float) * n, hipMemcpyHostToDevice, h->ctx->stream));
    HIP_TRY(hipMemcpyAsync(h->planes + L_U2 * h->pe, u + n, sizeof(float) * n, hipMemcpyHostToDevice, h->ctx->stream));
    HIP_TRY(hipStreamSynchronize(h->ctx->stream));
    return FALDOI_OK;
}

extern "C" int faldoi_local_tvl1_get_flow(faldoi_local_tvl1 *h, float *u)
{
    if (!h || !u) return fail(FALDOI_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->ctx->device));
    const size_t n = (size_t)h->nx * h->ny;
    HIP_TRY(hipMemcpyAsync(u, h->planes + L_U1 * h->pe, sizeof(float) * n, hipMemcpyDeviceToHost, h->ctx->stream));
    HIP_TRY(hipMemcpyAsync(u + n, h->planes + L_U2 * h->pe, sizeof(float) * n, hipMemcpyDeviceToHost, h->ctx->stream));
    HIP_TRY(hipStreamSynchronize(h->ctx->stream));
    return FALDOI_OK;
}

static int local_patches_to_device(faldoi_local_tvl1 *h, int npatches, const int *patches)
{
    if (npatches < 1 || npatches > (1 << 20)) return fail(FALDOI_ERR_INVALID, "npatches=%d outside 1..2^20", npatches);
    for (int i = 0; i < npatches; ++i) {
        const int ii = patches[4 * i], ij = patches[4 * i + 1], ei = patches[4 * i + 2], ej = patches[4 * i + 3];
        if (ii < 0 || ij < 0 || ei > h->nx || ej > h->ny || ei - ii < 2 || ej - ij < 2 || ei - ii > kPatchMax ||
            ej - ij > kPatchMax)
            return fail(FALDOI_ERR_INVALID, "patch %d [%d,%d)x[%d,%d) outside the image or not 2..%d wide/high", i, ii, ei, ij,
                        ej, kPatchMax);
    }
    HIP_TRY(hipSetDevice(h->ctx->device));
    if (h->cap < npatches) {
        if (h->d_patches) (void)hipFree(h->d_patches);
        if (h->d_ener) (void)hipFree(h->d_ener);
        if (h->d_iters) (void)hipFree(h->d_iters);
        h->d_patches = nullptr; h->d_ener = nullptr; h->d_iters = nullptr; h->cap = 0;
        HIP_TRY(hipMalloc((void **)&h->d_patches, sizeof(PatchDesc) * npatches));
        HIP_TRY(hipMalloc((void **)&h->d_ener, sizeof(float) * npatches));
        HIP_TRY(hipMalloc((void **)&h->d_iters, sizeof(int) * npatches));
        h->cap = npatches;
    }
    HIP_TRY(hipMemcpyAsync(h->d_patches, patches, sizeof(PatchDesc) * npatches, hipMemcpyHostToDevice, h->ctx->stream));
    return FALDOI_OK;
}

extern "C" int faldoi_local_tvl1_set_fixed(faldoi_local_tvl1 *h, const int *fixed_points)
{
    if (!h || !fixed_points) return fail(FALDOI_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->ctx->device));
    HIP_TRY(hipMemcpyAsync(h->d_fixed, fixed_points, sizeof(int) * (size_t)h->nx * h->ny, hipMemcpyHostToDevice, h->ctx->stream));
    HIP_TRY(hipStreamSynchronize(h->ctx->stream));
    return FALDOI_OK;
}

static int faldoi_local_tvl1_interpolate_impl(faldoi_local_tvl1 *h, int npatches, const int *patches)
{
    if (!h || !patches) return fail(FALDOI_ERR_INVALID, "NULL argument");
    int rc = local_patches_to_device(h, npatches, patches);
    if (rc) return rc;
    hipStream_t s = h->ctx->stream;
    hipLaunchKernelGGL(interpolate_poisson_kernel, dim3((2 * npatches + 63) / 64), dim3(64), 0, s, h->planes, h->pe, h->nx,
                       h->d_fixed, h->d_patches, npatches);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return FALDOI_OK;
}

extern "C" int faldoi_local_tvl1_interpolate(faldoi_local_tvl1 *h, int npatches, const int *patches)
{
    return abi_guard("the window filling", [&] {
        OpScope scope(h ? h->ctx : nullptr);
        return faldoi_local_tvl1_interpolate_impl(h, npatches, patches);
    });
}

static int faldoi_local_tvl1_estimate_impl(faldoi_local_tvl1 *h, int npatches, const int *patches, int max_iter_patch,
                                          float *ener_out, int *iters_out)
{
    if (!h || !patches || !ener_out) return fail(FALDOI_ERR_INVALID, "NULL argument");
    if (max_iter_patch < 0 || max_iter_patch > 100000) return fail(FALDOI_ERR_INVALID, "bad max_iter_patch");
    int rc = local_patches_to_device(h, npatches, patches);
    if (rc) return rc;
    hipStream_t s = h->ctx->stream;
    hipLaunchKernelGGL(local_tvl1_patch_kernel, dim3(npatches), dim3(kPatchThreads), 0, s, h->planes, h->pe, h->nx, h->ny,
                       h->d_patches, max_iter_patch, h->d_ener, h->d_iters);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ener_out, h->d_ener, sizeof(float) * npatches, hipMemcpyDeviceToHost, s));
    if (iters_out) HIP_TRY(hipMemcpyAsync(iters_out, h->d_iters, sizeof(int) * npatches, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return FALDOI_OK;
}

extern "C" int faldoi_local_tvl1_estimate(faldoi_local_tvl1 *h, int npatches, const int *patches, int max_iter_patch,
                                          float *ener_out, int *iters_out)
{
    return abi_guard("the local operator", [&] {
        OpScope scope(h ? h->ctx : nullptr);
        return faldoi_local_tvl1_estimate_impl(h, npatches, patches, max_iter_patch, ener_out, iters_out);
    });
}

// ------------------------------------------------------------------------------------------ self-test
static int faldoi_selftest_division_impl(faldoi_ctx *ctx, unsigned seed, unsigned long long count,
                                        unsigned long long *mismatches)
{
    if (!ctx || !mismatches) return fail(FALDOI_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float *dump = nullptr;
    const int dump_cap = dev_env("FALDOI_SELFTEST_DUMP") ? 64 : 0;
    if (dump_cap) HIP_TRY(hipMalloc((void **)&dump, sizeof(float) * 4 * dump_cap));
    HIP_TRY(hipMalloc((void **)&d, sizeof h));
    hipError_t e = hipMemsetAsync(d, 0, sizeof h, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(selftest_division_kernel, dim3(2048), dim3(256), 0, ctx->stream, seed, count, d, dump,
                           dump_cap);
        e = hipGetLastError();
    }
    if (e == hipSuccess && dump_cap) {
        std::vector<float> hd(4 * dump_cap, 0.f);
        e = hipMemcpy(hd.data(), dump, sizeof(float) * hd.size(), hipMemcpyDeviceToHost);
        for (int i = 0; i < dump_cap && e == hipSuccess; ++i) {
            unsigned b[4];
            memcpy(b, &hd[4 * i], 16);
            if (b[0] | b[1]) fprintf(stderr, "selftest offender a=%08x d=%08x ieee=%08x fp64=%08x\n", b[0], b[1], b[2], b[3]);
        }
        (void)hipFree(dump);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(FALDOI_ERR_HIP, "selftest failed: %s", hipGetErrorString(e));
    *mismatches = h[0];
    // breakdown for the text of faldoi_last_error: dual-site, subnormal-result, huge-divisor cases; the one-product form, the
    // divisors it refuses (div_once_ok) and how many of their quotients it would have got wrong
    fail(FALDOI_OK, "division selftest: %llu mismatches (dual site %llu, subnormal result %llu, divisor>1e30 %llu, one-product form "
         "%llu); one-product form refused %llu quotients, %llu of them would differ",
         h[0], h[1], h[2], h[3], h[5], h[6], h[7]);
    return FALDOI_OK;
}

extern "C" int faldoi_selftest_division(faldoi_ctx *ctx, unsigned seed, unsigned long long count,
                                        unsigned long long *mismatches)
{
    return abi_guard("the division self-test", [&] {
        OpScope scope(ctx);                       // (kernels of this call count as work on the device: DeviceGate)
        return faldoi_selftest_division_impl(ctx, seed, count, mismatches);
    });
}

// ------------------------------------------------------------------------------------------ primitives

static int prim_check(faldoi_ctx *ctx, int nx, int ny)
{
    if (!ctx) return fail(FALDOI_ERR_INVALID, "ctx is NULL");
    if (nx < 2 || ny < 2 || (long)nx * ny > (1L << 28)) return fail(FALDOI_ERR_INVALID, "image size %dx%d unsupported", nx, ny);
    HIP_TRY(hipSetDevice(ctx->device));
    return FALDOI_OK;
}

// Runs `launch(d_in[0..nin), d_out[0..nout))` on device copies of the host planes.
template <typename F>
static int run_planes(faldoi_ctx *ctx, int nx, int ny, const float *const *in, int nin, float *const *out, int nout, F launch)
{
    int rc = prim_check(ctx, nx, ny);
    if (rc) return rc;
    OpScope scope(ctx);                                  // (the primitives' kernels count as work on the device too: DeviceGate)
    for (int i = 0; i < nin; ++i) if (!in[i]) return fail(FALDOI_ERR_INVALID, "NULL input plane");
    for (int i = 0; i < nout; ++i) if (!out[i]) return fail(FALDOI_ERR_INVALID, "NULL output plane");
    const size_t n = (size_t)nx * ny, nb = n * sizeof(float);
    DevBuf buf;
    HIP_TRY(hipMalloc((void **)&buf.p, nb * (nin + nout)));
    float *d_in[4], *d_out[4];
    for (int i = 0; i < nin; ++i) {
        d_in[i] = buf.p + i * n;
        HIP_TRY(hipMemcpyAsync(d_in[i], in[i], nb, hipMemcpyHostToDevice, ctx->stream));
    }
    for (int i = 0; i < nout; ++i) d_out[i] = buf.p + (nin + i) * n;
    const int blocks = (int)((n + 255) / 256);
    launch(dim3(blocks < 4096 ? blocks : 4096), dim3(256), ctx->stream, d_in, d_out);
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < nout; ++i) HIP_TRY(hipMemcpyAsync(out[i], d_out[i], nb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FALDOI_OK;
}

extern "C" int faldoi_centered_gradient(faldoi_ctx *ctx, const float *in, float *dx, float *dy, int nx, int ny)
{
    const float *ins[1] = {in};
    float *outs[2] = {dx, dy};
    return run_planes(ctx, nx, ny, ins, 1, outs, 2, [&](dim3 g, dim3 b, hipStream_t s, float **i, float **o) {
        hipLaunchKernelGGL(centered_gradient_kernel, g, b, 0, s, i[0], o[0], o[1], nx, ny, (size_t)0, (size_t)0, 1);
    });
}

extern "C" int faldoi_forward_gradient(faldoi_ctx *ctx, const float *f, float *fx, float *fy, int nx, int ny)
{
    const float *ins[1] = {f};
    float *outs[2] = {fx, fy};
    return run_planes(ctx, nx, ny, ins, 1, outs, 2, [&](dim3 g, dim3 b, hipStream_t s, float **i, float **o) {
        hipLaunchKernelGGL(forward_gradient_kernel, g, b, 0, s, i[0], o[0], o[1], nx, ny);
    });
}

extern "C" int faldoi_divergence(faldoi_ctx *ctx, const float *v1, const float *v2, float *dv, int nx, int ny)
{
    const float *ins[2] = {v1, v2};
    float *outs[1] = {dv};
    return run_planes(ctx, nx, ny, ins, 2, outs, 1, [&](dim3 g, dim3 b, hipStream_t s, float **i, float **o) {
        hipLaunchKernelGGL(divergence_kernel, g, b, 0, s, i[0], i[1], o[0], nx, ny);
    });
}

extern "C" int faldoi_bicubic_warp(faldoi_ctx *ctx, const float *in, const float *u, const float *v, float *out, int nx,
                                   int ny, int border_out)
{
    const float *ins[3] = {in, u, v};
    float *outs[1] = {out};
    return run_planes(ctx, nx, ny, ins, 3, outs, 1, [&](dim3 g, dim3 b, hipStream_t s, float **i, float **o) {
        hipLaunchKernelGGL(bicubic_warp_kernel, g, b, 0, s, i[0], i[1], i[2], o[0], nx, ny, border_out);
    });
}

#include "grow_host.inc"
