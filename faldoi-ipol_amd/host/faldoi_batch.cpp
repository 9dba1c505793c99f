// faldoi_batch -- many independent image pairs sharded over the GPUs of one node (BASELINE configs[3]), in C++:
// one process per GPU, RCCL directly.  The reference's batch surface is its driver script chain
// (scripts_python/faldoi_sift.py:274-318: one local_faldoi + one global_faldoi call per pair, pairs in sequence);
// pairs never exchange data, so the only communication is the distribution of the inputs
// and the collection of the flows:
//
//   rank 0   decodes every job's frames and flows, BROADCASTS the job table (ncclBroadcast), SENDS every other
//            rank its shard (uint8 frames + fp32 flows; grouped ncclSend / ncclRecv -- RCCL has no scatter)
//   rank r   solves pairs r, r + N, r + 2N, ... (round-robin, as faldoi-ipol_amd/shard.py) in ONE batched launch
//            sequence on its own GPU through the C ABI of libfaldoi_hip.so: for pipeline jobs the local step of the
//            whole shard (faldoi_match_growing_batch), then the global step (faldoi_tvl1_batch_*)
//   rank 0   RECEIVES the flows and writes the .flo files.
//
// usage:  faldoi_batch jobs.txt [-gpus N] [-w 5] [-max_inner 400] [-tol_stop 1]
//                      [-split_img 0|1|2] [-h_parts 3] [-v_parts 2] [-first_h_parts 0] [-first_v_parts 0]
//   jobs.txt (as faldoi-ipol_amd/batch.py): one job per line, all of one form and one image size:
//     `I0.png I1.png in_flow.flo|zero out.flo`   what one global_faldoi call takes (2-line list-file semantics)
//     `I0.png I1.png s0.flo s1.flo out.flo`      local step from the sparse seed flows, then the global step
//   -gpus N: the program starts its N ranks itself (fork before any GPU call; the RCCL unique id travels through
//   pipes).  Under an external launcher (RANK, LOCAL_RANK, WORLD_SIZE set, e.g. `torchrun --no-python`) the id
//   travels through a file that is unique per run: /tmp/faldoi_batch_<launcher pid>_<MASTER_PORT>_<restart count>.id
//   (or FALDOI_BATCH_ID_FILE); rank 0 removes whatever is there before it publishes and unlinks the file however it
//   leaves, the others only accept a file whose writer is alive (and, for the default name, their own sibling).
//   Nothing waits for ever: every phase that can block on another rank (rendezvous, communicator set-up, each
//   collective) runs under a watchdog -- FALDOI_BATCH_TIMEOUT_S, default 1800 s -- that names the phase and ends the
//   process with exit code 4; a self-started parent then ends its remaining ranks.
//
// Built twice from this source: against RCCL + HIP + libfaldoi_hip (the product), and with -DFALDOI_BATCH_STUB
// (CPU test of the rank logic: pipes instead of RCCL, `flow + 1` instead of the solver, no GPU library at all).
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <atomic>
#include <condition_variable>
#include <csignal>
#include <mutex>

#include <sys/prctl.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "image_io.hpp"
#include "options.hpp"

#ifndef FALDOI_BATCH_STUB
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "faldoi_hip.h"
#endif

using namespace faldoi_host;

namespace {

struct Job {
    std::vector<std::string> tok;      // 4 tokens: global-only job; 5: pipeline job
};

// batch.py parse_jobs
std::vector<Job> parse_jobs(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::vector<Job> jobs;
    int ln = 0;
    for (std::string line; std::getline(in, line);) {
        ++ln;
        std::istringstream ss(line);
        Job j;
        for (std::string t; ss >> t;) j.tok.push_back(t);
        if (j.tok.empty() || j.tok[0][0] == '#') continue;
        if (j.tok.size() != 4 && j.tok.size() != 5)
            throw std::runtime_error(path + ":" + std::to_string(ln) + ": expected `I0 I1 in_flow|zero out.flo` or `I0 I1 s0.flo s1.flo out.flo`");
        if (!jobs.empty() && jobs[0].tok.size() != j.tok.size())
            throw std::runtime_error(path + ": global-only and pipeline jobs cannot be mixed");
        jobs.push_back(j);
    }
    if (jobs.empty()) throw std::runtime_error(path + ": no jobs");
    return jobs;
}

// shard.assign_pairs: pair i -> rank i % world
std::vector<int> shard_of(int njobs, int world, int rank)
{
    std::vector<int> idx;
    for (int i = rank; i < njobs; i += world) idx.push_back(i);
    return idx;
}

struct Header {                        // what rank 0 broadcasts before any payload: the job table, or why there is none
    int32_t status;                    // 0 ok, 1 = rank 0 could not read the inputs (every rank leaves with exit code 3)
    int32_t njobs, w, h, planes;       // planes: 2 = initial flow (global-only), 4 = forward + backward seed flows (pipeline)
};

// ------------------------------------------------------------------------------------------ bounded waits
// A rank that waits for another one (rendezvous file, ncclCommInitRank, a collective) cannot know whether the peer is slow or
// gone: RCCL's calls have no timeout of their own.  Every such phase is bracketed by a Watch; when a phase outlives the limit
// the watchdog thread says which one and ends the PROCESS (a thread parked inside RCCL cannot be unwound) with exit code 4.
class Watchdog {
public:
    static Watchdog &get() { static Watchdog *w = new Watchdog; return *w; }   // never destroyed: its thread waits on it until the process ends
    static double limit_s()
    {
        const char *v = getenv("FALDOI_BATCH_TIMEOUT_S");
        const double t = v ? atof(v) : 0.0;
        return t > 0.0 ? t : 1800.0;
    }
    void arm(const char *phase, int rank)
    {
        std::lock_guard<std::mutex> g(m_);
        phase_ = phase; rank_ = rank; armed_ = true;
        deadline_ = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(limit_s()));
        if (!started_) { started_ = true; std::thread([this] { run(); }).detach(); }
        cv_.notify_all();
    }
    void disarm() { std::lock_guard<std::mutex> g(m_); armed_ = false; cv_.notify_all(); }
    void set_children(const std::vector<pid_t> &kids) { std::lock_guard<std::mutex> g(m_); kids_ = kids; }
private:
    void run()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (!armed_) { cv_.wait(lk); continue; }
            if (cv_.wait_until(lk, deadline_) == std::cv_status::timeout && armed_ && std::chrono::steady_clock::now() >= deadline_) {
                fprintf(stderr, "faldoi_batch: rank %d: no progress in \"%s\" for %.0f s (FALDOI_BATCH_TIMEOUT_S): a peer is gone or hung; giving up\n",
                        rank_, phase_, limit_s());
                fflush(stderr);
                for (pid_t k : kids_) (void)kill(k, SIGKILL);          // a self-started rank 0 takes its ranks with it
                _exit(4);
            }
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::chrono::steady_clock::time_point deadline_;
    const char *phase_ = "";
    int rank_ = 0;
    bool armed_ = false, started_ = false;
    std::vector<pid_t> kids_;
};
struct Watch {
    Watch(const char *phase, int rank) { Watchdog::get().arm(phase, rank); }
    ~Watch() { Watchdog::get().disarm(); }
};

// ------------------------------------------------------------------------------------------ rendezvous through a file (launcher mode)
// The payload (RCCL's unique id) travels from rank 0 to the others through a file.  What can go wrong with a fixed name: an earlier
// run on the same port that died leaves its file behind, and a rank of the NEXT run reads that stale id before rank 0 has replaced it
// -- ncclCommInitRank then never completes.  So: the default name carries the launcher's pid (every rank's parent) and the restart
// count; rank 0 removes what is there before anything else, writes {magic, its pid, payload} to a temporary name and renames it, and
// unlinks it however it leaves (RendezvousFile's destructor); a reader accepts a file only if its writer is alive and -- for the
// default name -- a child of the reader's own parent.
struct RendezvousRecord {
    uint64_t magic;
    int64_t writer_pid;
    unsigned char payload[256];
};
constexpr uint64_t kRendezvousMagic = 0x46414c444f494231ull;           // "FALDOIB1"

[[maybe_unused]] std::string rendezvous_default_name()
{
    const char *port = getenv("MASTER_PORT"), *restart = getenv("TORCHELASTIC_RESTART_COUNT");
    return std::string("/tmp/faldoi_batch_") + std::to_string((long)getppid()) + "_" + (port ? port : "0") + "_" + (restart ? restart : "0") + ".id";
}
long parent_of(long pid)                                               // field 4 of /proc/<pid>/stat; -1: no such process
{
    std::ifstream f("/proc/" + std::to_string(pid) + "/stat");
    std::string line;
    if (!f || !std::getline(f, line)) return -1;
    const size_t close = line.rfind(')');                              // the command name may hold spaces and parentheses
    if (close == std::string::npos) return -1;
    std::istringstream ss(line.substr(close + 1));
    std::string state;
    long ppid = -1;
    ss >> state >> ppid;
    return ppid;
}
struct RendezvousFile {
    std::string path;
    bool owner = false;
    ~RendezvousFile() { if (owner) (void)unlink(path.c_str()); }
    void publish(const void *payload, size_t bytes)
    {
        if (bytes > sizeof(RendezvousRecord::payload)) throw std::runtime_error("rendezvous payload too large");
        (void)unlink(path.c_str());                                     // a leftover of an earlier run must not be read as ours
        owner = true;
        RendezvousRecord rec{};
        rec.magic = kRendezvousMagic;
        rec.writer_pid = (int64_t)getpid();
        memcpy(rec.payload, payload, bytes);
        const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&rec, sizeof rec, 1, f) != 1) { if (f) fclose(f); (void)unlink(tmp.c_str()); throw std::runtime_error("cannot write " + tmp); }
        fclose(f);
        if (rename(tmp.c_str(), path.c_str())) { (void)unlink(tmp.c_str()); throw std::runtime_error("cannot publish " + path); }
    }
    // Polls until a record of a live writer is there (same_parent: the writer must be a child of this process's parent); bounded by
    // the watchdog's limit.
    void fetch(void *payload, size_t bytes, bool same_parent) const
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            RendezvousRecord rec{};
            FILE *f = fopen(path.c_str(), "rb");
            const bool got = f && fread(&rec, sizeof rec, 1, f) == 1;
            if (f) fclose(f);
            if (got && rec.magic == kRendezvousMagic) {
                const long pp = parent_of((long)rec.writer_pid);
                if (pp >= 0 && (!same_parent || pp == (long)getppid())) { memcpy(payload, rec.payload, bytes); return; }
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > Watchdog::limit_s())
                throw std::runtime_error("no rendezvous record of a live rank 0 in " + path);
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
    }
};

// ------------------------------------------------------------------------------------------ communicators
struct Comm {
    int rank = 0, world = 1;
    virtual ~Comm() {}
    virtual void bcast(void *host, size_t bytes) = 0;                                                   // from rank 0
    // rank 0 holds `world` chunks of `bytes` (chunk r for rank r) and every rank receives its own
    virtual void scatter(const void *all_on_root, void *mine, size_t bytes) = 0;
    virtual void gather(const void *mine, void *all_on_root, size_t bytes) = 0;
    virtual void barrier() = 0;
};

#ifdef FALDOI_BATCH_STUB
// CPU test build: rank 0 is connected to every other rank by a socket pair created before the fork.
struct PipeComm : Comm {
    std::vector<int> fd;               // rank 0: fd[r] to rank r; rank r: fd[0] to rank 0
    static void put(int f, const void *p, size_t n)
    {
        const char *c = (const char *)p;
        while (n) { const ssize_t k = write(f, c, n); if (k <= 0) { if (errno == EINTR) continue; throw std::runtime_error("pipe write failed"); } c += k; n -= (size_t)k; }
    }
    static void get(int f, void *p, size_t n)
    {
        char *c = (char *)p;
        while (n) { const ssize_t k = read(f, c, n); if (k <= 0) { if (k < 0 && errno == EINTR) continue; throw std::runtime_error("pipe read failed"); } c += k; n -= (size_t)k; }
    }
    void bcast(void *host, size_t bytes) override
    {
        if (rank == 0) for (int r = 1; r < world; ++r) put(fd[r], host, bytes);
        else get(fd[0], host, bytes);
    }
    void scatter(const void *all, void *mine, size_t bytes) override
    {
        if (rank == 0) {
            memcpy(mine, all, bytes);
            for (int r = 1; r < world; ++r) put(fd[r], (const char *)all + (size_t)r * bytes, bytes);
        } else get(fd[0], mine, bytes);
    }
    void gather(const void *mine, void *all, size_t bytes) override
    {
        if (rank == 0) {
            memcpy(all, mine, bytes);
            for (int r = 1; r < world; ++r) get(fd[r], (char *)all + (size_t)r * bytes, bytes);
        } else put(fd[0], mine, bytes);
    }
    void barrier() override { char c = 0; gather(&c, std::vector<char>(world).data(), 1); bcast(&c, 1); }
};
#else
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) throw std::runtime_error(std::string(#x) + ": " + ncclGetErrorString(r_)); } while (0)

// RCCL moves device memory: every call stages through a device buffer of this process's GPU.
struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    void *dev = nullptr;
    size_t cap = 0;
    void init(int device, const ncclUniqueId &id)
    {
        HIP_OK(hipSetDevice(device));
        HIP_OK(hipStreamCreate(&stream));
        NCCL_OK(ncclCommInitRank(&comm, world, id, rank));
    }
    ~RcclComm() override
    {
        if (comm) (void)ncclCommDestroy(comm);
        if (dev) (void)hipFree(dev);
        if (stream) (void)hipStreamDestroy(stream);
    }
    void *staging(size_t bytes)
    {
        if (cap < bytes) {
            if (dev) HIP_OK(hipFree(dev));
            dev = nullptr;
            HIP_OK(hipMalloc(&dev, bytes));
            cap = bytes;
        }
        return dev;
    }
    void bcast(void *host, size_t bytes) override
    {
        void *d = staging(bytes);
        if (rank == 0) HIP_OK(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, stream));
        NCCL_OK(ncclBroadcast(d, d, bytes, ncclChar, 0, comm, stream));
        if (rank != 0) HIP_OK(hipMemcpyAsync(host, d, bytes, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
    }
    void scatter(const void *all, void *mine, size_t bytes) override
    {
        char *d = (char *)staging(rank == 0 ? bytes * (size_t)world : bytes);
        if (rank == 0) HIP_OK(hipMemcpyAsync(d, all, bytes * (size_t)world, hipMemcpyHostToDevice, stream));
        NCCL_OK(ncclGroupStart());
        if (rank == 0) for (int r = 1; r < world; ++r) NCCL_OK(ncclSend(d + (size_t)r * bytes, bytes, ncclChar, r, comm, stream));
        else NCCL_OK(ncclRecv(d, bytes, ncclChar, 0, comm, stream));
        NCCL_OK(ncclGroupEnd());
        HIP_OK(hipMemcpyAsync(mine, d, bytes, hipMemcpyDeviceToHost, stream));     // rank 0: chunk 0 is its own
        HIP_OK(hipStreamSynchronize(stream));
    }
    void gather(const void *mine, void *all, size_t bytes) override
    {
        char *d = (char *)staging(rank == 0 ? bytes * (size_t)world : bytes);
        HIP_OK(hipMemcpyAsync(d, mine, bytes, hipMemcpyHostToDevice, stream));       // rank 0: into chunk 0
        NCCL_OK(ncclGroupStart());
        if (rank == 0) for (int r = 1; r < world; ++r) NCCL_OK(ncclRecv(d + (size_t)r * bytes, bytes, ncclChar, r, comm, stream));
        else NCCL_OK(ncclSend(d, bytes, ncclChar, 0, comm, stream));
        NCCL_OK(ncclGroupEnd());
        if (rank == 0) HIP_OK(hipMemcpyAsync(all, d, bytes * (size_t)world, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
    }
    void barrier() override
    {
        int *d = (int *)staging(sizeof(int));
        NCCL_OK(ncclAllReduce(d, d, 1, ncclInt, ncclSum, comm, stream));
        HIP_OK(hipStreamSynchronize(stream));
    }
};
#endif

// ------------------------------------------------------------------------------------------ the work of one rank
struct Options {
    int warps = 5, max_inner = 400, tol_stop = 1;
    int split_img = 0, h_parts = 3, v_parts = 2, first_h_parts = 0, first_v_parts = 0;
};

// frames: n x 2 x 3 x h x w uint8 (planar R, G, B of I0 then I1); flows: n x planes x h x w fp32 -> out: n x 2 x h x w
void solve_shard(int device, const Options &o, int n, int w, int h, int planes, const uint8_t *frames, const float *flows, float *out)
{
    const size_t px = (size_t)w * h;
    if (n == 0) return;
#ifdef FALDOI_BATCH_STUB
    (void)device; (void)o; (void)frames;
    for (int i = 0; i < n; ++i)
        for (size_t q = 0; q < 2 * px; ++q) out[(size_t)i * 2 * px + q] = flows[(size_t)i * planes * px + q] + 1.0f;
#else
    faldoi_ctx *ctx = nullptr;
    if (faldoi_ctx_create(device, &ctx)) throw std::runtime_error(std::string("no usable GPU: ") + faldoi_last_error());
    auto fail = [&](const char *what) { const std::string m = std::string(what) + ": " + faldoi_last_error(); faldoi_ctx_destroy(ctx); return std::runtime_error(m); };
    std::vector<std::vector<float>> i0(n), i1(n);                       // iio's split layout: planar floats, file sample values
    for (int i = 0; i < n; ++i) {
        i0[i].resize(3 * px);
        i1[i].resize(3 * px);
        const uint8_t *f = frames + (size_t)i * 6 * px;
        for (size_t q = 0; q < 3 * px; ++q) { i0[i][q] = (float)f[q]; i1[i][q] = (float)f[3 * px + q]; }
    }
    std::vector<float> init((size_t)n * 2 * px);
    if (planes == 4) {                                                  // pipeline jobs: the local step of the whole shard at once
        faldoi_local_params lp;
        faldoi_local_default_params(&lp);
        lp.split_img = o.split_img; lp.h_parts = o.h_parts; lp.v_parts = o.v_parts;
        lp.first_h_parts = o.first_h_parts; lp.first_v_parts = o.first_v_parts;
        std::vector<const float *> go(n), ba(n), p0(n), p1(n);
        std::vector<float *> outp(n), enep(n);
        std::vector<float> ene((size_t)n * px);
        for (int i = 0; i < n; ++i) {
            go[i] = flows + (size_t)i * 4 * px; ba[i] = go[i] + 2 * px;
            p0[i] = i0[i].data(); p1[i] = i1[i].data();
            outp[i] = init.data() + (size_t)i * 2 * px; enep[i] = ene.data() + (size_t)i * px;
        }
        faldoi_local_stats st;
        if (faldoi_match_growing_batch(ctx, n, go.data(), ba.data(), p0.data(), p1.data(), nullptr, nullptr, w, h, 3, &lp, enep.data(),
                                       outp.data(), &st))
            throw fail("local step");
    } else {
        memcpy(init.data(), flows, sizeof(float) * init.size());
    }
    faldoi_tvl1_batch *b = nullptr;
    if (faldoi_tvl1_batch_create(ctx, n, w, h, &b)) throw fail("batch allocation");
    for (int i = 0; i < n; ++i)
        if (faldoi_tvl1_batch_upload_images(b, i, i0[i].data(), i1[i].data(), nullptr, 3, init.data() + (size_t)i * 2 * px, nullptr)) {
            faldoi_tvl1_batch_destroy(b);
            throw fail("upload");
        }
    faldoi_tvl1_params prm;
    faldoi_tvl1_default_params(&prm);
    prm.warps = o.warps; prm.max_inner = o.max_inner; prm.tol_stop = o.tol_stop;
    int rc = faldoi_tvl1_batch_solve(b, &prm, nullptr);
    for (int i = 0; i < n && !rc; ++i) rc = faldoi_tvl1_batch_download(b, i, out + (size_t)i * 2 * px, nullptr);
    faldoi_tvl1_batch_destroy(b);
    if (rc) throw fail("global step");
    faldoi_ctx_destroy(ctx);
#endif
}

// Rank 0: decode the inputs.  frames: njobs x 2 x 3 x h x w uint8; flows: njobs x planes x h x w
void load_inputs(const std::vector<Job> &jobs, int &w, int &h, int &planes, std::vector<uint8_t> &frames, std::vector<float> &flows)
{
    const int n = (int)jobs.size();
    planes = jobs[0].tok.size() == 5 ? 4 : 2;
    w = h = 0;
    for (int j = 0; j < n; ++j) {
        for (int k = 0; k < 2; ++k) {
            const Image im = read_image_split(jobs[j].tok[k]);
            if (im.pd != 1 && im.pd != 3) throw std::runtime_error(jobs[j].tok[k] + ": gray or RGB images only");
            if (j == 0 && k == 0) {
                w = im.w; h = im.h;
                frames.assign((size_t)n * 6 * w * h, 0);
                flows.assign((size_t)n * planes * w * h, 0.0f);
            }
            if (im.w != w || im.h != h) throw std::runtime_error(jobs[j].tok[k] + ": all frames of a batch must have one size");
            const size_t px = (size_t)w * h;
            uint8_t *dst = frames.data() + ((size_t)j * 2 + k) * 3 * px;
            for (int c = 0; c < 3; ++c)
                for (size_t q = 0; q < px; ++q) {
                    const float v = im.data[(size_t)(im.pd == 3 ? c : 0) * px + q];
                    dst[(size_t)c * px + q] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : (int)(v + 0.5f));
                }
        }
        const size_t px = (size_t)w * h;
        for (int k = 0; k < planes / 2; ++k) {
            const std::string &name = jobs[j].tok[2 + k];
            float *dst = flows.data() + ((size_t)j * planes + 2 * k) * px;
            if (planes == 2 && name == "zero") continue;
            const Image fl = read_flo(name);
            if (fl.w != w || fl.h != h) throw std::runtime_error(name + ": flow size differs from the frames'");
            memcpy(dst, fl.data.data(), sizeof(float) * 2 * px);
        }
    }
}

int run_rank(Comm &comm, int device, const Options &o, const std::string &jobs_file)
{
    Header hd{};
    std::vector<Job> jobs;
    std::vector<uint8_t> frames;
    std::vector<float> flows;
    if (comm.rank == 0) {
        // a failure while reading the inputs must not leave the other ranks blocked: rank 0 always broadcasts a header
        try {
            jobs = parse_jobs(jobs_file);
            int w, h, planes;
            load_inputs(jobs, w, h, planes, frames, flows);
            hd = Header{ 0, (int32_t)jobs.size(), w, h, planes };
        } catch (const std::exception &e) {
            fprintf(stderr, "faldoi_batch: cannot read the jobs: %s\n", e.what());
            hd.status = 1;
        }
    }
    { Watch w("broadcast of the job table (rank 0 reads the inputs first)", comm.rank); comm.bcast(&hd, sizeof hd); }
    if (hd.status) return 3;
    const int njobs = hd.njobs, w = hd.w, h = hd.h, planes = hd.planes, world = comm.world;
    const size_t px = (size_t)w * h;
    const int cap = (njobs + world - 1) / world;                        // equal-size messages: the short shards are padded
    const std::vector<int> mine = shard_of(njobs, world, comm.rank);
    const size_t fr_chunk = (size_t)cap * 6 * px, fl_chunk = (size_t)cap * planes * px * sizeof(float);
    std::vector<uint8_t> fr_all, fr_mine(fr_chunk);
    std::vector<float> fl_all, fl_mine((size_t)cap * planes * px);
    if (comm.rank == 0) {
        fr_all.assign(fr_chunk * world, 0);
        fl_all.assign((size_t)cap * planes * px * world, 0.0f);
        for (int r = 0; r < world; ++r) {
            const std::vector<int> idx = shard_of(njobs, world, r);
            for (size_t k = 0; k < idx.size(); ++k) {
                memcpy(fr_all.data() + r * fr_chunk + k * 6 * px, frames.data() + (size_t)idx[k] * 6 * px, 6 * px);
                memcpy(fl_all.data() + ((size_t)r * cap + k) * planes * px, flows.data() + (size_t)idx[k] * planes * px, sizeof(float) * planes * px);
            }
        }
    }
    {
        Watch w("distribution of the shards", comm.rank);
        comm.scatter(fr_all.data(), fr_mine.data(), fr_chunk);
        comm.scatter(fl_all.data(), fl_mine.data(), fl_chunk);
    }
    if (const char *v = getenv("FALDOI_BATCH_TEST_HANG_RANK"))           // test hook: this rank stops taking part (a hung peer)
        if (atoi(v) == comm.rank) for (;;) pause();
    if (const char *v = getenv("FALDOI_BATCH_TEST_DIE_RANK"))            // test hook: this rank dies between two collectives
        if (atoi(v) == comm.rank) _exit(9);
    std::vector<float> out_mine((size_t)cap * 2 * px, 0.0f), out_all;
    int status = 0;
    try {
        solve_shard(device, o, (int)mine.size(), w, h, planes, fr_mine.data(), fl_mine.data(), out_mine.data());
    } catch (const std::exception &e) {
        fprintf(stderr, "faldoi_batch: rank %d: %s\n", comm.rank, e.what());
        status = 3;                                                     // still takes part in the gather: nobody is left waiting
    }
    if (comm.rank == 0) out_all.resize((size_t)cap * 2 * px * world);
    std::vector<int32_t> st_all(world, 0);
    const int32_t st_mine = status;
    {
        Watch w("collection of the flows (waits for the slowest rank's solve)", comm.rank);
        comm.gather(out_mine.data(), out_all.data(), sizeof(float) * cap * 2 * px);
        comm.gather(&st_mine, st_all.data(), sizeof(int32_t));
    }
    if (comm.rank == 0) {
        for (int r = 0; r < world; ++r) if (st_all[r]) status = st_all[r];
        if (!status) {
            for (int r = 0; r < world; ++r) {
                const std::vector<int> idx = shard_of(njobs, world, r);
                for (size_t k = 0; k < idx.size(); ++k)
                    write_flo(jobs[idx[k]].tok.back(), out_all.data() + ((size_t)r * cap + k) * 2 * px, w, h);
            }
            printf("faldoi_batch: %d pairs on %d GPU(s) written\n", njobs, world);
        }
    }
    int32_t fin = status;
    {
        Watch w("final verdict", comm.rank);
        comm.bcast(&fin, sizeof fin);                                   // every rank leaves with rank 0's verdict
        comm.barrier();
    }
    return fin;
}

// Rank 0 of a self-started run collects its ranks.  When it leaves with an error itself, or a rank does, the ranks that are still
// running are waiting for somebody who will not come: they are given a moment to notice (their own watchdogs and pipe errors), then ended.
int reap(const std::vector<pid_t> &kids, int rc)
{
    std::vector<pid_t> left(kids);
    const auto t0 = std::chrono::steady_clock::now();
    while (!left.empty()) {
        for (size_t k = 0; k < left.size();) {
            int st = 0;
            const pid_t r = waitpid(left[k], &st, WNOHANG);
            if (r == 0) { ++k; continue; }
            if (r < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != rc) rc = rc ? rc : ((r > 0 && WIFEXITED(st) && WEXITSTATUS(st)) ? WEXITSTATUS(st) : 3);
            left.erase(left.begin() + (long)k);
        }
        if (left.empty()) break;
        if (rc && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
            for (pid_t k : left) (void)kill(k, SIGKILL);
            for (pid_t k : left) { int st = 0; (void)waitpid(k, &st, 0); }
            break;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    return rc;
}

}  // namespace

int main(int argc, char *argv[])
{
    std::vector<std::string> args(argv, argv + argc);
    // faldoi_batch --rendezvous-selftest FILE publish|fetch|fetch-sibling TOKEN: the launcher-mode rendezvous alone (CPU tests)
    if (argc == 5 && std::string(argv[1]) == "--rendezvous-selftest") {
        try {
            RendezvousFile rv;
            rv.path = argv[2];
            char tok[64] = {0};
            const std::string mode = argv[3];
            if (mode == "publish") {
                strncpy(tok, argv[4], sizeof tok - 1);
                rv.publish(tok, sizeof tok);
                printf("published\n");
                fflush(stdout);
                (void)getchar();                                        // holds the record (and stays alive) until stdin closes
                return 0;
            }
            rv.fetch(tok, sizeof tok, mode == "fetch-sibling");
            printf("%s\n", tok);
            return 0;
        } catch (const std::exception &e) {
            fprintf(stderr, "faldoi_batch: %s\n", e.what());
            return 3;
        }
    }
    Options o;
    int gpus = 1;
    try {
        gpus = std::stoi(pick_option(args, "gpus", "1"));
        o.warps = std::stoi(pick_option(args, "w", "5"));
        o.max_inner = std::stoi(pick_option(args, "max_inner", "400"));
        o.tol_stop = std::stoi(pick_option(args, "tol_stop", "1"));
        o.split_img = std::stoi(pick_option(args, "split_img", "0"));
        o.h_parts = std::stoi(pick_option(args, "h_parts", "3"));
        o.v_parts = std::stoi(pick_option(args, "v_parts", "2"));
        o.first_h_parts = std::stoi(pick_option(args, "first_h_parts", "0"));
        o.first_v_parts = std::stoi(pick_option(args, "first_v_parts", "0"));
    } catch (const std::exception &) {
        args.clear();
    }
    if (args.size() != 2 || gpus < 1 || gpus > 64) {
        fprintf(stderr, "usage: %s jobs.txt [-gpus N] [-w 5] [-max_inner 400] [-tol_stop 1] [-split_img 0|1|2] [-h_parts 3] [-v_parts 2] "
                        "[-first_h_parts 0] [-first_v_parts 0]\n", argv[0]);
        return 1;
    }
    const std::string jobs_file = args[1];
    const char *env_world = getenv("WORLD_SIZE");
    try {
        if (env_world) {
#ifdef FALDOI_BATCH_STUB
            fprintf(stderr, "faldoi_batch (stub build): start it with -gpus N\n");
            return 1;
#else
            // ---- under an external launcher: the unique id travels through a file that is unique per run (RendezvousFile)
            RcclComm comm;
            comm.world = atoi(env_world);
            comm.rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
            const int device = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : comm.rank;
            const bool named = getenv("FALDOI_BATCH_ID_FILE") != nullptr;
            RendezvousFile rv;
            rv.path = named ? getenv("FALDOI_BATCH_ID_FILE") : rendezvous_default_name();
            ncclUniqueId id;
            {
                Watch w("rendezvous and communicator set-up", comm.rank);
                if (comm.rank == 0) {
                    (void)unlink(rv.path.c_str());                      // before anything else: no rank may pick up an earlier run's id
                    NCCL_OK(ncclGetUniqueId(&id));
                    rv.publish(&id, sizeof id);
                } else {
                    rv.fetch(&id, sizeof id, !named);
                }
                comm.init(device, id);
            }
            return run_rank(comm, device, o, jobs_file);                // (rv's destructor unlinks the file on every way out)
#endif
        }
        // ---- self-started ranks: fork BEFORE anything touches a GPU; rank 0 hands the unique id (RCCL) or nothing (stub) down pipes
        std::vector<int> up(gpus, -1), down(gpus, -1);                  // socket pair per rank r >= 1: up[r] parent side (rank 0), down[r] child side
        for (int r = 1; r < gpus; ++r) {
            int sv[2];
            if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) throw std::runtime_error("socketpair failed");
            up[r] = sv[0];
            down[r] = sv[1];
        }
        std::vector<pid_t> kids;
        int my_rank = 0;
        for (int r = 1; r < gpus; ++r) {
            const pid_t pid = fork();
            if (pid < 0) throw std::runtime_error("fork failed");
            if (pid == 0) {
                my_rank = r;
                kids.clear();
                (void)prctl(PR_SET_PDEATHSIG, SIGKILL);                 // a rank never outlives rank 0, however that one ends
                if (getppid() == 1) _exit(3);
                break;
            }
            kids.push_back(pid);
        }
        if (my_rank == 0) Watchdog::get().set_children(kids);
        for (int r = 1; r < gpus; ++r) {                                // keep only this rank's ends
            if (my_rank == 0) close(down[r]);
            else { close(up[r]); if (r != my_rank) close(down[r]); }
        }
        int rc;
#ifdef FALDOI_BATCH_STUB
        PipeComm comm;
        comm.rank = my_rank;
        comm.world = gpus;
        comm.fd.assign(gpus, -1);
        if (my_rank == 0) for (int r = 1; r < gpus; ++r) comm.fd[r] = up[r];
        else comm.fd[0] = down[my_rank];
        try {
            rc = run_rank(comm, my_rank, o, jobs_file);
        } catch (const std::exception &e) {
            fprintf(stderr, "faldoi_batch: rank %d: %s\n", my_rank, e.what());
            rc = 3;
        }
#else
        try {
            ncclUniqueId id;
            RcclComm comm;
            comm.rank = my_rank;
            comm.world = gpus;
            {
                Watch w("hand-over of the RCCL id and communicator set-up", my_rank);
                if (my_rank == 0) {
                    NCCL_OK(ncclGetUniqueId(&id));
                    for (int r = 1; r < gpus; ++r)
                        if (write(up[r], &id, sizeof id) != (ssize_t)sizeof id) throw std::runtime_error("cannot hand the RCCL id to a rank");
                } else if (read(down[my_rank], &id, sizeof id) != (ssize_t)sizeof id) {
                    throw std::runtime_error("no RCCL id from rank 0");
                }
                comm.init(my_rank, id);
            }
            rc = run_rank(comm, my_rank, o, jobs_file);
        } catch (const std::exception &e) {
            fprintf(stderr, "faldoi_batch: rank %d: %s\n", my_rank, e.what());
            rc = 3;
        }
#endif
        if (my_rank != 0) _exit(rc);
        return reap(kids, rc);
    } catch (const std::exception &e) {
        fprintf(stderr, "faldoi_batch: %s\n", e.what());
        return 3;
    }
}
