// faldoi_oracle_growing.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the local step of FALDOI: match_growing_variational
// (src/local_faldoi.cpp:1060-1741) for the TV-l2 coupled functional (-m 0) without image
// partitioning (-split_img 0) and without occlusions: seeds -> priority-queue growing
// (forward and backward) -> forward-backward pruning -> re-seeding, `loc_it` times, then a last
// forward growing.  The per-window operators (Poisson filling, of_estimation) are the C
// restatements of faldoi_oracle_local.c, which are pinned bit-for-bit against the reference's own
// objects.  The queue is std::priority_queue with the reference's comparator, i.e. exactly the
// container the reference uses (src/energy_structures.h:13-30), so pop order (ties included) is
// the reference's by construction.
//
// PARITY UNPINNED at this level: src/local_faldoi.cpp includes <boost/lexical_cast.hpp>, which
// this image lacks, so the orchestration cannot be compiled from the reference here, and the
// seed-dependent known answers of SURVEY App. D need the matcher binaries (not runnable here).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <set>
#include <vector>

extern "C" {
#include "faldoi_oracle.h"
}

namespace {

struct Candidate {                 // SparseOF, src/energy_structures.h:13-20
    int i, j;                      // column, row
    float u, v, sim;
    long id;                       // study only (speculation model below): order of insertion; never compared
};
struct CandidateOrder {            // CompareSparseOF :22-27: smallest energy on top
    bool operator()(const Candidate &a, const Candidate &b) const { return a.sim > b.sim; }
};
typedef std::priority_queue<Candidate, std::vector<Candidate>, CandidateOrder> CandidateQueue;

struct Direction {                 // OpticalFlowData + Tvl2CoupledOFStuff + the per-direction arrays of :1082-1095
    int w, h;
    const float *src, *dst;        // normalised source / target frame (i0n,i1n forward; i1n,i0n backward)
    const float *sal;              // saliency or NULL (= 1)
    std::vector<float> own_src, own_dst;   // partition crops of the frames (empty for the whole image)
    std::vector<float> dstx, dsty; // centred gradient of the target frame (prepare_stuff)
    std::vector<float> u;          // ofD->u1 | u2
    std::vector<float> state;      // xi11..xi22, div1, div2 (persistent, see faldoi_oracle_local.c)
    std::vector<float> ene, out;   // ene_Go / oft0
    std::vector<int> fixed, trust;
    CandidateQueue queue;
    long pops, solves, fills;
    int max_iter_patch, wr;
    int method;                    // 0: TV-l2 coupled, 1: with weights (M_TVL1_W), 2..7: the other functionals of of_estimation
    std::vector<float> weight;     // gaussian1Dweight(params.w_radio), prepare_stuff src/energy_model.cpp:429-436
    std::vector<float> duals;      // -m 2, 3, 6, 7: p.sc | q.sc, [48][n], zero where prepare_stuff zeroes them (persistent)
    const float *wp;               // ... and the constant non-local weights [24][n] (shared by both directions: Lab of i0)
};

struct Window { int ii, ij, ei, ej; };

Window window_at(int wr, int w, int h, int i, int j)      // get_index_patch, src/utils.cpp:36-54
{
    Window x;
    x.ii = (i - wr < 0) ? 0 : i - wr;
    x.ij = (j - wr < 0) ? 0 : j - wr;
    x.ei = (i + 1 + wr > w) ? w : i + 1 + wr;
    x.ej = (j + 1 + wr > h) ? h : j + 1 + wr;
    return x;
}

// ---- study (tools/growing_dependencies.py --spec S): a cost model of a committer + S speculative window solvers working on
// one growing.  It observes the real run (which stays untouched) and adds up modelled cycles: the committer pops in the
// true order; a free solver takes the best queued candidate nobody works on; a popped candidate whose solver has finished
// and whose window no commit has touched since that solver started costs only the queue work and the write-back.
struct SpecModel {
    int S = 0;
    const Direction *dir = NULL;
    double T = 0, T_serial = 0;
    long hits = 0, misses = 0, waits = 0, windows = 0, serial = 0;
    struct Solver { long cand = -1; double done = 0; long start_serial = 0; };
    std::vector<Solver> solver;
    std::vector<int> log_i, log_j;                       // commit log by serial number
    std::set<std::pair<float, long> > order;             // (energy, id) of the queued candidates: a stand-in for "the top of the queue"
    std::vector<Candidate> byid;
    double c_pop = 4500, c_push = 2200, c_fill = 31700, c_solve = 15500, c_apply = 2500;
};
static SpecModel g_spec;
extern "C" void orc_spec_model(int solvers) { g_spec = SpecModel(); g_spec.S = solvers; g_spec.solver.resize(solvers > 0 ? solvers : 0); }
extern "C" void orc_spec_report(double *out)
{
    out[0] = g_spec.T; out[1] = g_spec.T_serial; out[2] = (double)g_spec.hits; out[3] = (double)g_spec.misses;
    out[4] = (double)g_spec.waits; out[5] = (double)g_spec.windows;
}
static void spec_on_push(Direction &d, Candidate &c)
{
    if (g_spec.S <= 0 || (g_spec.dir && g_spec.dir != &d)) return;
    g_spec.dir = &d;
    c.id = (long)g_spec.byid.size();
    g_spec.byid.push_back(c);
    g_spec.order.insert(std::make_pair(c.sim, c.id));
    g_spec.T += g_spec.c_push;
    g_spec.T_serial += g_spec.c_push;
}
static bool spec_needs_fill(const Direction &d, int i, int j, int iteration)
{
    if (iteration == 0) return true;
    const Window x = window_at(d.wr, d.w, d.h, i, j);
    for (int l = x.ij; l < x.ej; l++)
        for (int k = x.ii; k < x.ei; k++)
            if (d.trust[l * d.w + k] == 0) return true;
    return false;
}
// called at every pop of the observed growing, before the window is processed
static void spec_on_pop(Direction &d, const Candidate &c, bool stale, int iteration)
{
    SpecModel &m = g_spec;
    if (m.S <= 0 || m.dir != &d) return;
    m.order.erase(std::make_pair(c.sim, c.id));
    m.T += m.c_pop;
    m.T_serial += m.c_pop;
    if (stale) return;
    const double full = (spec_needs_fill(d, c.i, c.j, iteration) ? m.c_fill : 0.0) + m.c_solve;
    m.T_serial += full;
    m.windows++;
    bool served = false;
    for (SpecModel::Solver &sv : m.solver)
        if (sv.cand == c.id) {
            bool valid = true;
            for (long q = sv.start_serial; q < m.serial && valid; q++)
                if (std::abs(m.log_i[q] - c.i) <= 2 * d.wr && std::abs(m.log_j[q] - c.j) <= 2 * d.wr) valid = false;
            if (valid && sv.done <= m.T) { m.T += m.c_apply; m.hits++; served = true; }
            else if (valid && sv.done + m.c_apply < m.T + full) { m.T = sv.done + m.c_apply; m.waits++; served = true; }
            sv.cand = -1;
        }
    if (!served) { m.T += full; m.misses++; }
    m.log_i.push_back(c.i);
    m.log_j.push_back(c.j);
    m.serial++;
}
// after the window's pushes: idle solvers pick work
static void spec_assign(Direction &d, int iteration)
{
    SpecModel &m = g_spec;
    if (m.S <= 0 || m.dir != &d) return;
    for (SpecModel::Solver &sv : m.solver) {
        if (sv.cand >= 0 && sv.done > m.T) continue;                      // busy
        if (sv.cand >= 0) continue;                                       // holds a finished result until its candidate is popped
        int looked = 0;
        for (std::set<std::pair<float, long> >::iterator it = m.order.begin(); it != m.order.end() && looked < 4 * m.S + 4; ++it, ++looked) {
            const Candidate &q = m.byid[it->second];
            if (d.fixed[q.j * d.w + q.i]) continue;
            bool taken = false;
            for (const SpecModel::Solver &o : m.solver) taken = taken || o.cand == q.id;
            if (taken) continue;
            if (looked == 0) continue;                                    // the very next pop is the committer's own
            sv.cand = q.id;
            sv.start_serial = m.serial;
            sv.done = m.T + (spec_needs_fill(d, q.i, q.j, iteration) ? m.c_fill : 0.0) + m.c_solve;
            break;
        }
    }
    // a held result whose candidate became stale (its pixel got fixed) or invalid is dropped
    for (SpecModel::Solver &sv : m.solver)
        if (sv.cand >= 0) {
            const Candidate &q = m.byid[sv.cand];
            bool drop = d.fixed[q.j * d.w + q.i] != 0;
            for (long k = sv.start_serial; k < m.serial && !drop; k++)
                if (std::abs(m.log_i[k] - q.i) <= 2 * d.wr && std::abs(m.log_j[k] - q.j) <= 2 * d.wr) drop = true;
            if (drop && sv.done <= m.T) sv.cand = -1;
        }
}

// add_neighbors, src/local_faldoi.cpp:679-727
void add_neighbors(Direction &d, int i, int j, int iteration, int wr)
{
    const int w = d.w, h = d.h, n = w * h;
    const Window x = window_at(wr, w, h, i, j);
    float *u1 = d.u.data(), *u2 = u1 + n;
    bool fill = (iteration == 0);
    if (!fill)                                                           // check_trustable_patch :639-656
        for (int l = x.ij; l < x.ej && !fill; l++)
            for (int k = x.ii; k < x.ei; k++)
                if (d.trust[l * w + k] == 0) { fill = true; break; }
    if (fill) {
        for (int l = x.ij; l < x.ej; l++)                                // copy_fixed_coordinates :609-628
            for (int k = x.ii; k < x.ei; k++) {
                const int p = l * w + k;
                if (d.fixed[p] == 1) { u1[p] = d.out[p]; u2[p] = d.out[n + p]; }
            }
        orc_interpolate_poisson(u1, u2, d.fixed.data(), w, x.ii, x.ij, x.ei, x.ej);
        d.fills++;
    }
    float ener;
    if (d.method >= 2) {                                                 // of_estimation's other branches, src/energy_model.cpp:704-795
        const int iiw = (i - wr < 0) ? -(i - wr) : 0, ijw = (j - wr < 0) ? -(j - wr) : 0;
        ener = orc_local_nl_estimate(d.method, d.src, d.dst, d.dstx.data(), d.dsty.data(), u1, u2,
                                     d.duals.empty() ? NULL : d.duals.data(), d.wp, w, h, x.ii, x.ij, x.ei, x.ej, d.max_iter_patch,
                                     d.weight.data(), iiw, ijw, wr, NULL);
    } else if (d.method == 1) {                                          // get_index_weight :571-597 with this call's radius
        const int iiw = (i - wr < 0) ? -(i - wr) : 0, ijw = (j - wr < 0) ? -(j - wr) : 0;
        ener = orc_local_tvl1w_estimate(d.src, d.dst, d.dstx.data(), d.dsty.data(), u1, u2, d.state.data(), w, h, x.ii, x.ij, x.ei,
                                        x.ej, d.max_iter_patch, NULL, d.weight.data(), iiw, ijw, wr);
    } else {
        ener = orc_local_tvl1_estimate(d.src, d.dst, d.dstx.data(), d.dsty.data(), u1, u2, d.state.data(), w, h, x.ii, x.ij, x.ei,
                                       x.ej, d.max_iter_patch, NULL);
    }
    d.solves++;
    static const int nb[4][2] = { { 0, 1 }, { 0, -1 }, { 1, 0 }, { -1, 0 } };   // insert_candidates :497-537
    for (int k = 0; k < 4; k++) {
        const int px = i + nb[k][0], py = j + nb[k][1];
        if (px < 0 || px >= w || py < 0 || py >= h) continue;
        const int q = py * w + px;
        const float new_ener = ener * (d.sal ? d.sal[q] : 1.0f);
        if (!d.fixed[q] && new_ener < d.ene[q]) {
            d.ene[q] = ener;
            Candidate c = { px, py, u1[q], u2[q], new_ener, 0 };
            spec_on_push(d, c);
            d.queue.push(c);
        }
    }
    const int p = j * w + i;                                             // :718-726
    if (d.ene[p] > ener) {
        d.out[p] = u1[p];
        d.out[n + p] = u2[p];
        d.ene[p] = ener;
    }
}

void reset_growing(Direction &d)                                         // :755-761, :839-845, :860-870
{
    const int n = d.w * d.h;
    for (int p = 0; p < n; p++) {
        d.fixed[p] = 0;
        d.ene[p] = INFINITY;
        d.out[p] = d.out[n + p] = NAN;
    }
}

// insert_initial_seeds, src/local_faldoi.cpp:748-796
void insert_initial_seeds(Direction &d, const float *in_flow)
{
    const int w = d.w, h = d.h, n = w * h;
    reset_growing(d);
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            const int p = j * w + i;
            if (std::isfinite(in_flow[p]) && std::isfinite(in_flow[n + p])) {
                d.out[p] = in_flow[p];
                d.out[n + p] = in_flow[n + p];
                d.fixed[p] = 1;
                add_neighbors(d, i, j, 0, 1);                            // w_radio = 1 for the seeds, :764
                d.out[p] = d.out[n + p] = NAN;
                d.fixed[p] = 0;
            }
        }
    for (int p = 0; p < n; p++)
        if (std::isfinite(in_flow[p]) && std::isfinite(in_flow[n + p])) {
            d.out[p] = in_flow[p];
            d.out[n + p] = in_flow[n + p];
            d.fixed[p] = 1;
            d.ene[p] = 0.0f;
        }
}

// Study hook (tools/growing_dependencies.py): the order in which the forward growing minimises its windows.
static int *g_trace = NULL;
static long g_trace_cap = 0, g_trace_len = 0;
static const Direction *g_trace_dir = NULL;
extern "C" void orc_trace_windows(int *buf, long cap) { g_trace = buf; g_trace_cap = cap; g_trace_len = 0; g_trace_dir = NULL; }
extern "C" long orc_trace_length(void) { return g_trace_len; }

// -partial_res 1 (src/local_faldoi.cpp:942-1036): the forward growings' flow at the reference's moments, kept in memory instead of
// "../Results/Partial_results/*.flo" (orc_partial_enable / _count / _get); the test `percent > P[k] && percent < P[k+1]` and the
// float percentage are the reference's, statement for statement.
struct PartialDump { int percent, iteration, part_idx, w, h; std::vector<float> flow; };
static bool g_partial_on = false;
static std::vector<PartialDump> g_partial;
static void partial_keep(const Direction &d, int percent, int iteration, int part_idx)
{
    PartialDump p{ percent, iteration, part_idx, d.w, d.h, std::vector<float>(d.out.begin(), d.out.begin() + 2 * (size_t)d.w * d.h) };
#pragma omp critical(orc_partial)
    g_partial.push_back(std::move(p));
}
extern "C" void orc_partial_enable(int on) { g_partial_on = on != 0; g_partial.clear(); }
extern "C" int orc_partial_count(void) { return (int)g_partial.size(); }
extern "C" void orc_partial_get(int k, int *meta /* percent, iteration, part_idx, w, h */, float *flow /* may be NULL */)
{
    const PartialDump &p = g_partial[(size_t)k];
    meta[0] = p.percent; meta[1] = p.iteration; meta[2] = p.part_idx; meta[3] = p.w; meta[4] = p.h;
    if (flow) memcpy(flow, p.flow.data(), sizeof(float) * p.flow.size());
}

// Study hook (round 4): how often could the window of the NEXT candidate have been computed beside the current one?  After a pop the
// queue's new top c' is looked at: it qualifies if it is not stale and its window does not overlap the current one; it "holds" if it is
// still the top once the current window's candidates have been pushed (then it is the next pop and its window read nothing the current
// one wrote).  A window that was served this way does not look ahead itself (the device pairs windows two by two).
static long g_twin[6];     // windows, attempts, holds, stale top, overlapping top, displaced by a push
extern "C" void orc_twin_stats(long *out, int reset) { for (int k = 0; k < 6; k++) { out[k] = g_twin[k]; if (reset) g_twin[k] = 0; } }
static bool same_candidate(const Candidate &a, const Candidate &b)
{
    return a.i == b.i && a.j == b.j && memcmp(&a.u, &b.u, 4) == 0 && memcmp(&a.v, &b.v, 4) == 0 && memcmp(&a.sim, &b.sim, 4) == 0;
}

// local_growing, src/local_faldoi.cpp:891-1039
void local_growing(Direction &d, int iteration, bool fwd = false, int part_idx = -1)
{
    bool twin_served = false;
    Candidate twin_next{};
    const int w = d.w, n = d.w * d.h;
    int percent_print[5] = { 30, 70, 80, 95, 100 };                       // :896
    int fixed = 0;
    const int size = n;
    while (!d.queue.empty()) {
        const Candidate c = d.queue.top();
        d.queue.pop();
        d.pops++;
        const int p = c.j * w + c.i;
        spec_on_pop(d, c, d.fixed[p] != 0, iteration);
        if (d.fixed[p]) continue;
        d.fixed[p] = 1;
        d.out[p] = c.u;
        d.out[n + p] = c.v;
        d.ene[p] = c.sim;
        if (g_trace && (g_trace_dir == NULL || g_trace_dir == &d)) {
            g_trace_dir = &d;                                             // the first growing that gets here (one direction)
            if (g_trace_len + 3 <= g_trace_cap) { g_trace[g_trace_len] = c.i; g_trace[g_trace_len + 1] = c.j; g_trace[g_trace_len + 2] = iteration; }
            g_trace_len += 3;
        }
        fixed++;                                                          // :925
        bool attempt = false;
        Candidate peek{};
        g_twin[0]++;
        if (twin_served && same_candidate(c, twin_next)) {
            twin_served = false;                                          // this window was computed beside its predecessor
        } else {
            twin_served = false;
            if (!d.queue.empty()) {
                peek = d.queue.top();
                const int pp = peek.j * w + peek.i;
                if (d.fixed[pp]) g_twin[3]++;
                else if (std::abs(peek.i - c.i) <= 2 * d.wr && std::abs(peek.j - c.j) <= 2 * d.wr) g_twin[4]++;
                else { attempt = true; g_twin[1]++; }
            }
        }
        add_neighbors(d, c.i, c.j, iteration, d.wr);
        if (attempt) {
            if (!d.queue.empty() && same_candidate(d.queue.top(), peek)) { g_twin[2]++; twin_served = true; twin_next = peek; }
            else g_twin[5]++;
        }
        spec_assign(d, iteration);
        if (g_partial_on) {                                               // :942-994
            float percent = 100 * fixed * 1.0 / size * 1.0;
            for (int k = 0; k < 4; k++)
                if (percent > percent_print[k] && percent < percent_print[k + 1])
                    if (fwd) {
                        partial_keep(d, percent_print[k], iteration, part_idx);
                        percent_print[k] = 200;
                    }
        }
    }
    if (g_partial_on && fwd) partial_keep(d, 100, iteration, part_idx);    // :996-1035
}

// fb_consistency_check, src/local_faldoi.cpp:167-190: |u(x) + v(x + u(x))| <= epsilon
double fb_consistency_check(const float *in0, const float *in1, int *trust, int w, int h, float epsilon)
{
    const int n = w * h;
    std::vector<float> u1w(n), u2w(n);
    orc_bicubic_warp(in1, in0, in0 + n, u1w.data(), w, h, 1);
    orc_bicubic_warp(in1 + n, in0, in0 + n, u2w.data(), w, h, 1);
    long kept = 0;
    for (int p = 0; p < n; p++) {
        const float tolerance = hypotf(in0[p] + u1w[p], in0[n + p] + u2w[p]);
        trust[p] = (tolerance > epsilon) ? 0 : 1;
        kept += trust[p];
    }
    return (double)kept / n;
}

void delete_not_trustable_candidates(Direction &d)                       // :283-311
{
    const int n = d.w * d.h;
    for (int p = 0; p < n; p++)
        if (d.trust[p] == 0) {
            d.out[p] = d.out[n + p] = NAN;
            d.u[p] = d.u[n + p] = NAN;
            d.ene[p] = INFINITY;
        }
}

void insert_potential_candidates(Direction &d)                           // :813-846 (in == out_flow at the call site)
{
    const int w = d.w, h = d.h, n = w * h;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            const int p = j * w + i;
            if (std::isfinite(d.out[p]) && std::isfinite(d.out[n + p])) {
                Candidate c = { i, j, d.out[p], d.out[n + p], d.ene[p], 0 };
                spec_on_push(d, c);
                d.queue.push(c);
            }
        }
    reset_growing(d);
}

void init_direction(Direction &d, const float *src, const float *dst, const float *sal, int w, int h, int wr,
                    int max_iter_patch, int method)
{
    d.method = method;
    d.weight.assign(2 * wr + 1, 0.0f);
    orc_gaussian1d_weight(d.weight.data(), wr);
    const size_t n = (size_t)w * h;
    d.w = w; d.h = h; d.src = src; d.dst = dst; d.sal = sal; d.wr = wr; d.max_iter_patch = max_iter_patch;
    d.dstx.assign(n, 0.0f); d.dsty.assign(n, 0.0f);
    orc_centered_gradient(dst, d.dstx.data(), d.dsty.data(), w, h);     // prepare_stuff, src/energy_model.cpp:660-684
    d.u.assign(2 * n, 0.0f);       // `new float[]` in the reference (:186): fresh zero pages in practice, and
                                   // every window is filled before it is first read
    d.state.assign(6 * n, 0.0f);
    d.ene.assign(n, 0.0f); d.out.assign(2 * n, 0.0f);
    d.fixed.assign(n, 0); d.trust.assign(n, 1);
    d.pops = d.solves = d.fills = 0;
    d.wp = NULL;
    d.duals.clear();
}

}  // namespace

// Hooks for unit tests of the device heap: the reference's container, driven step by step.
extern "C" void *orc_queue_new(void) { return new CandidateQueue(); }
extern "C" void orc_queue_free(void *q) { delete (CandidateQueue *)q; }
extern "C" void orc_queue_push(void *q, int i, int j, float u, float v, float sim)
{
    Candidate c = { i, j, u, v, sim, 0 };
    ((CandidateQueue *)q)->push(c);
}
extern "C" int orc_queue_pop(void *q, int *i, int *j, float *u, float *v, float *sim)
{
    CandidateQueue *pq = (CandidateQueue *)q;
    if (pq->empty()) return 0;
    const Candidate c = pq->top();
    pq->pop();
    *i = c.i; *j = c.j; *u = c.u; *v = c.v; *sim = c.sim;
    return 1;
}

extern "C" double orc_fb_consistency_check(const float *in0, const float *in1, int *trust, int w, int h, float epsilon)
{
    return fb_consistency_check(in0, in1, trust, w, h, epsilon);
}

/* i0n, i1n: normalised frames (orc_preprocess_local).  go, ba: seed flows (2 planes, NaN = no seed).
 * sal0, sal1: saliency planes or NULL.  out_flow (2 planes) and ene_val receive oft0 / ene_Go.
 * stats (may be NULL): [0..2] forward pops / patch solves / Poisson fills, [3..5] backward. */
namespace {

// ---- image partitions (-split_img 1), src/aux_partitions.cpp ----
struct Partition {
    int w, h, ox, oy;
    Direction d[2];
    long queued[2];                // queue_Go_size / queue_Ba_size: never reset (src/aux_partitions.cpp:1998,2021)
};

// init_subimage_partitions, src/aux_partitions.cpp:47-270: a cols x rows grid; the remainders go to the last column / row
void make_partitions(std::vector<Partition> &P, int cols, int rows, const Direction *img)
{
    const int w = img[0].w, h = img[0].h;
    P.resize((size_t)cols * rows);
    const int sw = w / cols, sh = h / rows;
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) {
            Partition &p = P[(size_t)r * cols + c];
            p.w = sw + (c == cols - 1 ? w % cols : 0);
            p.h = sh + (r == rows - 1 ? h % rows : 0);
            p.ox = c * sw;
            p.oy = r * sh;
            p.queued[0] = p.queued[1] = 0;
            for (int k = 0; k < 2; k++) {
                Direction &d = p.d[k];
                const size_t n = (size_t)p.w * p.h;
                d.own_src.resize(n); d.own_dst.resize(n);
                for (int j = 0; j < p.h; j++)
                    for (int i = 0; i < p.w; i++) {
                        d.own_src[(size_t)j * p.w + i] = img[k].src[(size_t)(p.oy + j) * w + p.ox + i];
                        d.own_dst[(size_t)j * p.w + i] = img[k].dst[(size_t)(p.oy + j) * w + p.ox + i];
                    }
                init_direction(d, d.own_src.data(), d.own_dst.data(), NULL, p.w, p.h, img[k].wr, img[k].max_iter_patch, img[k].method);
            }
        }
}

// image_to_partitions, src/aux_partitions.cpp:2056-2156 (with update_of_data :285 and update_tvl2_stuffof :403):
// every per-pixel state plane is copied between the image and the partitions, the gradients of the target frame
// included (so a partition works on crops of the image's planes); to_part: the candidates are distributed too.
void image_to_partitions(std::vector<Partition> &P, Direction *img, bool to_part)
{
    const int w = img[0].w;
    const size_t n = (size_t)w * img[0].h;
    for (Partition &p : P) {
        const size_t m = (size_t)p.w * p.h;
        for (int k = 0; k < 2; k++) {
            Direction &g = img[k], &d = p.d[k];
            for (int j = 0; j < p.h; j++)
                for (int i = 0; i < p.w; i++) {
                    const size_t a = (size_t)(p.oy + j) * w + p.ox + i, b = (size_t)j * p.w + i;
                    if (to_part) {
                        d.fixed[b] = g.fixed[a]; d.trust[b] = g.trust[a];
                        d.u[b] = g.u[a]; d.u[m + b] = g.u[n + a];
                        for (int q = 0; q < 6; q++) d.state[q * m + b] = g.state[q * n + a];
                        d.dstx[b] = g.dstx[a]; d.dsty[b] = g.dsty[a];
                        d.ene[b] = g.ene[a]; d.out[b] = g.out[a]; d.out[m + b] = g.out[n + a];
                    } else {
                        g.fixed[a] = d.fixed[b]; g.trust[a] = d.trust[b];
                        g.u[a] = d.u[b]; g.u[n + a] = d.u[m + b];
                        for (int q = 0; q < 6; q++) g.state[q * n + a] = d.state[q * m + b];
                        g.dstx[a] = d.dstx[b]; g.dsty[a] = d.dsty[b];
                        g.ene[a] = d.ene[b]; g.out[a] = d.out[b]; g.out[n + a] = d.out[m + b];
                    }
                }
        }
    }
    for (int k = 0; k < 2; k++) {
        if (to_part) {                                                   // update_candidate_queues :1970-2031
            CandidateQueue tmp = img[k].queue;
            while (!tmp.empty()) {
                Candidate c = tmp.top();
                const int i = c.i, j = c.j;
                for (Partition &p : P)
                    if (i >= p.ox && i <= p.ox + p.w - 1 && j >= p.oy && j <= p.oy + p.h - 1) {
                        c.i = i >= p.w ? i - p.ox : i;                   // sic (:1992-1993): not `i - off_x` whenever i < width
                        c.j = j >= p.h ? j - p.oy : j;
                        p.d[k].queue.push(c);
                        p.queued[k]++;
                    }
                tmp.pop();
            }
        } else {
            img[k].queue = CandidateQueue();
        }
    }
}

bool any_empty_queues(const std::vector<Partition> &P)                   // :2161-2180
{
    for (const Partition &p : P)
        if (p.queued[0] == 0 || p.queued[1] == 0) return true;
    return false;
}

void prune_and_reseed(Direction *d, int w, int h, float fb_thresh)       // :1231-1266
{
    fb_consistency_check(d[0].out.data(), d[1].out.data(), d[0].trust.data(), w, h, fb_thresh);   // pruning_method :209
    fb_consistency_check(d[1].out.data(), d[0].out.data(), d[1].trust.data(), w, h, fb_thresh);
    for (int k = 0; k < 2; k++) {
        delete_not_trustable_candidates(d[k]);
        insert_potential_candidates(d[k]);
    }
}

}  // namespace

// ---- test hooks for the partition bookkeeping (tests/test_oracle_cpu.py pins them against the reference's own
// init_subimage_partitions / image_to_partitions objects through oracle/_ref): an image-level state whose planes the
// test sets, one partition grid, transfers in both directions.  Plane ids: 0 u1, 1 u2, 2..5 xi11 xi12 xi21 xi22,
// 6 div_xi1, 7 div_xi2, 8 I1x, 9 I1y, 10 ene, 11 / 12 out flow u / v, 13 fixed, 14 trust (ints as floats).
namespace {
struct PartsHook {
    std::vector<float> i0n, i1n;
    Direction d[2];
    std::vector<Partition> P;
};
float *hook_plane(Direction &d, int id, int **ip)
{
    const size_t n = (size_t)d.w * d.h;
    *ip = NULL;
    if (id <= 1) return d.u.data() + id * n;
    if (id <= 7) return d.state.data() + (size_t)(id - 2) * n;
    if (id == 8) return d.dstx.data();
    if (id == 9) return d.dsty.data();
    if (id == 10) return d.ene.data();
    if (id <= 12) return d.out.data() + (size_t)(id - 11) * n;
    *ip = id == 13 ? d.fixed.data() : d.trust.data();
    return NULL;
}
void hook_copy(Direction &d, int id, float *v, bool set)
{
    int *ip;
    float *p = hook_plane(d, id, &ip);
    const size_t n = (size_t)d.w * d.h;
    for (size_t i = 0; i < n; i++) {
        if (set) { if (p) p[i] = v[i]; else ip[i] = (int)v[i]; }
        else v[i] = p ? p[i] : (float)ip[i];
    }
}
}  // namespace

extern "C" void *orc_parts_create(const float *i0n, const float *i1n, int w, int h, int cols, int rows, int wr)
{
    PartsHook *r = new PartsHook();
    r->i0n.assign(i0n, i0n + (size_t)w * h);
    r->i1n.assign(i1n, i1n + (size_t)w * h);
    init_direction(r->d[0], r->i0n.data(), r->i1n.data(), NULL, w, h, wr, 4, 0);
    init_direction(r->d[1], r->i1n.data(), r->i0n.data(), NULL, w, h, wr, 4, 0);
    for (int k = 0; k < 2; k++) {                       // the arrays of prepare_data_for_growing, src/local_faldoi.cpp:860-870
        std::fill(r->d[k].ene.begin(), r->d[k].ene.end(), INFINITY);
        std::fill(r->d[k].out.begin(), r->d[k].out.end(), NAN);
        std::fill(r->d[k].trust.begin(), r->d[k].trust.end(), 0);
    }
    make_partitions(r->P, cols, rows, r->d);
    return r;
}
extern "C" void orc_parts_destroy(void *h) { delete (PartsHook *)h; }
extern "C" int orc_parts_count(void *h) { return (int)((PartsHook *)h)->P.size(); }
extern "C" void orc_parts_geometry(void *h, int p, int *geom)
{
    const Partition &q = ((PartsHook *)h)->P[p];
    geom[0] = q.w; geom[1] = q.h; geom[2] = q.ox; geom[3] = q.oy;
}
extern "C" void orc_parts_set_image_plane(void *h, int dir, int id, const float *v)
{ hook_copy(((PartsHook *)h)->d[dir], id, const_cast<float *>(v), true); }
extern "C" void orc_parts_get_image_plane(void *h, int dir, int id, float *v) { hook_copy(((PartsHook *)h)->d[dir], id, v, false); }
extern "C" void orc_parts_set_part_plane(void *h, int part, int dir, int id, const float *v)
{ hook_copy(((PartsHook *)h)->P[part].d[dir], id, const_cast<float *>(v), true); }
extern "C" void orc_parts_get_part_plane(void *h, int part, int dir, int id, float *v)
{ hook_copy(((PartsHook *)h)->P[part].d[dir], id, v, false); }
extern "C" void orc_parts_get_part_frame(void *h, int part, int which, float *v)
{
    const Direction &d = ((PartsHook *)h)->P[part].d[0];             // forward: src = i0n crop, dst = i1n crop
    const std::vector<float> &f = which ? d.own_dst : d.own_src;
    memcpy(v, f.data(), sizeof(float) * f.size());
}
extern "C" void orc_parts_push(void *h, int dir, int i, int j, float u, float v, float sim)
{
    Candidate c = { i, j, u, v, sim, 0 };
    ((PartsHook *)h)->d[dir].queue.push(c);
}
extern "C" int orc_parts_image_queue_size(void *h, int dir) { return (int)((PartsHook *)h)->d[dir].queue.size(); }
extern "C" void orc_parts_transfer(void *h, int img_to_part)
{
    PartsHook *r = (PartsHook *)h;
    image_to_partitions(r->P, r->d, img_to_part != 0);
}
extern "C" int orc_parts_any_empty(void *h) { return any_empty_queues(((PartsHook *)h)->P) ? 1 : 0; }
extern "C" int orc_parts_drain_queue(void *h, int part, int dir, float *out, int cap)
{
    CandidateQueue &q = ((PartsHook *)h)->P[part].d[dir].queue;
    int k = 0;
    while (!q.empty() && k < cap) {
        const Candidate c = q.top();
        q.pop();
        out[5 * k] = (float)c.i; out[5 * k + 1] = (float)c.j; out[5 * k + 2] = c.u; out[5 * k + 3] = c.v; out[5 * k + 4] = c.sim;
        k++;
    }
    return k;
}

/* split_img = 0: whole-image growings.  split_img = 1 (-split_img 1 -h_parts H -v_parts V): local iteration 0 on the
 * whole image, odd iterations on an H x V grid of partitions, even ones on a V x H grid, the last growing on the H x V
 * grid (src/local_faldoi.cpp:1284-1700).  split_img = 2 (not in the reference): iteration 0 on the V x H grid too, or on
 * a first_h_parts x first_v_parts grid of its own when those are > 0.
 * stats as before, partitions included. */
extern "C" void orc_local_faldoi_split(const float *i0n, const float *i1n, const float *go, const float *ba, const float *sal0,
                                       const float *sal1, int w, int h, int wr, int loc_it, int max_iter_patch, float fb_thresh,
                                       int method, int split_img, int h_parts, int v_parts, int first_h_parts,
                                       int first_v_parts, float *out_flow, float *ene_val,
                                       long *stats)
{
    const size_t n = (size_t)w * h;
    Direction d[2];
    init_direction(d[0], i0n, i1n, sal0, w, h, wr, max_iter_patch, method);
    init_direction(d[1], i1n, i0n, sal1, w, h, wr, max_iter_patch, method);
#pragma omp parallel for num_threads(2)
    for (int k = 0; k < 2; k++) insert_initial_seeds(d[k], k == 0 ? go : ba);     // :1130-1139
    std::vector<Partition> PA, PB, PC;
    if (split_img) {                                                               // :1160-1172
        make_partitions(PA, h_parts, v_parts, d);
        make_partitions(PB, v_parts, h_parts, d);
        if (split_img == 2 && first_h_parts > 0) make_partitions(PC, first_h_parts, first_v_parts, d);   // not in the reference
    }
    auto grow_partitions = [&](std::vector<Partition> &P, int it, int ndir) {
        const int count = (int)P.size() * ndir;
#pragma omp parallel for schedule(dynamic)
        for (int q = 0; q < count; q++) local_growing(P[q / ndir].d[q % ndir], it, q % ndir == 0, q / ndir);
    };
    for (int it = 0; it < loc_it; it++) {                                          // :1184-1640
        if (!split_img || (it == 0 && split_img != 2)) {
#pragma omp parallel for num_threads(2)
            for (int k = 0; k < 2; k++) local_growing(d[k], it, k == 0, -1);
        } else {
            std::vector<Partition> &P = (it == 0 && !PC.empty()) ? PC : (it % 2 != 0) ? PA : PB;
            image_to_partitions(P, d, true);
            if (!any_empty_queues(P)) {
                grow_partitions(P, it, 2);
                image_to_partitions(P, d, false);
            } else {                                                               // :1349-1370: whole image instead
#pragma omp parallel for num_threads(2)
                for (int k = 0; k < 2; k++) local_growing(d[k], it, k == 0, -1);
            }
        }
        prune_and_reseed(d, w, h, fb_thresh);
    }
    if (split_img) {                                                               // :1642-1693
        image_to_partitions(PA, d, true);
        if (!any_empty_queues(PA)) {
            grow_partitions(PA, loc_it, 1);
            image_to_partitions(PA, d, false);
        } else {
            local_growing(d[0], loc_it, true, loc_it);                             // :1681 (part_idx = iter, sic)
        }
    } else {
        local_growing(d[0], loc_it, true, loc_it);                                 // :1703
    }
    memcpy(out_flow, d[0].out.data(), sizeof(float) * 2 * n);
    memcpy(ene_val, d[0].ene.data(), sizeof(float) * n);
    if (stats) {
        for (int k = 0; k < 2; k++) { stats[3 * k] = d[k].pops; stats[3 * k + 1] = d[k].solves; stats[3 * k + 2] = d[k].fills; }
        for (std::vector<Partition> *P : { &PA, &PB, &PC })
            for (Partition &p : *P)
                for (int k = 0; k < 2; k++) { stats[3 * k] += p.d[k].pops; stats[3 * k + 1] += p.d[k].solves; stats[3 * k + 2] += p.d[k].fills; }
    }
}

/* match_growing_variational for any functional 0..7 from the RAW frames (planar, pd channels, 0..255): prepare_stuff's
 * per-method preparation (src/energy_model.cpp:276-686) and then the same growing as above.  -m 2 builds both frames from
 * i0 (:318-319); the non-local functionals take their weights from the Lab image of i0 for both directions (:322-329 etc.).
 * Image partitions are only defined for the TV-l2 functionals (upstream README.md:359). */
extern "C" void orc_local_faldoi_raw(const float *i0, const float *i1, int pd, const float *go, const float *ba, const float *sal0,
                                     const float *sal1, int w, int h, int wr, int loc_it, int max_iter_patch, float fb_thresh,
                                     int method, float *out_flow, float *ene_val, long *stats)
{
    const size_t n = (size_t)w * h;
    std::vector<float> a(n), b(n);
    orc_preprocess_local(i0, method == 2 ? i0 : i1, w, h, pd, a.data(), b.data());
    if (method < 2) {
        orc_local_faldoi(a.data(), b.data(), go, ba, sal0, sal1, w, h, wr, loc_it, max_iter_patch, fb_thresh, method, out_flow,
                         ene_val, stats);
        return;
    }
    const bool nl = method == 2 || method == 3 || method == 6 || method == 7;
    std::vector<float> lab, wp, wt;
    if (nl) {
        lab.resize(n * pd);
        wp.resize(n * 24);
        wt.resize(n);
        orc_rgb_to_lab_local(i0, (int)n, lab.data());
        orc_nltv_table_local(lab.data(), pd, w, h, wp.data(), wt.data());
    }
    Direction d[2];
    init_direction(d[0], a.data(), b.data(), sal0, w, h, wr, max_iter_patch, method);
    init_direction(d[1], b.data(), a.data(), sal1, w, h, wr, max_iter_patch, method);
    for (int k = 0; k < 2; k++)
        if (nl) { d[k].wp = wp.data(); d[k].duals.assign(n * 48, 0.0f); }
#pragma omp parallel for num_threads(2)
    for (int k = 0; k < 2; k++) insert_initial_seeds(d[k], k == 0 ? go : ba);
    for (int it = 0; it < loc_it; it++) {
#pragma omp parallel for num_threads(2)
        for (int k = 0; k < 2; k++) local_growing(d[k], it, k == 0, -1);
        prune_and_reseed(d, w, h, fb_thresh);
    }
    local_growing(d[0], loc_it, true, loc_it);
    memcpy(out_flow, d[0].out.data(), sizeof(float) * 2 * n);
    memcpy(ene_val, d[0].ene.data(), sizeof(float) * n);
    if (stats)
        for (int k = 0; k < 2; k++) { stats[3 * k] = d[k].pops; stats[3 * k + 1] = d[k].solves; stats[3 * k + 2] = d[k].fills; }
}

extern "C" void orc_local_faldoi(const float *i0n, const float *i1n, const float *go, const float *ba, const float *sal0,
                                 const float *sal1, int w, int h, int wr, int loc_it, int max_iter_patch, float fb_thresh,
                                 int method, float *out_flow, float *ene_val, long *stats)
{
    orc_local_faldoi_split(i0n, i1n, go, ba, sal0, sal1, w, h, wr, loc_it, max_iter_patch, fb_thresh, method, 0, 1, 1, 0, 0, out_flow,
                           ene_val, stats);
}

/* prepare_stuff's image part for TV-l2 coupled, src/energy_model.cpp:660-684 */
extern "C" void orc_preprocess_local(const float *i0, const float *i1, int w, int h, int pd, float *i0n, float *i1n)
{
    const size_t n = (size_t)w * h;
    if (pd != 1) {
        orc_rgb2gray(i0, w, h, i0n);
        orc_rgb2gray(i1, w, h, i1n);
    } else {
        memcpy(i0n, i0, sizeof(float) * n);
        memcpy(i1n, i1, sizeof(float) * n);
    }
    orc_image_normalization_2(i0n, i1n, i0n, i1n, (int)n);
    orc_gaussian(i0n, w, h, 0.9f);                                     // PRESMOOTHING_SIGMA
    orc_gaussian(i1n, w, h, 0.9f);
}
