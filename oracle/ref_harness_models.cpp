// ref_harness_models.cpp -- extern "C" doorways into the reference's OWN energy-model, table and partition
// objects (second part of oracle/_ref/libfaldoi_ref.so; see ref_harness.cpp for the ground rules).
//
// TEST INFRASTRUCTURE ONLY.  No algorithm lives here: every function forwards to a reference function compiled
// where its source lies under /root/reference/src (oracle/Makefile): energy_model.cpp (initialize_auxiliar_stuff,
// prepare_stuff, of_estimation -- the operator surface of the local step, all functionals), aux_energy_model.cpp
// (NLTV dual tables, CSAD neighbour tables), nltv_model.cpp / tvcsad_model.cpp / nltvcsad_model.cpp and their
// weighted forms, aux_partitions.cpp (init_subimage_partitions, image_to_partitions).  None of these files includes
// boost, so they build here without any stand-in header.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <queue>
#include <vector>

#include "energy_structures.h"           // reference: src/energy_structures.h
#include "energy_model.h"                // reference: src/energy_model.h
#include "aux_energy_model.h"            // reference: src/aux_energy_model.h
#include "aux_partitions.h"              // reference: src/aux_partitions.h

void rgb_to_lab(const float *in, int size, float *out);          // src/energy_model.cpp:60 (no header declares it)
// src/mask.c:360, the C-linkage copy: the one energy_model.cpp binds to, because it includes mask.h inside extern "C"
// before utils.h (src/energy_model.cpp:18-23), so prepare_stuff's weights come from here (exp in double), not from
// the C++ copy in src/utils.cpp:633 (exp of a float is expf there) -- the two differ in the last bit of some weights
extern "C" void gaussian1Dweight(float *I, int r);

namespace {

// Every float plane of the structure a functional uses, so that "what `new float[]` holds before the first call"
// is defined: zero, which is what fresh pages from the allocator hold for frame-sized arrays.
std::vector<float *> planes_of(SpecificOFStuff &s, int method)
{
    switch (method) {
    case M_NLTVL1: { NonLocalTVL1Stuff &t = s.nltvl1;
        return { t.v1, t.v2, t.rho_c, t.grad, t.u1_, t.u2_, t.u1_tmp, t.u2_tmp, t.I1x, t.I1y, t.I1w, t.I1wx, t.I1wy, t.div_p, t.div_q }; }
    case M_NLTVL1_W: { NonLocalTVL1Stuff_W &t = s.nltvl1w;
        return { t.v1, t.v2, t.rho_c, t.grad, t.u1_, t.u2_, t.u1_tmp, t.u2_tmp, t.I1x, t.I1y, t.I1w, t.I1wx, t.I1wy, t.div_p, t.div_q }; }
    case M_TVCSAD: { TvCsadStuff &t = s.tvcsad;
        return { t.xi11, t.xi12, t.xi21, t.xi22, t.u1x, t.u1y, t.u2x, t.u2y, t.v1, t.v2, t.rho_c, t.grad, t.u1_, t.u2_, t.u1_tmp,
                 t.u2_tmp, t.I1x, t.I1y, t.I1w, t.I1wx, t.I1wy, t.div_xi1, t.div_xi2 }; }
    case M_TVCSAD_W: { TvCsadStuff_W &t = s.tvcsadw;
        return { t.xi11, t.xi12, t.xi21, t.xi22, t.u1x, t.u1y, t.u2x, t.u2y, t.v1, t.v2, t.rho_c, t.grad, t.u1_, t.u2_, t.u1_tmp,
                 t.u2_tmp, t.I1x, t.I1y, t.I1w, t.I1wx, t.I1wy, t.div_xi1, t.div_xi2 }; }
    case M_NLTVCSAD: { NonLocalTvCsadStuff &t = s.nltvcsad;
        return { t.v1, t.v2, t.rho_c, t.grad, t.u1_, t.u2_, t.u1_tmp, t.u2_tmp, t.I1x, t.I1y, t.I1w, t.I1wx, t.I1wy, t.div_p, t.div_q }; }
    case M_NLTVCSAD_W: { NonLocalTvCsadStuff_W &t = s.nltvcsadw;
        return { t.v1, t.v2, t.rho_c, t.grad, t.u1_, t.u2_, t.u1_tmp, t.u2_tmp, t.I1x, t.I1y, t.I1w, t.I1wx, t.I1wy, t.div_p, t.div_q }; }
    case M_TVL1_W: { Tvl2CoupledOFStuff_W &t = s.tvl2w;
        return { t.xi11, t.xi12, t.xi21, t.xi22, t.u1x, t.u1y, t.u2x, t.u2y, t.v1, t.v2, t.rho_c, t.grad, t.u1_, t.u2_, t.u1Aux,
                 t.u2Aux, t.I1x, t.I1y, t.I1w, t.I1wx, t.I1wy, t.div_xi1, t.div_xi2, t.u_N }; }
    default: { Tvl2CoupledOFStuff &t = s.tvl2;
        return { t.xi11, t.xi12, t.xi21, t.xi22, t.u1x, t.u1y, t.u2x, t.u2y, t.v1, t.v2, t.rho_c, t.grad, t.u1_, t.u2_, t.u1Aux,
                 t.u2Aux, t.I1x, t.I1y, t.I1w, t.I1wx, t.I1wy, t.div_xi1, t.div_xi2, t.u_N }; }
    }
}

DualVariables *duals_of(SpecificOFStuff &s, int method, int which)
{
    switch (method) {
    case M_NLTVL1: return which ? s.nltvl1.q : s.nltvl1.p;
    case M_NLTVL1_W: return which ? s.nltvl1w.q : s.nltvl1w.p;
    case M_NLTVCSAD: return which ? s.nltvcsad.q : s.nltvcsad.p;
    case M_NLTVCSAD_W: return which ? s.nltvcsadw.q : s.nltvcsadw.p;
    default: return nullptr;
    }
}

struct RefOp {
    int method, w, h, wr;
    Parameters params;
    OpticalFlowData of[2];
    SpecificOFStuff stuff[2];
    std::vector<float> sal[2];
    float *i0n = nullptr, *i1n = nullptr;
};

Parameters make_params(int method, int w, int h, int wr, int max_iter_patch)
{
    Parameters p;
    memset(&p, 0, sizeof p);
    p.lambda = PAR_DEFAULT_LAMBDA; p.theta = PAR_DEFAULT_THETA; p.tau = PAR_DEFAULT_TAU; p.tol_OF = PAR_DEFAULT_TOL_D;
    p.warps = PAR_DEFAULT_NWARPS_LOCAL; p.w = w; p.h = h; p.pd = 1; p.w_radio = wr; p.val_method = method;
    p.step_algorithm = LOCAL_STEP; p.iterations_of = LOCAL_ITER; p.max_iter_patch = max_iter_patch;
    p.h_parts = HOR_PARTS; p.v_parts = VER_PARTS; p.epsilon = FB_TOL;
    return p;
}

void zero_fill(OpticalFlowData &of, SpecificOFStuff &st, int method, int w, int h)
{
    const size_t n = (size_t)w * h;
    for (float *p : planes_of(st, method)) memset(p, 0, sizeof(float) * n);
    memset(of.u1, 0, sizeof(float) * 2 * n);
    memset(of.u1_ba, 0, sizeof(float) * 2 * n);
    memset(of.u1_filter, 0, sizeof(float) * 2 * n);
    memset(of.chi, 0, sizeof(float) * n);
    memset(of.fixed_points, 0, sizeof(int) * n);
    memset(of.trust_points, 0, sizeof(int) * n);
}

}  // namespace

extern "C" {

// ---- the operator surface of the local step, any functional 0..7 (src/energy_model.cpp:161-802) ----
// create = init_Optical_Flow_Data (:177) + initialize_auxiliar_stuff (:202), forward (0) and backward (1)
void *ref_op_create(int method, int w, int h, int wr, int max_iter_patch)
{
    RefOp *r = new RefOp();
    r->method = method; r->w = w; r->h = h; r->wr = wr;
    r->params = make_params(method, w, h, wr, max_iter_patch);
    for (int k = 0; k < 2; k++) {
        r->sal[k].assign((size_t)w * h, 1.0f);
        r->of[k] = init_Optical_Flow_Data(r->sal[k].data(), r->params, w, h);
        initialize_auxiliar_stuff(r->stuff[k], r->of[k], w, h);
        zero_fill(r->of[k], r->stuff[k], method, w, h);
    }
    return r;
}

// prepare_stuff (:276) as match_growing_variational calls it (src/local_faldoi.cpp:1097-1100): i_1 / i2 are only read
// by the occlusion model.  i0n / i1n receive the planes the growings are then handed.
void ref_op_prepare(void *h, const float *i0, const float *i1, int pd, float *i0n, float *i1n)
{
    RefOp *r = (RefOp *)h;
    r->params.pd = pd;
    r->of[0].params.pd = r->of[1].params.pd = pd;
    float *o0 = nullptr, *o1 = nullptr, *o_1 = nullptr, *o2 = nullptr;
    prepare_stuff(&r->stuff[0], &r->of[0], &r->stuff[1], &r->of[1], i0, i1, i0, i1, pd, &o0, &o1, &o_1, &o2, r->w, r->h);
    r->i0n = o0;
    r->i1n = o1;
    const size_t n = (size_t)r->w * r->h;
    if (i0n) memcpy(i0n, o0, sizeof(float) * n);
    if (i1n) memcpy(i1n, o1, sizeof(float) * n);
}

void ref_op_set_flow(void *h, int dir, const float *u)
{ RefOp *r = (RefOp *)h; memcpy(r->of[dir].u1, u, sizeof(float) * 2 * (size_t)r->w * r->h); }

void ref_op_get_flow(void *h, int dir, float *u)
{ RefOp *r = (RefOp *)h; memcpy(u, r->of[dir].u1, sizeof(float) * 2 * (size_t)r->w * r->h); }

// One of_estimation call (:691) on the window {ii, ij, ei, ej} centred at (i, j), the way add_neighbors makes it
// (src/local_faldoi.cpp:679-712): wr_call = the radius in force (1 while seeding), which the weighted functionals
// read from params.w_radio; iiw / ijw = what get_index_weight (:571-597, a static function of local_faldoi.cpp)
// stores for them -- computed by the caller.
float ref_op_estimate(void *h, int dir, int i, int j, int ii, int ij, int ei, int ej, int wr_call, int iiw, int ijw)
{
    RefOp *r = (RefOp *)h;
    PatchIndexes idx;
    idx.i = i; idx.j = j; idx.ii = ii; idx.ij = ij; idx.ei = ei; idx.ej = ej;
    r->of[dir].params.w_radio = wr_call;
    SpecificOFStuff &s = r->stuff[dir];
    switch (r->method) {
    case M_TVL1_W: s.tvl2w.iiw = iiw; s.tvl2w.ijw = ijw; break;
    case M_NLTVCSAD_W: s.nltvcsadw.iiw = iiw; s.nltvcsadw.ijw = ijw; break;
    case M_NLTVL1_W: s.nltvl1w.iiw = iiw; s.nltvl1w.ijw = ijw; break;
    case M_TVCSAD_W: s.tvcsadw.iiw = iiw; s.tvcsadw.ijw = ijw; break;
    default: break;
    }
    float ener = 0.f;
    const float *a = dir == 0 ? r->i0n : r->i1n, *b = dir == 0 ? r->i1n : r->i0n;
    of_estimation(&s, &r->of[dir], &ener, a, b, a, idx, r->w, r->h);
    return ener;
}

// state readers: a named float plane of the functional's structure, or the dual tables as planes
int ref_op_plane(void *h, int dir, const char *name, float *out)
{
    RefOp *r = (RefOp *)h;
    SpecificOFStuff &s = r->stuff[dir];
    const size_t n = (size_t)r->w * r->h;
    const float *p = nullptr;
    const bool tv = r->method == M_TVL1 || r->method == M_TVL1_W || r->method == M_TVCSAD || r->method == M_TVCSAD_W;
#define PICK(field)                                                                                                     \
    (r->method == M_TVL1 ? s.tvl2.field : r->method == M_TVL1_W ? s.tvl2w.field : r->method == M_TVCSAD ? s.tvcsad.field \
                                                                                                         : s.tvcsadw.field)
#define PICKNL(field)                                                                                                   \
    (r->method == M_NLTVL1 ? s.nltvl1.field : r->method == M_NLTVL1_W ? s.nltvl1w.field                                 \
     : r->method == M_NLTVCSAD ? s.nltvcsad.field : s.nltvcsadw.field)
    if (tv) {
        if (!strcmp(name, "div_xi1")) p = PICK(div_xi1);
        else if (!strcmp(name, "div_xi2")) p = PICK(div_xi2);
        else if (!strcmp(name, "xi11")) p = PICK(xi11);
        else if (!strcmp(name, "xi12")) p = PICK(xi12);
        else if (!strcmp(name, "xi21")) p = PICK(xi21);
        else if (!strcmp(name, "xi22")) p = PICK(xi22);
        else if (!strcmp(name, "I1x")) p = PICK(I1x);
        else if (!strcmp(name, "I1y")) p = PICK(I1y);
        else if (!strcmp(name, "v1")) p = PICK(v1);
        else if (!strcmp(name, "v2")) p = PICK(v2);
    } else {
        if (!strcmp(name, "div_p")) p = PICKNL(div_p);
        else if (!strcmp(name, "div_q")) p = PICKNL(div_q);
        else if (!strcmp(name, "I1x")) p = PICKNL(I1x);
        else if (!strcmp(name, "I1y")) p = PICKNL(I1y);
        else if (!strcmp(name, "v1")) p = PICKNL(v1);
        else if (!strcmp(name, "v2")) p = PICKNL(v2);
    }
#undef PICK
#undef PICKNL
    if (!p) return -1;
    memcpy(out, p, sizeof(float) * n);
    return 0;
}

// NLTV dual tables of a direction: sc / wp as [24][n] planes, wt as [n]; which = 0 (p) or 1 (q).  Returns -1 if the
// functional has none.
int ref_op_duals(void *h, int dir, int which, float *sc, float *wp, float *wt)
{
    RefOp *r = (RefOp *)h;
    DualVariables *d = duals_of(r->stuff[dir], r->method, which);
    if (!d) return -1;
    const size_t n = (size_t)r->w * r->h;
    const int nd = NL_DUAL_VAR;
    for (size_t i = 0; i < n; i++) {
        for (int j = 0; j < nd; j++) {
            if (sc) sc[(size_t)j * n + i] = d[i].sc[j];
            if (wp) wp[(size_t)j * n + i] = d[i].wp[j];
        }
        if (wt) wt[i] = d[i].wt;
    }
    return 0;
}

// the Gaussian window weights prepare_stuff left in the structure (weighted functionals): 2 * wr + 1 floats
int ref_op_weight(void *h, int dir, float *out)
{
    RefOp *r = (RefOp *)h;
    SpecificOFStuff &s = r->stuff[dir];
    const float *wgt = r->method == M_TVL1_W ? s.tvl2w.weight : r->method == M_NLTVCSAD_W ? s.nltvcsadw.weight
                     : r->method == M_NLTVL1_W ? s.nltvl1w.weight : r->method == M_TVCSAD_W ? s.tvcsadw.weight : nullptr;
    if (!wgt) return -1;
    memcpy(out, wgt, sizeof(float) * (2 * r->wr + 1));
    return 0;
}

void ref_op_destroy(void *h)
{
    RefOp *r = (RefOp *)h;
    for (int k = 0; k < 2; k++) free_auxiliar_stuff(&r->stuff[k], &r->of[k]);
    delete r;
}

// ---- tables (src/aux_energy_model.cpp, src/energy_model.cpp) ----
void ref_rgb_to_lab(const float *in, int size, float *out) { rgb_to_lab(in, size, out); }      // src/energy_model.cpp:60

void ref_gaussian1d_weight_c(float *I, int r) { gaussian1Dweight(I, r); }                    // src/mask.c:360

// nltv_ini_dual_variables (src/aux_energy_model.cpp:112): per slot the value, weight, neighbour and mirror slot
void ref_nltv_table(const float *lab, int pd, int w, int h, float *sc, float *wp, int *api, int *apj, int *rp, float *wt)
{
    const size_t n = (size_t)w * h;
    const int nd = NL_DUAL_VAR, radius = NL_BETA;
    std::vector<DualVariables> p(n), q(n);
    nltv_ini_dual_variables(const_cast<float *>(lab), pd, w, h, nd, radius, p.data(), q.data());
    for (size_t i = 0; i < n; i++) {
        for (int j = 0; j < nd; j++) {
            sc[(size_t)j * n + i] = p[i].sc[j];
            wp[(size_t)j * n + i] = p[i].wp[j];
            api[(size_t)j * n + i] = p[i].api[j];
            apj[(size_t)j * n + i] = p[i].apj[j];
            rp[(size_t)j * n + i] = p[i].rp[j];
        }
        wt[i] = p[i].wt;
    }
}

// csad_ini_pos_nei (src/aux_energy_model.cpp:216)
void ref_csad_table(int w, int h, int *api, int *apj, int *count)
{
    const size_t n = (size_t)w * h;
    const int ndt = DT_NEI, rdt = DT_R;
    std::vector<PosNei> t(n);
    csad_ini_pos_nei(w, h, ndt, rdt, t.data());
    for (size_t i = 0; i < n; i++) {
        for (int j = 0; j < ndt; j++) {
            api[(size_t)j * n + i] = t[i].api[j];
            apj[(size_t)j * n + i] = t[i].apj[j];
        }
        count[i] = t[i].n;
    }
}

// ---- image partitions (src/aux_partitions.cpp), TV-l2 coupled ----
// An image-level state (the arrays match_growing_variational owns, src/local_faldoi.cpp:1082-1113) plus the
// reference's partition vector for one grid.
struct RefParts {
    int w, h, cols, rows;
    Parameters params;
    OpticalFlowData of[2];
    SpecificOFStuff stuff[2];
    std::vector<float> sal[2], oft[2], ene[2], occ[2];
    pq_cand queue[2];
    std::vector<PartitionData *> parts;
};

void *ref_parts_create(const float *i0, const float *i1, int pd, const float *i0n, const float *i1n, int w, int h,
                       int cols, int rows, int wr)
{
    RefParts *r = new RefParts();
    r->w = w; r->h = h; r->cols = cols; r->rows = rows;
    r->params = make_params(M_TVL1, w, h, wr, MAX_ITERATIONS_LOCAL);
    r->params.pd = pd;
    const size_t n = (size_t)w * h;
    for (int k = 0; k < 2; k++) {
        r->sal[k].assign(n, 1.0f);
        r->oft[k].assign(2 * n, NAN);
        r->ene[k].assign(n, INFINITY);
        r->occ[k].assign(n, 0.0f);
        r->of[k] = init_Optical_Flow_Data(r->sal[k].data(), r->params, w, h);
        initialize_auxiliar_stuff(r->stuff[k], r->of[k], w, h);
        zero_fill(r->of[k], r->stuff[k], M_TVL1, w, h);
    }
    BilateralFilterData *bf0 = init_weights_bilateral(const_cast<float *>(i0), w, h);      // :1103-1104 of local_faldoi.cpp
    BilateralFilterData *bf1 = init_weights_bilateral(const_cast<float *>(i1), w, h);
    init_subimage_partitions(i0, i1, i0, i1, i0n, i1n, nullptr, nullptr, bf0, bf1, r->sal[0].data(), r->sal[1].data(), w, h, cols,
                             rows, &r->parts, r->params);
    for (PartitionData *p : r->parts) {                        // queue_*_size is uninitialised in the reference
        p->queue_Go_size = 0;                                  // (new PartitionData without an initialiser); a counter that
        p->queue_Ba_size = 0;                                  // starts anywhere else makes anyEmptyQueues meaningless
    }
    return r;
}

int ref_parts_count(void *h) { return (int)((RefParts *)h)->parts.size(); }

void ref_parts_geometry(void *h, int p, int *geom /* width, height, off_x, off_y */)
{
    PartitionData *d = ((RefParts *)h)->parts[p];
    geom[0] = d->width; geom[1] = d->height; geom[2] = d->off_x; geom[3] = d->off_y;
}

// plane ids shared with the oracle's hooks: 0 u1, 1 u2, 2 xi11, 3 xi12, 4 xi21, 5 xi22, 6 div_xi1, 7 div_xi2, 8 I1x, 9 I1y,
// 10 ene, 11 oft u, 12 oft v, 13 fixed (as float), 14 trust (as float)
static float *img_plane(RefParts *r, int dir, int id, int **ip)
{
    Tvl2CoupledOFStuff &t = r->stuff[dir].tvl2;
    *ip = nullptr;
    const size_t n = (size_t)r->w * r->h;
    switch (id) {
    case 0: return r->of[dir].u1; case 1: return r->of[dir].u2;
    case 2: return t.xi11; case 3: return t.xi12; case 4: return t.xi21; case 5: return t.xi22;
    case 6: return t.div_xi1; case 7: return t.div_xi2; case 8: return t.I1x; case 9: return t.I1y;
    case 10: return r->ene[dir].data(); case 11: return r->oft[dir].data(); case 12: return r->oft[dir].data() + n;
    case 13: *ip = r->of[dir].fixed_points; return nullptr; case 14: *ip = r->of[dir].trust_points; return nullptr;
    default: return nullptr;
    }
}
static float *part_plane(PartitionData *d, int dir, int id, int **ip)
{
    Tvl2CoupledOFStuff &t = (dir ? d->stuffBa : d->stuffGo).tvl2;
    OpticalFlowData &of = dir ? d->ofBa : d->ofGo;
    *ip = nullptr;
    const size_t n = (size_t)d->width * d->height;
    switch (id) {
    case 0: return of.u1; case 1: return of.u2;
    case 2: return t.xi11; case 3: return t.xi12; case 4: return t.xi21; case 5: return t.xi22;
    case 6: return t.div_xi1; case 7: return t.div_xi2; case 8: return t.I1x; case 9: return t.I1y;
    case 10: return dir ? d->ene_Ba : d->ene_Go; case 11: return dir ? d->oft1 : d->oft0; case 12: return (dir ? d->oft1 : d->oft0) + n;
    case 13: *ip = of.fixed_points; return nullptr; case 14: *ip = of.trust_points; return nullptr;
    default: return nullptr;
    }
}

void ref_parts_set_image_plane(void *h, int dir, int id, const float *v)
{
    RefParts *r = (RefParts *)h;
    int *ip;
    float *p = img_plane(r, dir, id, &ip);
    const size_t n = (size_t)r->w * r->h;
    for (size_t i = 0; i < n; i++) { if (p) p[i] = v[i]; else ip[i] = (int)v[i]; }
}
void ref_parts_get_image_plane(void *h, int dir, int id, float *v)
{
    RefParts *r = (RefParts *)h;
    int *ip;
    float *p = img_plane(r, dir, id, &ip);
    const size_t n = (size_t)r->w * r->h;
    for (size_t i = 0; i < n; i++) v[i] = p ? p[i] : (float)ip[i];
}
void ref_parts_set_part_plane(void *h, int part, int dir, int id, const float *v)
{
    PartitionData *d = ((RefParts *)h)->parts[part];
    int *ip;
    float *p = part_plane(d, dir, id, &ip);
    const size_t n = (size_t)d->width * d->height;
    for (size_t i = 0; i < n; i++) { if (p) p[i] = v[i]; else ip[i] = (int)v[i]; }
}
void ref_parts_get_part_plane(void *h, int part, int dir, int id, float *v)
{
    PartitionData *d = ((RefParts *)h)->parts[part];
    int *ip;
    float *p = part_plane(d, dir, id, &ip);
    const size_t n = (size_t)d->width * d->height;
    for (size_t i = 0; i < n; i++) v[i] = p ? p[i] : (float)ip[i];
}
// partition frames as init_subimage_partitions left them: 0 = i0n crop, 1 = i1n crop
void ref_parts_get_part_frame(void *h, int part, int which, float *v)
{
    PartitionData *d = ((RefParts *)h)->parts[part];
    memcpy(v, which ? d->i1n : d->i0n, sizeof(float) * (size_t)d->width * d->height);
}

void ref_parts_push(void *h, int dir, int i, int j, float u, float v, float sim)
{
    SparseOF e;
    e.i = i; e.j = j; e.u = u; e.v = v; e.sim_node = sim; e.occluded = 0.0f;
    ((RefParts *)h)->queue[dir].push(e);
}
int ref_parts_image_queue_size(void *h, int dir) { return (int)((RefParts *)h)->queue[dir].size(); }

// image_to_partitions (src/aux_partitions.cpp:2056) in either direction
void ref_parts_transfer(void *h, int img_to_part)
{
    RefParts *r = (RefParts *)h;
    image_to_partitions(r->oft[0].data(), r->oft[1].data(), r->ene[0].data(), r->ene[1].data(), r->occ[0].data(),
                        r->occ[1].data(), &r->of[0], &r->of[1], &r->stuff[0], &r->stuff[1], r->queue[0], r->queue[1],
                        (int)r->parts.size(), r->w, r->h, &r->parts, img_to_part != 0);
}
int ref_parts_any_empty(void *h)
{
    RefParts *r = (RefParts *)h;
    return anyEmptyQueues(&r->parts, (int)r->parts.size()) ? 1 : 0;
}
// pops the partition queue in its own order; returns the number of candidates written (5 floats each: i, j, u, v, sim)
int ref_parts_drain_queue(void *h, int part, int dir, float *out, int cap)
{
    PartitionData *d = ((RefParts *)h)->parts[part];
    pq_cand &q = dir ? d->queue_Ba : d->queue_Go;
    int k = 0;
    while (!q.empty() && k < cap) {
        const SparseOF e = q.top();
        q.pop();
        out[5 * k] = (float)e.i; out[5 * k + 1] = (float)e.j; out[5 * k + 2] = e.u; out[5 * k + 3] = e.v; out[5 * k + 4] = e.sim_node;
        k++;
    }
    return k;
}

}  // extern "C"
