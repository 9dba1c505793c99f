"""ctypes bindings for the test-only CPU checker (oracle/liboracle.so, oracle/_ref/libfaldoi_ref.so).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_I = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and _ref when the reference sources are present)."""
    so = os.path.join(_HERE, "liboracle.so")
    import fcntl
    srcs = [os.path.join(_HERE, f) for f in ("faldoi_oracle.c", "faldoi_oracle_nltv.c", "faldoi_oracle_local.c", "faldoi_oracle_local_nl.c",
                                             "faldoi_oracle_growing.cpp", "faldoi_oracle.h")]
    fd = os.open(os.path.join(_HERE, ".build.lock"), os.O_CREAT | os.O_RDWR, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)              # several ranks / xdist workers may get here together
        if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
        if os.path.isdir("/root/reference/src"):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)
    return so


_lib = None
_ref = None


def cpu_share():
    """Host threads this job may use: the affinity mask, capped at the GPU box's 16-per-GPU share."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(16, avail))


def _cap_threads():
    """The GPU box exposes many more hardware threads than the CPU share a job may use (16 per
    GPU); an OpenMP team sized by the former spends its time spinning.  Cap before libgomp loads."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, avail))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _cap_threads()
        L = C.CDLL(so)
        L.orc_centered_gradient.argtypes = [_F, _F, _F, C.c_int, C.c_int]
        L.orc_forward_gradient.argtypes = [_F, _F, _F, C.c_int, C.c_int]
        L.orc_divergence.argtypes = [_F, _F, _F, C.c_int, C.c_int]
        L.orc_bicubic_at.argtypes = [_F, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_bicubic_at.restype = C.c_float
        L.orc_bicubic_warp.argtypes = [_F, _F, _F, _F, C.c_int, C.c_int, C.c_int]
        L.orc_rgb2gray.argtypes = [_F, C.c_int, C.c_int, _F]
        L.orc_image_normalization_3.argtypes = [_F, _F, _F, _F, _F, _F, C.c_int]
        L.orc_image_normalization_2.argtypes = [_F, _F, _F, _F, C.c_int]
        L.orc_gaussian.argtypes = [_F, C.c_int, C.c_int, C.c_float]
        L.orc_gaussian_taps.argtypes = [C.c_float, _F, C.c_int]
        L.orc_gaussian_c.argtypes = [_F, C.c_int, C.c_int, C.c_float]
        L.orc_zoom_size.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_float]
        L.orc_zoom_out.argtypes = [_F, _F, C.c_int, C.c_int, C.c_float]
        L.orc_zoom_in.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_preprocess_global.argtypes = [_F, _F, _F, C.c_int, C.c_int, C.c_int, _F, _F, _F]
        L.orc_tvl1_global.argtypes = [_F, _F, _F, _F, _F, _F, _F, _F,
                                      C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _I, _F]
        L.orc_tvl1_global.restype = C.c_long
        L.orc_tvl1_iteration.argtypes = [_F] * 12 + [C.c_float] * 3 + [C.c_int] * 2
        L.orc_tvl1_iteration.restype = C.c_float
        L.orc_tvl1_warp_constants.argtypes = [_F] * 10 + [C.c_int] * 2
        L.orc_image_to_lab.argtypes = [_F, C.c_int, _F]
        L.orc_nltv_weights.argtypes = [_F, C.c_int, C.c_int, C.c_int, _F, _F]
        L.orc_nltvcsad_global.argtypes = [_F, _F, _F, C.c_int, C.c_float, C.c_float, C.c_float,
                                          C.c_int, C.c_int, C.c_int, C.c_int, _F, _F]
        L.orc_nltvl1_global.argtypes = L.orc_nltvcsad_global.argtypes
        L.orc_tvcsad_global.argtypes = [_F, _F] + [C.c_float] * 4 + [C.c_int] * 4 + [_F, _F, _F, _I, _F]
        L.orc_tvcsad_global.restype = C.c_long
        L.orc_elap_recursive.argtypes = [_F, _F, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]
        L.orc_interpolate_poisson.argtypes = [_F, _F, _I, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_local_tvl1_estimate.argtypes = [_F] * 7 + [C.c_int] * 7 + [C.POINTER(C.c_int)]
        L.orc_local_tvl1_estimate.restype = C.c_float
        _FN = C.c_void_p                         # float plane or NULL
        L.orc_preprocess_local.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, _F, _F]
        L.orc_gaussian1d_weight.argtypes = [_F, C.c_int]
        L.orc_local_tvl1w_estimate.argtypes = [_F] * 7 + [C.c_int] * 7 + [C.POINTER(C.c_int), _F, C.c_int, C.c_int, C.c_int]
        L.orc_local_tvl1w_estimate.restype = C.c_float
        L.orc_local_faldoi.argtypes = [_F, _F, _F, _F, _FN, _FN] + [C.c_int] * 5 + [C.c_float, C.c_int, _F, _F,
                                       np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")]
        L.orc_partial_enable.argtypes = [C.c_int]
        L.orc_partial_enable.restype = None
        L.orc_partial_count.restype = C.c_int
        L.orc_partial_get.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_partial_get.restype = None
        L.orc_local_faldoi_split.argtypes = [_F, _F, _F, _F, _FN, _FN] + [C.c_int] * 5 + [C.c_float] + [C.c_int] * 6 + [
            _F, _F, np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")]
        L.orc_local_faldoi_raw.argtypes = [_F, _F, C.c_int, _F, _F, _FN, _FN] + [C.c_int] * 5 + [C.c_float, C.c_int, _F, _F,
                                           np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")]
        L.orc_fb_consistency_check.argtypes = [_F, _F, _I, C.c_int, C.c_int, C.c_float]
        L.orc_fb_consistency_check.restype = C.c_double
        L.orc_queue_new.restype = C.c_void_p
        L.orc_queue_free.argtypes = [C.c_void_p]
        L.orc_queue_push.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
        L.orc_queue_pop.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 2 + [C.POINTER(C.c_float)] * 3
        L.orc_rgb_to_lab_local.argtypes = [_F, C.c_int, _F]
        L.orc_nltv_table_local.argtypes = [_F, C.c_int, C.c_int, C.c_int, _F, _F]
        L.orc_local_nl_estimate.restype = C.c_float
        L.orc_local_nl_estimate.argtypes = [C.c_int, _F, _F, _F, _F, _F, _F, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [
            _F, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_parts_create.restype = C.c_void_p
        L.orc_parts_create.argtypes = [_F, _F] + [C.c_int] * 5
        L.orc_parts_destroy.argtypes = [C.c_void_p]
        _bind_parts(L, "orc_parts_")
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads(cpu_share())      # libgomp may already be loaded (torch) with its own default
        _lib = L
    return _lib


def ref():
    """The reference's own primitives (None when oracle/_ref was not built / did not travel)."""
    global _ref
    if _ref is None:
        so = os.path.join(_HERE, "_ref", "libfaldoi_ref.so")
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        R.ref_centered_gradient.argtypes = [_F, _F, _F, C.c_int, C.c_int]
        R.ref_forward_gradient.argtypes = [_F, _F, _F, C.c_int, C.c_int]
        R.ref_divergence.argtypes = [_F, _F, _F, C.c_int, C.c_int]
        R.ref_gaussian.argtypes = [_F, C.c_int, C.c_int, C.c_float]
        R.ref_image_normalization.argtypes = [_F, _F, _F, _F, C.c_int]
        R.ref_image_normalization_3.argtypes = [_F, _F, _F, _F, _F, _F, C.c_int]
        R.ref_bicubic_at.argtypes = [_F, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        R.ref_bicubic_at.restype = C.c_float
        R.ref_bicubic_warp.argtypes = [_F, _F, _F, _F, C.c_int, C.c_int, C.c_int]
        if hasattr(R, "ref_local_tvl1_create"):
            R.ref_local_tvl1_create.argtypes = [C.c_int, C.c_int]
            R.ref_local_tvl1_create.restype = C.c_void_p
            R.ref_local_tvl1_destroy.argtypes = [C.c_void_p]
            R.ref_local_tvl1_estimate.argtypes = [C.c_void_p, _F, _F, _F, _F, _F, _F] + [C.c_int] * 5
            R.ref_local_tvl1_estimate.restype = C.c_float
            R.ref_local_tvl1_div.argtypes = [C.c_void_p, C.c_int]
            R.ref_local_tvl1_div.restype = C.POINTER(C.c_float)
        if hasattr(R, "ref_elap_recursive_separable"):
            if hasattr(R, "ref_local_tvl1w_create"):
                R.ref_local_tvl1w_create.restype = C.c_void_p
                R.ref_local_tvl1w_create.argtypes = [C.c_int, C.c_int, C.c_int]
                R.ref_local_tvl1w_destroy.argtypes = [C.c_void_p]
                R.ref_local_tvl1w_estimate.restype = C.c_float
                R.ref_local_tvl1w_estimate.argtypes = [C.c_void_p, _F, _F, _F, _F, _F, _F] + [C.c_int] * 8
                R.ref_gaussian1d_weight.argtypes = [_F, C.c_int]
                if hasattr(R, "ref_local_tvl1w_set_weight"):
                    R.ref_local_tvl1w_set_weight.argtypes = [C.c_void_p, _F]
            R.ref_elap_recursive_separable.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]
        if hasattr(R, "ref_zoom_out"):
            R.ref_zoom_size.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_float]
            R.ref_zoom_out.argtypes = [_F, _F, C.c_int, C.c_int, C.c_float]
            R.ref_zoom_in.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, C.c_int]
        if hasattr(R, "ref_op_create"):                         # ref_harness_models.cpp
            vp = C.c_void_p
            R.ref_op_create.restype = vp
            R.ref_op_create.argtypes = [C.c_int] * 5
            R.ref_op_prepare.argtypes = [vp, _F, _F, C.c_int, _F, _F]
            R.ref_op_set_flow.argtypes = [vp, C.c_int, _F]
            R.ref_op_get_flow.argtypes = [vp, C.c_int, _F]
            R.ref_op_estimate.restype = C.c_float
            R.ref_op_estimate.argtypes = [vp] + [C.c_int] * 10
            R.ref_op_plane.argtypes = [vp, C.c_int, C.c_char_p, _F]
            R.ref_op_duals.argtypes = [vp, C.c_int, C.c_int, _F, _F, _F]
            R.ref_op_destroy.argtypes = [vp]
            R.ref_rgb_to_lab.argtypes = [_F, C.c_int, _F]
            R.ref_nltv_table.argtypes = [_F, C.c_int, C.c_int, C.c_int, _F, _F, _I, _I, _I, _F]
            R.ref_csad_table.argtypes = [C.c_int, C.c_int, _I, _I, _I]
            R.ref_gaussian1d_weight_c.argtypes = [_F, C.c_int]
            R.ref_op_weight.argtypes = [vp, C.c_int, _F]
            R.ref_parts_create.restype = vp
            R.ref_parts_create.argtypes = [_F, _F, C.c_int, _F, _F] + [C.c_int] * 5
            _bind_parts(R, "ref_parts_")
        _ref = R
    return _ref


def _bind_parts(L, prefix):
    """The partition hooks have one signature set on both sides (oracle: orc_parts_*, reference: ref_parts_*)."""
    vp = C.c_void_p
    g = lambda name: getattr(L, prefix + name)
    g("count").argtypes = [vp]
    g("geometry").argtypes = [vp, C.c_int, _I]
    g("set_image_plane").argtypes = [vp, C.c_int, C.c_int, _F]
    g("get_image_plane").argtypes = [vp, C.c_int, C.c_int, _F]
    g("set_part_plane").argtypes = [vp, C.c_int, C.c_int, C.c_int, _F]
    g("get_part_plane").argtypes = [vp, C.c_int, C.c_int, C.c_int, _F]
    g("get_part_frame").argtypes = [vp, C.c_int, C.c_int, _F]
    g("push").argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    g("image_queue_size").argtypes = [vp, C.c_int]
    g("transfer").argtypes = [vp, C.c_int]
    g("any_empty").argtypes = [vp]
    g("drain_queue").argtypes = [vp, C.c_int, C.c_int, _F, C.c_int]


class Parts:
    """One partition grid over an image-level state, on the oracle (side="orc") or the reference (side="ref")."""

    def __init__(self, side, i0, i1, i0n, i1n, cols, rows, wr=5):
        self.side = side
        i0, i1, i0n, i1n = f32(i0), f32(i1), f32(i0n), f32(i1n)
        self.h, self.w = i0n.shape
        if side == "ref":
            self.L, self.pre = ref(), "ref_parts_"
            self._h = self.L.ref_parts_create(i0, i1, i0.shape[0], i0n, i1n, self.w, self.h, cols, rows, wr)
        else:
            self.L, self.pre = lib(), "orc_parts_"
            self._h = self.L.orc_parts_create(i0n, i1n, self.w, self.h, cols, rows, wr)

    def _f(self, name):
        return getattr(self.L, self.pre + name)

    def count(self):
        return self._f("count")(self._h)

    def geometry(self, p):
        g = np.zeros(4, np.int32)
        self._f("geometry")(self._h, p, g)
        return tuple(int(x) for x in g)            # width, height, off_x, off_y

    def set_image(self, d, pid, v):
        self._f("set_image_plane")(self._h, d, pid, f32(v))

    def get_image(self, d, pid):
        v = np.empty((self.h, self.w), np.float32)
        self._f("get_image_plane")(self._h, d, pid, v)
        return v

    def set_part(self, p, d, pid, v):
        self._f("set_part_plane")(self._h, p, d, pid, f32(v))

    def get_part(self, p, d, pid):
        w, h, _, _ = self.geometry(p)
        v = np.empty((h, w), np.float32)
        self._f("get_part_plane")(self._h, p, d, pid, v)
        return v

    def part_frame(self, p, which):
        w, h, _, _ = self.geometry(p)
        v = np.empty((h, w), np.float32)
        self._f("get_part_frame")(self._h, p, which, v)
        return v

    def push(self, d, i, j, u, v, sim):
        self._f("push")(self._h, d, int(i), int(j), float(u), float(v), float(sim))

    def image_queue_size(self, d):
        return self._f("image_queue_size")(self._h, d)

    def transfer(self, img_to_part):
        self._f("transfer")(self._h, int(bool(img_to_part)))

    def any_empty(self):
        return bool(self._f("any_empty")(self._h))

    def drain(self, p, d, cap=1 << 16):
        out = np.empty((cap, 5), np.float32)
        k = self._f("drain_queue")(self._h, p, d, out, cap)
        return out[:k].copy()


def ref_weighted_operator(w, h, wr=5):
    """The reference's weighted window operator (src/tvl2w_model.cpp) with the window weights prepare_stuff gives it
    (gaussian1Dweight of src/mask.c).  Returns the handle for ref_local_tvl1w_estimate / _destroy."""
    R = ref()
    hnd = R.ref_local_tvl1w_create(w, h, wr)
    wv = np.zeros(2 * wr + 1, np.float32)
    R.ref_gaussian1d_weight_c(wv, wr)
    R.ref_local_tvl1w_set_weight(hnd, wv)
    return hnd


class RefOp:
    """The reference's operator surface of the local step (initialize_auxiliar_stuff / prepare_stuff / of_estimation,
    src/energy_model.cpp) for functional `method`, forward (0) and backward (1) direction."""

    def __init__(self, method, w, h, wr=5, max_iter_patch=4):
        self.R = ref()
        self.method, self.w, self.h, self.wr = method, w, h, wr
        self._h = self.R.ref_op_create(method, w, h, wr, max_iter_patch)

    def prepare(self, i0, i1):
        i0, i1 = f32(i0), f32(i1)
        pd = 1 if i0.ndim == 2 else i0.shape[0]
        a, b = np.empty((self.h, self.w), np.float32), np.empty((self.h, self.w), np.float32)
        self.R.ref_op_prepare(self._h, i0, i1, pd, a, b)
        return a, b

    def set_flow(self, d, u):
        self.R.ref_op_set_flow(self._h, d, f32(u))

    def get_flow(self, d):
        u = np.empty((2, self.h, self.w), np.float32)
        self.R.ref_op_get_flow(self._h, d, u)
        return u

    def estimate(self, d, i, j, wr_call):
        ii, ij, ei, ej = patch_index(wr_call, self.w, self.h, i, j)
        iiw = -(i - wr_call) if i - wr_call < 0 else 0          # get_relative_index_weight, src/local_faldoi.cpp:548-558
        ijw = -(j - wr_call) if j - wr_call < 0 else 0
        return float(np.float32(self.R.ref_op_estimate(self._h, d, i, j, ii, ij, ei, ej, wr_call, iiw, ijw)))

    def plane(self, d, name):
        v = np.empty((self.h, self.w), np.float32)
        if self.R.ref_op_plane(self._h, d, name.encode(), v) != 0:
            raise KeyError(name)
        return v

    def duals(self, d, which=0):
        sc = np.empty((24, self.h, self.w), np.float32)
        wp = np.empty_like(sc)
        wt = np.empty((self.h, self.w), np.float32)
        if self.R.ref_op_duals(self._h, d, which, sc, wp, wt) != 0:
            raise KeyError("no dual tables for method %d" % self.method)
        return sc, wp, wt

    def close(self):
        if self._h:
            self.R.ref_op_destroy(self._h)
            self._h = None


# ---------------------------------------------------------------- convenience wrappers

def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def centered_gradient(img):
    h, w = img.shape
    dx, dy = np.empty_like(img), np.empty_like(img)
    lib().orc_centered_gradient(img, dx, dy, w, h)
    return dx, dy


def preprocess_global(i0, i1, im1=None):
    """i0,i1[,im1]: (pd,h,w) float32 planar 0..255.  Returns (i0n, i1n) as the solver sees them."""
    i0, i1 = f32(i0), f32(i1)
    im1 = i1 if im1 is None else f32(im1)
    pd, h, w = i0.shape
    o0, o1, o2 = (np.empty((h, w), np.float32) for _ in range(3))
    lib().orc_preprocess_global(i0, i1, im1, w, h, pd, o0, o1, o2)
    return o0, o1


def tvl1_global(i0n, i1n, u0, lam=40.0, theta=0.3, tau=0.125, tol=0.01, warps=5,
                max_inner=400, tol_stop=True, xi=None):
    """Runs the oracle solver.  u0: (2,h,w).  Returns (u (2,h,w), iters[warps], err[warps], xi)."""
    h, w = i0n.shape
    u = f32(u0).copy()
    xi = np.zeros((4, h, w), np.float32) if xi is None else f32(xi).copy()
    iters = np.zeros(warps, np.int32)
    errs = np.zeros(warps, np.float32)
    lib().orc_tvl1_global(f32(i0n), f32(i1n), u[0], u[1], xi[0], xi[1], xi[2], xi[3],
                          lam, theta, tau, tol, w, h, warps, max_inner, int(bool(tol_stop)),
                          iters, errs)
    return u, iters, errs, xi


def image_to_lab(rgb):
    rgb = f32(rgb)
    out = np.empty_like(rgb)
    lib().orc_image_to_lab(rgb, rgb.shape[1] * rgb.shape[2], out)
    return out


def nltvcsad_global(i0n, i1n, lab, u0, lam=0.85, theta=0.3, tau=0.1, warps=1, max_inner=400):
    """Reference parameter overrides for -m 6: src/global_faldoi.cpp:2138-2140."""
    h, w = i0n.shape
    u = f32(u0).copy()
    lab = f32(lab)
    lib().orc_nltvcsad_global(f32(i0n), f32(i1n), lab, lab.shape[0], lam, theta, tau, w, h, warps, max_inner,
                              u[0], u[1])
    return u


def nltvl1_global(i0n, i1n, lab, u0, lam=2.0, theta=0.3, tau=0.1, warps=1, max_inner=400):
    """Reference parameter overrides for -m 2: src/global_faldoi.cpp:2146-2148."""
    h, w = i0n.shape
    u = f32(u0).copy()
    lab = f32(lab)
    lib().orc_nltvl1_global(f32(i0n), f32(i1n), lab, lab.shape[0], lam, theta, tau, w, h, warps, max_inner,
                            u[0], u[1])
    return u


def zoom_size(nx, ny, factor):
    a, b = C.c_int(), C.c_int()
    lib().orc_zoom_size(nx, ny, C.byref(a), C.byref(b), factor)
    return a.value, b.value


def zoom_out(img, factor):
    img = f32(img)
    h, w = img.shape
    ww, hh = zoom_size(w, h, factor)
    out = np.empty((hh, ww), np.float32)
    lib().orc_zoom_out(img, out, w, h, factor)
    return out


def zoom_in(img, ww, hh):
    img = f32(img)
    h, w = img.shape
    out = np.empty((hh, ww), np.float32)
    lib().orc_zoom_in(img, out, w, h, ww, hh)
    return out


def patch_index(wr, w, h, i, j):
    """get_index_patch (src/utils.cpp:36-54, factor 1): clipped (2wr+1)^2 window -> (ii, ij, ei, ej)."""
    return max(i - wr, 0), max(j - wr, 0), min(i + 1 + wr, w), min(j + 1 + wr, h)


class LocalTvl1:
    """Stateful local TV-L1 patch operator (oracle side): persistent xi/div planes, zero at creation."""

    def __init__(self, i0n, i1n):
        self.i0, self.i1 = f32(i0n), f32(i1n)
        self.h, self.w = self.i0.shape
        self.i1x, self.i1y = centered_gradient(self.i1)
        self.state = np.zeros((6, self.h, self.w), np.float32)

    def estimate(self, u, patch, max_iter_patch=4):
        """u: (2,h,w) float32, updated in place inside the patch.  Returns (energy, iterations)."""
        it = C.c_int(0)
        ii, ij, ei, ej = patch
        e = lib().orc_local_tvl1_estimate(self.i0, self.i1, self.i1x, self.i1y, u[0], u[1], self.state,
                                          self.w, self.h, ii, ij, ei, ej, max_iter_patch, C.byref(it))
        return float(np.float32(e)), it.value


def preprocess_local(i0, i1):
    """prepare_stuff's image part (TV-l2 coupled): (pd,h,w) or (h,w) frames -> normalised (h,w) pair."""
    i0, i1 = f32(i0), f32(i1)
    pd = 1 if i0.ndim == 2 else i0.shape[0]
    h, w = i0.shape[-2:]
    a, b = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
    lib().orc_preprocess_local(i0, i1, w, h, pd, a, b)
    return a, b


def local_faldoi(i0n, i1n, go, ba, wr=5, loc_it=3, max_iter_patch=4, fb_thresh=2.0, sal0=None, sal1=None, method=0,
                 split_img=0, h_parts=3, v_parts=2, first_h_parts=0, first_v_parts=0, partial_res=False):
    """match_growing_variational (-m 0, no partitions).  go/ba: (2,h,w) seed flows, NaN = no seed.
    Returns (out_flow (2,h,w), ene_val (h,w), stats[6]); with partial_res=True a fourth value: what -partial_res 1 writes
    (src/local_faldoi.cpp:942-1036), as a dict {(percent, iteration, part_idx): flow (2,h',w')}."""
    i0n, i1n, go, ba = f32(i0n), f32(i1n), f32(go), f32(ba)
    h, w = i0n.shape
    out, ene, st = np.empty((2, h, w), np.float32), np.empty((h, w), np.float32), np.zeros(6, np.int64)
    keep = [f32(s) if s is not None else None for s in (sal0, sal1)]
    ptr = [k.ctypes.data if k is not None else None for k in keep]
    L = lib()
    L.orc_partial_enable(1 if partial_res else 0)
    L.orc_local_faldoi_split(i0n, i1n, go, ba, ptr[0], ptr[1], w, h, wr, loc_it, max_iter_patch, fb_thresh, method,
                             split_img, h_parts, v_parts, first_h_parts, first_v_parts, out, ene, st)
    if not partial_res:
        return out, ene, st
    dumps = {}
    for k in range(L.orc_partial_count()):
        meta = np.zeros(5, np.int32)
        L.orc_partial_get(k, meta.ctypes.data_as(C.c_void_p), None)
        flow = np.empty((2, int(meta[4]), int(meta[3])), np.float32)
        L.orc_partial_get(k, meta.ctypes.data_as(C.c_void_p), flow.ctypes.data_as(C.c_void_p))
        dumps[(int(meta[0]), int(meta[1]), int(meta[2]))] = flow
    L.orc_partial_enable(0)
    return out, ene, st, dumps


def local_faldoi_raw(i0, i1, go, ba, method, wr=5, loc_it=3, max_iter_patch=4, fb_thresh=2.0):
    """match_growing_variational for any functional 0..7 from the raw (pd,h,w) frames (whole image)."""
    i0, i1, go, ba = f32(i0), f32(i1), f32(go), f32(ba)
    pd, h, w = i0.shape
    out, ene, st = np.empty((2, h, w), np.float32), np.empty((h, w), np.float32), np.zeros(6, np.int64)
    lib().orc_local_faldoi_raw(i0, i1, pd, go, ba, None, None, w, h, wr, loc_it, max_iter_patch, fb_thresh, method, out, ene, st)
    return out, ene, st


def tvcsad_global(i0n, i1n, u0, lam=0.85, theta=0.3, tau=0.125, tol=0.01, warps=5, max_inner=400, xi=None):
    """tvcsad_PD with the reference's one-thread stop test.  Returns (u, iters[warps], err[warps], xi)."""
    i0n, i1n = f32(i0n), f32(i1n)
    h, w = i0n.shape
    u = f32(u0).copy()
    xi = np.zeros((4, h, w), np.float32) if xi is None else f32(xi).copy()
    its, errs = np.zeros(warps, np.int32), np.zeros(warps, np.float32)
    lib().orc_tvcsad_global(i0n, i1n, lam, theta, tau, tol, w, h, warps, max_inner, u[0], u[1], xi, its, errs)
    return u, its, errs, xi
