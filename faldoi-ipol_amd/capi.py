"""ctypes binding of the C ABI (include/faldoi_hip.h) -- the binding a host-language caller
would write; used by tests/, bench.py and __graft_entry__.  No torch types cross this boundary.

Fails loudly when libfaldoi_hip.so is missing: there is no CPU fallback in the product.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_STATE = range(6)
MAX_CHUNK = 16

# every symbol include/faldoi_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "faldoi_abi_version", "faldoi_device_count", "faldoi_ctx_create", "faldoi_ctx_destroy",
    "faldoi_ctx_synchronize", "faldoi_ctx_persist_stats", "faldoi_ctx_persist_retry", "faldoi_ctx_exact_redos", "faldoi_debug_nl_check", "faldoi_ctx_last_transfer_ms", "faldoi_last_error", "faldoi_tvl1_default_params", "faldoi_tvl1_global",
    "faldoi_tvl1_batch_create", "faldoi_tvl1_batch_destroy", "faldoi_tvl1_batch_upload",
    "faldoi_tvl1_batch_reset", "faldoi_tvl1_batch_solve", "faldoi_tvl1_batch_download",
    "faldoi_tvl1_batch_last_timing", "faldoi_tvl1_batch_last_schedule", "faldoi_tvl1_batch_last_persistent", "faldoi_tvl1_schedule_query", "faldoi_centered_gradient", "faldoi_forward_gradient",
    "faldoi_divergence", "faldoi_bicubic_warp", "faldoi_selftest_division",
    "faldoi_preprocess_global", "faldoi_tvl1_batch_upload_images",
    "faldoi_nltvcsad_default_params", "faldoi_nltvcsad_global", "faldoi_image_to_lab",
    "faldoi_nltvl1_default_params", "faldoi_nltvl1_global",
    "faldoi_zoom_size", "faldoi_zoom_out", "faldoi_zoom_in",
    "faldoi_local_tvl1_create", "faldoi_local_tvl1_destroy", "faldoi_local_tvl1_set_flow",
    "faldoi_local_tvl1_get_flow", "faldoi_local_tvl1_estimate", "faldoi_local_tvl1_set_fixed",
    "faldoi_local_tvl1_interpolate",
    "faldoi_local_default_params", "faldoi_preprocess_local", "faldoi_match_growing_variational",
    "faldoi_match_growing_batch", "faldoi_queue_replay",
    "faldoi_fb_consistency_check", "faldoi_tvcsad_default_params", "faldoi_tvcsad_global",
    "faldoi_queue_replay_device",
]


class Tvl1Params(C.Structure):
    _fields_ = [("lambda_", C.c_float), ("theta", C.c_float), ("tau", C.c_float), ("tol", C.c_float),
                ("warps", C.c_int), ("max_inner", C.c_int), ("tol_stop", C.c_int), ("chunk", C.c_int),
                ("verbose", C.c_int)]


class NltvCsadParams(C.Structure):
    _fields_ = [("lambda_", C.c_float), ("theta", C.c_float), ("tau", C.c_float), ("warps", C.c_int),
                ("max_inner", C.c_int), ("verbose", C.c_int)]


class Tvl1Stats(C.Structure):
    _fields_ = [("total_iters", C.c_long), ("warps", C.c_int), ("iters_cap", C.c_int),
                ("iters", C.POINTER(C.c_int)), ("err", C.POINTER(C.c_float)),
                ("solve_ms", C.c_double), ("chunk_launches", C.c_int)]


# faldoi_local_params::partial_cb (user, pair, percent, iteration, part_idx, w, h, flow)
PARTIAL_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float))


class LocalParams(C.Structure):
    _fields_ = [("w_radio", C.c_int), ("iterations_of", C.c_int), ("max_iter_patch", C.c_int), ("epsilon", C.c_float),
                ("preprocessed", C.c_int), ("budget", C.c_int), ("verbose", C.c_int), ("method", C.c_int),
                ("split_img", C.c_int), ("h_parts", C.c_int), ("v_parts", C.c_int),
                ("first_h_parts", C.c_int), ("first_v_parts", C.c_int),
                ("partial_cb", PARTIAL_CB), ("partial_user", C.c_void_p)]


class LocalStats(C.Structure):
    _fields_ = [("pops", C.c_longlong * 2), ("solves", C.c_longlong * 2), ("fills", C.c_longlong * 2),
                ("fb_kept", C.c_float * 16), ("growing_ms", C.c_float), ("launches", C.c_int),
                ("cycles", C.c_longlong * 3)]


class FaldoiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("faldoi_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


_lib_path = None


def library_path():
    """The library this process loaded (the product's unless FALDOI_LIB named another one); before the first call of lib(): the product's."""
    return _lib_path or _build.SO


def loaded_library():
    """Absolute path of the library this process runs on (None before the first call of lib()) and whether it is the product's."""
    return _lib_path, (_lib_path is not None and os.path.abspath(_build.SO) == _lib_path)


def lib():
    """Loads libfaldoi_hip.so (building it if the sources are newer).  Raises if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    so = _build.SO
    if os.environ.get("FALDOI_LIB"):
        # Development aid ONLY (timing-only builds of tools/nl_whatif.sh, the range-checked build of build.build_check): the library named
        # here is loaded as it is -- no freshness check, and a what-if build returns wrong results by construction.  Said loudly, and
        # recorded: loaded_library() is what bench.py prints and what makes it refuse every counter-derived field.
        so = os.environ["FALDOI_LIB"]
        import sys
        sys.stderr.write("faldoi_ipol_amd: FALDOI_LIB is set -- loading %s INSTEAD of the product library %s\n" % (so, _build.SO))
    elif _build.needs_build():
        so = _build.build()
    if not os.path.exists(so):
        raise RuntimeError("libfaldoi_hip.so is missing and could not be built; the HIP path has no fallback")
    L = C.CDLL(so)
    global _lib_path
    _lib_path = os.path.abspath(so)
    vp = C.c_void_p
    L.faldoi_abi_version.restype = C.c_int
    L.faldoi_last_error.restype = C.c_char_p
    L.faldoi_device_count.argtypes = [C.POINTER(C.c_int)]
    L.faldoi_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.faldoi_ctx_destroy.argtypes = [vp]
    L.faldoi_ctx_destroy.restype = None
    L.faldoi_ctx_synchronize.argtypes = [vp]
    L.faldoi_ctx_persist_stats.argtypes = [vp] + [C.POINTER(C.c_long)] * 4
    L.faldoi_ctx_persist_retry.argtypes = [vp]
    L.faldoi_ctx_exact_redos.argtypes = [vp, C.POINTER(C.c_long)]
    L.faldoi_ctx_last_transfer_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.faldoi_debug_nl_check.argtypes = [vp, C.POINTER(C.c_uint * 4), C.c_int, C.POINTER(C.c_int)]
    L.faldoi_tvl1_default_params.argtypes = [C.POINTER(Tvl1Params)]
    L.faldoi_tvl1_default_params.restype = None
    L.faldoi_tvl1_global.argtypes = [vp, _F, _F, _F, vp, C.c_int, C.c_int, C.POINTER(Tvl1Params),
                                     C.POINTER(Tvl1Stats)]
    L.faldoi_tvl1_batch_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.faldoi_tvl1_batch_destroy.argtypes = [vp]
    L.faldoi_tvl1_batch_destroy.restype = None
    L.faldoi_tvl1_batch_upload.argtypes = [vp, C.c_int, _F, _F, _F, vp]
    L.faldoi_tvl1_batch_reset.argtypes = [vp]
    L.faldoi_tvl1_batch_solve.argtypes = [vp, C.POINTER(Tvl1Params), C.POINTER(Tvl1Stats)]
    L.faldoi_tvl1_batch_download.argtypes = [vp, C.c_int, _F, vp]
    L.faldoi_tvl1_batch_last_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                                C.POINTER(C.c_long), C.POINTER(C.c_double)]
    L.faldoi_tvl1_batch_last_schedule.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.faldoi_tvl1_batch_last_persistent.argtypes = [vp, C.POINTER(C.c_int)]
    L.faldoi_tvl1_schedule_query.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Tvl1Params), C.c_int] + [C.POINTER(C.c_int)] * 5
    L.faldoi_centered_gradient.argtypes = [vp, _F, _F, _F, C.c_int, C.c_int]
    L.faldoi_forward_gradient.argtypes = [vp, _F, _F, _F, C.c_int, C.c_int]
    L.faldoi_divergence.argtypes = [vp, _F, _F, _F, C.c_int, C.c_int]
    L.faldoi_bicubic_warp.argtypes = [vp, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int]
    L.faldoi_preprocess_global.argtypes = [vp, _F, _F, vp, C.c_int, C.c_int, C.c_int, _F, _F]
    L.faldoi_tvl1_batch_upload_images.argtypes = [vp, C.c_int, _F, _F, vp, C.c_int, _F, vp]
    L.faldoi_nltvcsad_default_params.argtypes = [C.POINTER(NltvCsadParams)]
    L.faldoi_nltvcsad_default_params.restype = None
    L.faldoi_nltvcsad_global.argtypes = [vp, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.POINTER(NltvCsadParams), _F,
                                         C.POINTER(C.c_double)]
    L.faldoi_image_to_lab.argtypes = [_F, C.c_int, C.c_int, _F]
    L.faldoi_nltvl1_default_params.argtypes = [C.POINTER(NltvCsadParams)]
    L.faldoi_nltvl1_default_params.restype = None
    L.faldoi_nltvl1_global.argtypes = L.faldoi_nltvcsad_global.argtypes
    L.faldoi_zoom_size.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.faldoi_zoom_out.argtypes = [vp, _F, _F, C.c_int, C.c_int, C.c_float]
    L.faldoi_zoom_in.argtypes = [vp, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int]
    L.faldoi_local_tvl1_create.argtypes = [vp, _F, _F, C.c_int, C.c_int, C.POINTER(vp)]
    L.faldoi_local_tvl1_destroy.argtypes = [vp]
    L.faldoi_local_tvl1_destroy.restype = None
    L.faldoi_local_tvl1_set_flow.argtypes = [vp, _F]
    L.faldoi_local_tvl1_get_flow.argtypes = [vp, _F]
    L.faldoi_local_tvl1_estimate.argtypes = [vp, C.c_int, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"),
                                             C.c_int, _F, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")]
    L.faldoi_local_tvl1_set_fixed.argtypes = [vp, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")]
    L.faldoi_local_tvl1_interpolate.argtypes = [vp, C.c_int, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")]
    L.faldoi_selftest_division.argtypes = [vp, C.c_uint, C.c_ulonglong, C.POINTER(C.c_ulonglong)]
    L.faldoi_local_default_params.argtypes = [C.POINTER(LocalParams)]
    L.faldoi_local_default_params.restype = None
    L.faldoi_preprocess_local.argtypes = [vp, _F, _F, C.c_int, C.c_int, C.c_int, _F, _F]
    L.faldoi_match_growing_variational.argtypes = [vp, _F, _F, _F, _F, vp, vp, C.c_int, C.c_int, C.c_int,
                                                   C.POINTER(LocalParams), _F, _F, C.POINTER(LocalStats)]
    _I32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    L.faldoi_queue_replay.argtypes = [C.c_int, _I32, _F, _I32, C.POINTER(C.c_int)]
    L.faldoi_queue_replay_device.argtypes = [vp, C.c_int, _I32, _F, _I32, C.POINTER(C.c_int)]
    L.faldoi_fb_consistency_check.argtypes = [vp, _F, _F, C.c_int, C.c_int, C.c_float, _I32, C.POINTER(C.c_int)]
    L.faldoi_tvcsad_default_params.argtypes = [C.POINTER(Tvl1Params)]
    L.faldoi_tvcsad_default_params.restype = None
    L.faldoi_tvcsad_global.argtypes = [vp, _F, _F, _F, vp, C.c_int, C.c_int, C.POINTER(Tvl1Params), C.POINTER(Tvl1Stats)]
    pp = C.POINTER(C.c_void_p)
    L.faldoi_match_growing_batch.argtypes = [vp, C.c_int, pp, pp, pp, pp, pp, pp, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(LocalParams), pp, pp, C.POINTER(LocalStats)]
    _lib = L
    return L


def _check(rc):
    if rc != OK:
        raise FaldoiError(rc, lib().faldoi_last_error().decode("utf-8", "replace"))


def default_params(**over):
    p = Tvl1Params()
    lib().faldoi_tvl1_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, "lambda_" if k in ("lambda", "lam") else k, v)
    return p


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _as(a, shape, name):
    """float32, C-contiguous, and of exactly `shape`: the C ABI takes raw pointers and ONE (nx, ny), so an array of another size
    would be read (or written) past its end inside the library -- a segmentation fault of the host process, with no message."""
    a = _f32(a)
    if a.shape != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One device + one HIP stream (faldoi_ctx)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().faldoi_ctx_create(int(device), C.byref(self._h)))
        self.device = int(device)

    def close(self):
        if self._h:
            lib().faldoi_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def persist_stats(self):
        """dict(launches, fallbacks, busy, backed_off): this context's history with the persistent TV-L1 kernel."""
        v = [C.c_long() for _ in range(4)]
        _check(lib().faldoi_ctx_persist_stats(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("launches", "fallbacks", "busy", "backed_off"), (x.value for x in v)))

    def persist_retry(self):
        _check(lib().faldoi_ctx_persist_retry(self._h))

    def last_transfer_ms(self):
        """(h2d_ms, d2h_ms): host wall time of this context's last batch upload / download (inside tvl1_global too)."""
        a, b = C.c_double(0), C.c_double(0)
        _check(lib().faldoi_ctx_last_transfer_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def exact_redos(self):
        """Solves repeated with the general quotients (a divisor was an even integer >= 6: images that were not normalised)."""
        n = C.c_long(0)
        _check(lib().faldoi_ctx_exact_redos(self._h, C.byref(n)))
        return n.value

    # ---- primitives (host arrays in/out) ----
    def centered_gradient(self, img):
        img = _f32(img)
        h, w = img.shape
        dx, dy = np.empty_like(img), np.empty_like(img)
        _check(lib().faldoi_centered_gradient(self._h, img, dx, dy, w, h))
        return dx, dy

    def forward_gradient(self, img):
        img = _f32(img)
        h, w = img.shape
        fx, fy = np.empty_like(img), np.empty_like(img)
        _check(lib().faldoi_forward_gradient(self._h, img, fx, fy, w, h))
        return fx, fy

    def divergence(self, v1, v2):
        v1 = _f32(v1)
        h, w = v1.shape
        v2 = _as(v2, (h, w), "v2")
        d = np.empty_like(v1)
        _check(lib().faldoi_divergence(self._h, v1, v2, d, w, h))
        return d

    def bicubic_warp(self, img, u, v, border_out=True):
        img = _f32(img)
        h, w = img.shape
        u, v = _as(u, (h, w), "u"), _as(v, (h, w), "v")
        o = np.empty_like(img)
        _check(lib().faldoi_bicubic_warp(self._h, img, u, v, o, w, h, int(bool(border_out))))
        return o

    def preprocess_global(self, i0, i1, im1=None):
        """(pd,h,w) planar images 0..255 -> (i0n, i1n) as the solver takes them (GPU)."""
        i0 = _f32(i0)
        pd, h, w = i0.shape
        i1 = _as(i1, (pd, h, w), "i1")
        im1 = None if im1 is None else _as(im1, (pd, h, w), "im1")
        o0, o1 = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
        _check(lib().faldoi_preprocess_global(self._h, i0, i1, _ptr(im1), w, h, pd, o0, o1))
        return o0, o1

    def preprocess_local(self, i0, i1):
        """prepare_stuff's image part of the local step: (pd,h,w) or (h,w) frames -> normalised (h,w) pair."""
        i0 = _f32(i0)
        i1 = _as(i1, i0.shape, "i1")
        pd = 1 if i0.ndim == 2 else i0.shape[0]
        h, w = i0.shape[-2:]
        o0, o1 = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
        _check(lib().faldoi_preprocess_local(self._h, i0, i1, w, h, pd, o0, o1))
        return o0, o1

    def tvcsad_global(self, i0n, i1n, u0, xi=None, **params):
        """Mirror of tvcsad_PD (one-thread stop test).  Returns (u, iters[warps], err[warps], xi, stats)."""
        i0n = _f32(i0n)
        h, w = i0n.shape
        i1n = _as(i1n, (h, w), "i1n")
        u = _as(u0, (2, h, w), "u0").copy()
        xi = np.zeros((4, h, w), np.float32) if xi is None else _as(xi, (4, h, w), "xi").copy()
        p = Tvl1Params()
        lib().faldoi_tvcsad_default_params(C.byref(p))
        for k, v in params.items():
            setattr(p, "lambda_" if k in ("lambda", "lam") else k, v)
        its, errs = np.zeros(p.warps, np.int32), np.zeros(p.warps, np.float32)
        st = Tvl1Stats()
        st.iters_cap = p.warps
        st.iters = its.ctypes.data_as(C.POINTER(C.c_int))
        st.err = errs.ctypes.data_as(C.POINTER(C.c_float))
        _check(lib().faldoi_tvcsad_global(self._h, i0n, i1n, u, xi.ctypes.data_as(C.c_void_p), w, h, C.byref(p), C.byref(st)))
        return u, its, errs, xi, st

    def queue_replay(self, ops, sims):
        """queue_replay on the device queue (one wavefront).  Returns the popped ids."""
        ops = np.ascontiguousarray(ops, np.int32)
        sims = _f32(sims)
        out = np.empty(len(ops), np.int32)
        n = C.c_int(0)
        _check(lib().faldoi_queue_replay_device(self._h, len(ops), ops, sims, out, C.byref(n)))
        return out[:n.value]

    def fb_consistency_check(self, in0, in1, epsilon=2.0):
        """Mirror of fb_consistency_check: (2,h,w) forward / backward flows -> trust (h,w) int32."""
        in0 = _f32(in0)
        h, w = in0.shape[1:]
        in0, in1 = _as(in0, (2, h, w), "in0"), _as(in1, (2, h, w), "in1")
        trust = np.empty((h, w), np.int32)
        _check(lib().faldoi_fb_consistency_check(self._h, in0, in1, w, h, epsilon, trust, None))
        return trust

    @staticmethod
    def _local_params(params):
        p = LocalParams()
        lib().faldoi_local_default_params(C.byref(p))
        for k, v in params.items():
            if k == "partial_res":
                # -partial_res 1: `v` is a dict that receives {(pair, percent, iteration, part_idx): flow (2,h,w)} (or a falsy value)
                if v is not None and v is not False:
                    sink = v

                    def keep(_user, pair, percent, iteration, part_idx, w, h, flow, sink=sink):
                        sink[(pair, percent, iteration, part_idx)] = np.ctypeslib.as_array(flow, shape=(2, h, w)).copy()
                    p.partial_cb = PARTIAL_CB(keep)
                    p._keepalive = p.partial_cb                     # the C side holds a bare pointer during the call
                continue
            setattr(p, {"wr": "w_radio", "loc_it": "iterations_of", "fb_thresh": "epsilon"}.get(k, k), v)
        return p

    @staticmethod
    def _check_pair_shapes(i0, i1, go, ba, sal0=None, sal1=None, what="pair"):
        """The C ABI takes raw pointers and ONE (nx, ny, pd): every array of a pair must have that shape."""
        if i0.ndim not in (2, 3):
            raise ValueError("%s: frames must be (h,w) or (pd,h,w), got %s" % (what, i0.shape))
        h, w = i0.shape[-2:]
        if i1.shape != i0.shape:
            raise ValueError("%s: i1 has shape %s, i0 has %s" % (what, i1.shape, i0.shape))
        for name, a in (("go", go), ("ba", ba)):
            if a.shape != (2, h, w):
                raise ValueError("%s: %s has shape %s, expected %s" % (what, name, a.shape, (2, h, w)))
        for name, a in (("sal0", sal0), ("sal1", sal1)):
            if a is not None and a.shape != (h, w):
                raise ValueError("%s: %s has shape %s, expected %s" % (what, name, a.shape, (h, w)))
        return (1 if i0.ndim == 2 else i0.shape[0]), h, w

    def match_growing(self, i0, i1, go, ba, sal0=None, sal1=None, **params):
        """Mirror of match_growing_variational (-m 0, no partitions).  i0/i1: (pd,h,w) frames 0..255, or normalised
        (h,w) planes with preprocessed=1.  go/ba: (2,h,w) seed flows, NaN = no seed.
        Returns (out_flow (2,h,w), ene_val (h,w), LocalStats)."""
        i0, i1, go, ba = _f32(i0), _f32(i1), _f32(go), _f32(ba)
        sal0 = None if sal0 is None else _f32(sal0)
        sal1 = None if sal1 is None else _f32(sal1)
        pd, h, w = self._check_pair_shapes(i0, i1, go, ba, sal0, sal1)
        out, ene, st = np.empty((2, h, w), np.float32), np.empty((h, w), np.float32), LocalStats()
        p = self._local_params(params)
        _check(lib().faldoi_match_growing_variational(self._h, go, ba, i0, i1, _ptr(sal0), _ptr(sal1), w, h, pd,
                                                      C.byref(p), ene, out, C.byref(st)))
        return out, ene, st

    def match_growing_batch(self, i0s, i1s, gos, bas, **params):
        """Several pairs of one size at once: lists of arrays as in match_growing.  Returns (flows, energies, stats)."""
        k = len(i0s)
        i0s, i1s, gos, bas = ([_f32(a) for a in x] for x in (i0s, i1s, gos, bas))
        if k < 1 or not (len(i1s) == len(gos) == len(bas) == k):
            raise ValueError("match_growing_batch: the four lists must have one entry per pair")
        shapes = [self._check_pair_shapes(i0s[j], i1s[j], gos[j], bas[j], what="pair %d" % j) for j in range(k)]
        if any(sh != shapes[0] for sh in shapes):
            raise ValueError("match_growing_batch: all pairs must have one size and channel count, got %s" % sorted(set(shapes)))
        pd, h, w = shapes[0]
        outs = [np.empty((2, h, w), np.float32) for _ in range(k)]
        enes = [np.empty((h, w), np.float32) for _ in range(k)]

        def arr(xs):
            return (C.c_void_p * k)(*[x.ctypes.data for x in xs])
        st = LocalStats()
        p = self._local_params(params)
        _check(lib().faldoi_match_growing_batch(self._h, k, arr(gos), arr(bas), arr(i0s), arr(i1s), None, None, w, h, pd,
                                                C.byref(p), arr(enes), arr(outs), C.byref(st)))
        return outs, enes, st

    def _nl_global(self, defaults, solve, i0n, i1n, lab, u0, params):
        i0n, lab = _f32(i0n), _f32(lab)
        h, w = i0n.shape
        i1n = _as(i1n, (h, w), "i1n")
        if lab.ndim != 3:
            raise ValueError("lab must be (pd,h,w), got %s" % (lab.shape,))
        lab = _as(lab, (lab.shape[0], h, w), "lab")
        u = _as(u0, (2, h, w), "u0").copy()
        p = NltvCsadParams()
        defaults(C.byref(p))
        for k, v in params.items():
            setattr(p, "lambda_" if k in ("lambda", "lam") else k, v)
        ms = C.c_double(0)
        _check(solve(self._h, i0n, i1n, lab, lab.shape[0], w, h, C.byref(p), u, C.byref(ms)))
        return u, ms.value

    def nltvcsad_global(self, i0n, i1n, lab, u0, **params):
        """Mirror of nltvcsad_PD.  Returns (u, device ms)."""
        return self._nl_global(lib().faldoi_nltvcsad_default_params, lib().faldoi_nltvcsad_global,
                               i0n, i1n, lab, u0, params)

    def nl_check_report(self, reset=True):
        """faldoi_debug_nl_check: (bits, violations, largest offending value, launches checked, compiled_in)."""
        out = (C.c_uint * 4)()
        comp = C.c_int(0)
        _check(lib().faldoi_debug_nl_check(self._h, C.byref(out), 1 if reset else 0, C.byref(comp)))
        return int(out[0]), int(out[1]), int(out[2]), int(out[3]), bool(comp.value)

    def nltvl1_global(self, i0n, i1n, lab, u0, **params):
        """Mirror of nltvl1_PD.  Returns (u, device ms)."""
        return self._nl_global(lib().faldoi_nltvl1_default_params, lib().faldoi_nltvl1_global,
                               i0n, i1n, lab, u0, params)

    def zoom_out(self, img, factor):
        img = _f32(img)
        h, w = img.shape
        ww, hh = C.c_int(), C.c_int()
        _check(lib().faldoi_zoom_size(w, h, factor, C.byref(ww), C.byref(hh)))
        out = np.empty((hh.value, ww.value), np.float32)
        _check(lib().faldoi_zoom_out(self._h, img, out, w, h, factor))
        return out

    def zoom_in(self, img, ww, hh):
        img = _f32(img)
        h, w = img.shape
        out = np.empty((hh, ww), np.float32)
        _check(lib().faldoi_zoom_in(self._h, img, out, w, h, ww, hh))
        return out

    def selftest_division(self, count=1 << 28, seed=12345):
        bad = C.c_ulonglong(0)
        _check(lib().faldoi_selftest_division(self._h, int(seed), int(count), C.byref(bad)))
        self.selftest_report = lib().faldoi_last_error().decode()      # breakdown, incl. what the one-product form refused
        if bad.value:
            print(self.selftest_report)
        return bad.value

    # ---- one-call solver (mirror of tvl2OF) ----
    def tvl1_global(self, i0n, i1n, u0, xi=None, **params):
        i0n = _f32(i0n)
        h, w = i0n.shape
        i1n = _as(i1n, (h, w), "i1n")
        u = _as(u0, (2, h, w), "u0").copy()
        xi = None if xi is None else _as(xi, (4, h, w), "xi").copy()
        p = default_params(**params)
        iters = np.zeros(p.warps, np.int32)
        errs = np.zeros(p.warps, np.float32)
        st = Tvl1Stats(iters_cap=p.warps, iters=iters.ctypes.data_as(C.POINTER(C.c_int)),
                       err=errs.ctypes.data_as(C.POINTER(C.c_float)))
        _check(lib().faldoi_tvl1_global(self._h, i0n, i1n, u, _ptr(xi), w, h, C.byref(p), C.byref(st)))
        return u, iters, errs, xi, st.solve_ms


class Tvl1Batch:
    """npairs same-size pairs resident in HBM (faldoi_tvl1_batch)."""

    def __init__(self, ctx, npairs, w, h):
        self.ctx, self.npairs, self.w, self.h = ctx, int(npairs), int(w), int(h)
        self._h = C.c_void_p()
        _check(lib().faldoi_tvl1_batch_create(ctx._h, self.npairs, self.w, self.h, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().faldoi_tvl1_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, pair, i0n, i1n, u0, xi=None):
        hw = (self.h, self.w)
        xi = None if xi is None else _as(xi, (4,) + hw, "xi")
        _check(lib().faldoi_tvl1_batch_upload(self._h, int(pair), _as(i0n, hw, "i0n"), _as(i1n, hw, "i1n"), _as(u0, (2,) + hw, "u0"), _ptr(xi)))

    def upload_images(self, pair, i0, i1, u0, im1=None, xi=None):
        i0 = _f32(i0)
        hw = (self.h, self.w)
        if i0.ndim != 3:
            raise ValueError("i0 must be (pd,h,w), got %s" % (i0.shape,))
        i0 = _as(i0, (i0.shape[0],) + hw, "i0")
        i1 = _as(i1, i0.shape, "i1")
        im1 = None if im1 is None else _as(im1, i0.shape, "im1")
        xi = None if xi is None else _as(xi, (4,) + hw, "xi")
        _check(lib().faldoi_tvl1_batch_upload_images(self._h, int(pair), i0, i1, _ptr(im1), i0.shape[0],
                                                     _as(u0, (2,) + hw, "u0"), _ptr(xi)))

    def solve(self, **params):
        p = default_params(**params)
        n = self.npairs
        iters = np.zeros((n, p.warps), np.int32)
        errs = np.zeros((n, p.warps), np.float32)
        stats = (Tvl1Stats * n)()
        for i in range(n):
            stats[i].iters_cap = p.warps
            stats[i].iters = iters[i].ctypes.data_as(C.POINTER(C.c_int))
            stats[i].err = errs[i].ctypes.data_as(C.POINTER(C.c_float))
        _check(lib().faldoi_tvl1_batch_solve(self._h, C.byref(p), stats))
        return iters, errs, stats[0].solve_ms, stats[0].chunk_launches

    def last_timing(self):
        ms, nl, pi, cms = C.c_double(), C.c_int(), C.c_long(), C.c_double()
        _check(lib().faldoi_tvl1_batch_last_timing(self._h, C.byref(ms), C.byref(nl), C.byref(pi), C.byref(cms)))
        return ms.value, nl.value, pi.value, cms.value

    def last_schedule(self):
        """(rows per thread, waves per workgroup, inner iterations per launch) of the last solve."""
        r, nw, k = C.c_int(), C.c_int(), C.c_int()
        _check(lib().faldoi_tvl1_batch_last_schedule(self._h, C.byref(r), C.byref(nw), C.byref(k)))
        return r.value, nw.value, k.value

    def last_persistent(self):
        """True when the last solve ran as one launch of the persistent kernel."""
        v = C.c_int()
        _check(lib().faldoi_tvl1_batch_last_persistent(self._h, C.byref(v)))
        return bool(v.value)

    def download(self, pair, want_xi=False):
        u = np.empty((2, self.h, self.w), np.float32)
        xi = np.empty((4, self.h, self.w), np.float32) if want_xi else None
        _check(lib().faldoi_tvl1_batch_download(self._h, int(pair), u, _ptr(xi)))
        return (u, xi) if want_xi else u


def schedule_query(npairs, w, h, compute_units=256, **params):
    """(persistent, rows per thread, waves, iterations per launch or exchange, workgroups) batch_solve would choose (host logic)."""
    p = default_params(**params)
    o = [C.c_int() for _ in range(5)]
    _check(lib().faldoi_tvl1_schedule_query(int(npairs), int(w), int(h), C.byref(p), int(compute_units), *[C.byref(v) for v in o]))
    return (bool(o[0].value),) + tuple(v.value for v in o[1:])


def queue_replay(ops, sims):
    """Replays push (1) / pop (0) operations on the candidate queue (host only).  Returns the popped ids."""
    ops = np.ascontiguousarray(ops, np.int32)
    sims = _f32(sims)
    out = np.empty(len(ops), np.int32)
    n = C.c_int(0)
    _check(lib().faldoi_queue_replay(len(ops), ops, sims, out, C.byref(n)))
    return out[:n.value]


def image_to_lab(rgb):
    """(3,h,w) planar RGB 0..255 -> planar Lab (host side of the library, libm)."""
    rgb = _f32(rgb)
    if rgb.ndim != 3 or rgb.shape[0] != 3:
        raise ValueError("image_to_lab takes (3,h,w) planar RGB, got %s" % (rgb.shape,))
    out = np.empty_like(rgb)
    _check(lib().faldoi_image_to_lab(rgb, rgb.shape[2], rgb.shape[1], out))
    return out


class LocalTvl1:
    """The local TV-l2 coupled patch operator (faldoi_local_tvl1): device-resident flow, duals, divergence."""

    def __init__(self, ctx, i0n, i1n):
        i0n = _f32(i0n)
        self.h, self.w = i0n.shape
        i1n = _as(i1n, i0n.shape, "i1n")
        self._ctx = ctx
        self._h = C.c_void_p()
        _check(lib().faldoi_local_tvl1_create(ctx._h, i0n, i1n, self.w, self.h, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().faldoi_local_tvl1_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_flow(self, u):
        _check(lib().faldoi_local_tvl1_set_flow(self._h, _as(u, (2, self.h, self.w), "u")))

    def get_flow(self):
        u = np.empty((2, self.h, self.w), np.float32)
        _check(lib().faldoi_local_tvl1_get_flow(self._h, u))
        return u

    def set_fixed(self, fixed):
        fixed = np.ascontiguousarray(fixed, np.int32)
        if fixed.shape != (self.h, self.w):
            raise ValueError("fixed has shape %s, expected %s" % (fixed.shape, (self.h, self.w)))
        _check(lib().faldoi_local_tvl1_set_fixed(self._h, fixed))

    def interpolate(self, patches):
        patches = np.ascontiguousarray(np.asarray(patches, np.int32).reshape(-1, 4))
        _check(lib().faldoi_local_tvl1_interpolate(self._h, len(patches), patches))

    def estimate(self, patches, max_iter_patch=4):
        """patches: (n,4) int32 rows {ii, ij, ei, ej}.  Returns (energies[n], iterations[n])."""
        patches = np.ascontiguousarray(np.asarray(patches, np.int32).reshape(-1, 4))
        n = len(patches)
        ener = np.empty(n, np.float32)
        iters = np.empty(n, np.int32)
        _check(lib().faldoi_local_tvl1_estimate(self._h, n, patches, int(max_iter_patch), ener, iters))
        return ener, iters
