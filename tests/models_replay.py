"""The window sequences of tests/test_oracle_models_cpu.py as two replays with the same output layout:

  replay_reference -- prepare_stuff + of_estimation of the reference's own compiled objects (oracle/_ref), driven the way
                      add_neighbors drives them; used by tests/golden/make_golden.py to write tests/golden/ref_models.npz
  replay_oracle    -- the CPU restatement on the same inputs; tests/test_oracle_models_golden_cpu.py compares it with the
                      committed vectors, so the as-called pins hold on machines where /root/reference never existed.
"""
import ctypes as C

import numpy as np

CROP = (430, 160, 44, 32)          # x0, y0, w, h of the example pair (tests/golden/sintel_clean_easy_frame_000{2,3}.png)
WINDOWS = 18


def windows(rng, w, h, count):
    """Centres as a growing visits them: corners, frame pixels, overlapping neighbours, interior."""
    pts = [(0, 0), (w - 1, 0), (0, h - 1), (w - 1, h - 1), (1, 1), (w // 2, 0), (0, h // 2), (w - 1, h // 2), (w // 2, h - 1)]
    i, j = w // 2, h // 2
    for _ in range(count):
        i = int(np.clip(i + rng.integers(-3, 4), 0, w - 1))
        j = int(np.clip(j + rng.integers(-3, 4), 0, h - 1))
        pts.append((i, j))
    pts += [(int(rng.integers(0, w)), int(rng.integers(0, h))) for _ in range(count)]
    return pts


def _start(method, d, h, w):
    rng = np.random.default_rng(1000 + 10 * method + d)
    u = (rng.standard_normal((2, h, w)) * 1.2).astype(np.float32)
    return rng, u


def replay_reference(oracle, method, i0, i1):
    h, w = i0.shape[1:]
    op = oracle.RefOp(method, w, h, wr=5)
    r0, r1 = op.prepare(i0, i1)
    out = {"frame0": r0.copy(), "frame1": r1.copy()}
    for d in (0, 1):
        rng, u = _start(method, d, h, w)
        op.set_flow(d, u)
        eners = []
        for k, (i, j) in enumerate(windows(rng, w, h, WINDOWS)):
            eners.append(op.estimate(d, i, j, 1 if k % 3 == 0 else 5))
        out["energy%d" % d] = np.array(eners, np.float32)
        out["flow%d" % d] = op.get_flow(d).copy()
        if method in (0, 1):
            out["div%d" % d] = np.stack([op.plane(d, "div_xi1"), op.plane(d, "div_xi2")]).copy()
        if method in (2, 3, 6, 7) and d == 0:
            sc0, wp0, _ = op.duals(d, 0)
            sc1, _, _ = op.duals(d, 1)
            valid = sc0 != -2.0                                  # slots outside the image keep the reference's marker
            out["dual_valid"] = valid
            out["duals"] = np.concatenate([np.where(valid, sc0, 0), np.where(sc1 != -2.0, sc1, 0)]).astype(np.float32)
            out["nl_weights"] = np.where(valid, wp0, 0).astype(np.float32)
    op.close()
    return out


def replay_oracle(oracle, method, i0, i1):
    h, w = i0.shape[1:]
    n = h * w
    L = oracle.lib()
    o0, o1 = oracle.preprocess_local(i0, i0 if method == 2 else i1)     # -m 2 builds both frames from i0 (src/energy_model.cpp:318-319)
    out = {"frame0": o0, "frame1": o1}
    nl = method in (2, 3, 6, 7)
    lab = np.empty_like(i0)
    L.orc_rgb_to_lab_local(i0, n, lab)
    wp, wt = np.zeros((24, h, w), np.float32), np.zeros((h, w), np.float32)
    if nl:
        L.orc_nltv_table_local(lab, 3, w, h, wp, wt)
    weight = np.zeros(11, np.float32)
    L.orc_gaussian1d_weight(weight, 5)
    for d, (src, dst) in enumerate(((o0, o1), (o1, o0))):
        gx, gy = oracle.centered_gradient(dst)
        rng, u = _start(method, d, h, w)
        state = np.zeros((6, h, w), np.float32)
        duals = np.zeros((48, h, w), np.float32)
        eners = []
        for k, (i, j) in enumerate(windows(rng, w, h, WINDOWS)):
            wr = 1 if k % 3 == 0 else 5
            ii, ij, ei, ej = oracle.patch_index(wr, w, h, i, j)
            it = C.c_int(0)
            iiw, ijw = max(0, -(i - wr)), max(0, -(j - wr))
            if method == 0:
                e = L.orc_local_tvl1_estimate(src, dst, gx, gy, u[0], u[1], state, w, h, ii, ij, ei, ej, 4, C.byref(it))
            elif method == 1:
                e = L.orc_local_tvl1w_estimate(src, dst, gx, gy, u[0], u[1], state, w, h, ii, ij, ei, ej, 4, C.byref(it), weight,
                                               iiw, ijw, wr)
            else:
                e = L.orc_local_nl_estimate(method, src, dst, gx, gy, u[0], u[1], duals.ctypes.data if nl else None,
                                            wp.ctypes.data if nl else None, w, h, ii, ij, ei, ej, 4, weight, iiw, ijw, wr,
                                            C.byref(it))
            eners.append(e)
        out["energy%d" % d] = np.array(eners, np.float32)
        out["flow%d" % d] = u.copy()
        if method in (0, 1):
            out["div%d" % d] = state[4:6].copy()
        if nl and d == 0:
            out["duals"] = duals.copy()
            out["nl_weights"] = wp.copy()
    return out
