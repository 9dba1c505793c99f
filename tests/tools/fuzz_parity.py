"""Randomised parity sweep: the HIP path against the CPU checker over many random sizes / parameters.

Wider than tests/ (which pins chosen cases): every case draws its frame size, crop, seeds and parameters from a seeded
generator, so a failure is reproducible from the printed case number.  Runs on the GPU box:
    python tests/tools/fuzz_parity.py [--family tvl1,growing,tvcsad,nltv] [--cases 200] [--seed 0] [--minutes 10]"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import faldoi_ipol_amd as F  # noqa: E402
import util  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)

ap = argparse.ArgumentParser()
ap.add_argument("--family", default="tvl1,growing,tvcsad,nltv")
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--minutes", type=float, default=10.0)
ap.add_argument("--override", default="", help="growing family: k=v,... applied over the drawn parameters (with --only)")
ap.add_argument("--only", type=int, default=-1, help="run this one case and say what differs")
a = ap.parse_args()
ctx = F.Context()
FULL0, FULL1 = util.sintel_pair()
HARD0 = util.load_rgb_planar(os.path.join(util.GOLDEN, "sintel_final_hard_frame_0002.png"))
HARD1 = util.load_rgb_planar(os.path.join(util.GOLDEN, "sintel_final_hard_frame_0003.png"))


def crop_pair(rng, w, h):
    a0, a1 = (FULL0, FULL1) if rng.random() < 0.6 else (HARD0, HARD1)
    x0, y0 = int(rng.integers(0, 1024 - w)), int(rng.integers(0, 436 - h))
    return (np.ascontiguousarray(a0[:, y0:y0 + h, x0:x0 + w]), np.ascontiguousarray(a1[:, y0:y0 + h, x0:x0 + w])), (x0, y0, w, h)


def case_tvl1(rng):
    w, h = int(rng.integers(3, 230)), int(rng.integers(3, 160))
    if rng.random() < 0.5:
        i0n, i1n, _ = util.synthetic_pair(max(w, 4), max(h, 4), seed=int(rng.integers(1 << 30)), amp=float(rng.uniform(0, 6)))
        i0n, i1n = np.ascontiguousarray(i0n[:h, :w]), np.ascontiguousarray(i1n[:h, :w])
        desc = "synthetic %dx%d" % (w, h)
    else:
        w, h = max(w, 8), max(h, 8)
        (c0, c1), cr = crop_pair(rng, w, h)
        i0n, i1n = O.preprocess_global(c0, c1)
        desc = "crop %s" % (cr,)
    if rng.random() < 0.15:
        # frames that were never normalised, integer-valued: with a zero start flow |grad I|^2 is often an even integer >= 6, which the
        # one-product quotient must not be used with (faldoi_device.hpp: div_once_ok) -- the library repeats the solve in the general form
        i0n, i1n = np.round(i0n * np.float32(255)).astype(np.float32), np.round(i1n * np.float32(255)).astype(np.float32)
        desc += " raw"
    u0 = (rng.normal(0, float(rng.choice([0, 0.5, 3.0])), (2, h, w))).astype(np.float32)
    prm = dict(warps=int(rng.integers(1, 5)), max_inner=int(rng.integers(1, 45)), tol_stop=bool(rng.random() < 0.5),
               lam=float(rng.choice([40.0, 10.0, 0.5, 120.0])), theta=float(rng.choice([0.3, 0.1, 0.8, 0.3, 0.3, 6.0, 2.0])),
               tau=float(rng.choice([0.125, 0.25, 0.05])), tol=float(rng.choice([0.01, 0.001, 0.1])))
    chunk = int(rng.choice([0, 1, 2, 3, 5, 7, 10]))
    xi = (rng.normal(0, 0.3, (4, h, w))).astype(np.float32) if rng.random() < 0.3 else None
    desc += " %s chunk=%d xi=%s" % (prm, chunk, xi is not None)
    want, wit, werr, wxi = O.tvl1_global(i0n, i1n, u0, xi=xi, **prm)
    g = dict(prm)
    g["lambda_"] = g.pop("lam")
    g["tol_stop"] = int(g["tol_stop"])
    got, git, gerr, gxi, _ = ctx.tvl1_global(i0n, i1n, u0, xi=xi if xi is not None else np.zeros((4, h, w), np.float32), chunk=chunk, **g)
    ok = util.bits_equal(got, want) and np.array_equal(git, wit) and util.bits_equal(gerr, werr) and util.bits_equal(gxi, wxi)
    return ok, desc


def seed_flows(rng, a, b, frac, noise):
    out = []
    for src, dst in ((a, b), (b, a)):
        u = O.tvl1_global(src, dst, np.zeros((2,) + src.shape, np.float32), warps=2, max_inner=30)[0]
        s = np.full_like(u, np.nan)
        m = rng.random(src.shape) < frac
        s[:, m] = np.round(u[:, m] + rng.normal(0, noise, u[:, m].shape))
        out.append(s)
    return out


def case_growing(rng):
    w, h = int(rng.integers(12, 110)), int(rng.integers(12, 80))
    (c0, c1), cr = crop_pair(rng, w, h)
    i0n, i1n = O.preprocess_local(c0, c1)
    go, ba = seed_flows(rng, i0n, i1n, float(rng.choice([0.003, 0.01, 0.03, 0.1])), float(rng.choice([0, 0, 1.5])))
    prm = dict(wr=int(rng.integers(1, 6)), loc_it=int(rng.integers(0, 4)), max_iter_patch=int(rng.integers(1, 7)),
               fb_thresh=float(rng.choice([2.0, 0.5, 0.13, 5.0])), method=int(rng.integers(0, 2)))
    split = {}
    sal = (None, None)
    r = rng.random()
    if r < 0.4 and w >= 40 and h >= 40:
        split = dict(split_img=int(rng.integers(1, 3)), h_parts=int(rng.integers(1, 4)), v_parts=int(rng.integers(1, 4)))
        if split["split_img"] == 2 and rng.random() < 0.5:
            split.update(first_h_parts=int(rng.integers(1, 3)), first_v_parts=int(rng.integers(1, 3)))
    elif r < 0.6:
        sal = tuple(rng.uniform(0.2, 1.0, (h, w)).astype(np.float32) for _ in range(2))
    budget = int(rng.choice([16384, 257, 5000]))
    for kv in a.override.split(","):
        if kv:
            k, v = kv.split("=")
            if k == "budget":
                budget = int(v)
            elif k in ("split_img", "h_parts", "v_parts", "first_h_parts", "first_v_parts"):
                split[k] = int(v)
            else:
                prm[k] = float(v) if k == "fb_thresh" else int(v)
    desc = "crop %s %s %s sal=%s seeds=%d/%d budget=%d" % (cr, prm, split, sal[0] is not None, np.isfinite(go[0]).sum(), np.isfinite(ba[0]).sum(), budget)
    want, wene, st = O.local_faldoi(i0n, i1n, go, ba, sal0=sal[0], sal1=sal[1], **prm, **split)
    got, gene, gst = ctx.match_growing(i0n, i1n, go, ba, sal[0], sal[1], preprocessed=1, budget=budget, **prm, **split)
    ok = (util.bits_equal(got, want) and util.bits_equal(gene, wene) and list(gst.solves) == [st[1], st[4]]
          and list(gst.pops) == [st[0], st[3]] and list(gst.fills) == [st[2], st[5]])
    if not ok:
        desc += " | flow: %s | energy: %s | pops/windows/fills device %s checker %s" % (
            util.mismatch_report(got, want), util.mismatch_report(gene, wene),
            [list(gst.pops), list(gst.solves), list(gst.fills)], [int(v) for v in st])
        desc += " | device kept per iteration (fwd, bwd): %s" % [round(v * w * h) for v in gst.fb_kept[:2 * prm["loc_it"]]]
    return ok, desc


def case_tvcsad(rng):
    w, h = int(rng.integers(10, 120)), int(rng.integers(10, 90))
    (c0, c1), cr = crop_pair(rng, w, h)
    i0n, i1n = O.preprocess_global(c0, c1)
    u0 = (rng.normal(0, float(rng.choice([0, 0.5])), (2, h, w))).astype(np.float32)
    prm = dict(warps=int(rng.integers(1, 4)), max_inner=int(rng.integers(1, 60)), tol=float(rng.choice([0.01, 0.002, 0.05])),
               lam=float(rng.choice([0.85, 0.3, 3.0])), theta=0.3, tau=0.125)
    desc = "crop %s %s" % (cr, prm)
    want, wit, werr, wxi = O.tvcsad_global(i0n, i1n, u0, **prm)
    g = dict(prm)
    g["lambda_"] = g.pop("lam")
    res = ctx.tvcsad_global(i0n, i1n, u0, **g)
    got, git = res[0], res[1]
    return util.bits_equal(got, want) and np.array_equal(git, wit), desc


def case_nltv(rng):
    # (sizes up to five tile columns of the fused kernel and several tile rows; every tile height the host can pick)
    w, h = (int(rng.integers(8, 90)), int(rng.integers(8, 70))) if rng.random() < 0.5 else (int(rng.integers(60, 330)), int(rng.integers(12, 120)))
    rows = rng.choice(["", "8", "12", "14", "16"])
    if rows:
        os.environ["FALDOI_NL_ROWS"] = str(rows)
    else:
        os.environ.pop("FALDOI_NL_ROWS", None)
    (c0, c1), cr = crop_pair(rng, w, h)
    i0n, i1n = O.preprocess_global(c0, c1)
    lab = O.image_to_lab(c0)
    u0 = (rng.normal(0, float(rng.choice([0, 0.5])), (2, h, w))).astype(np.float32)
    prm = dict(warps=int(rng.integers(1, 3)), max_inner=int(rng.integers(1, 40 if w * h < 8000 else 10)))
    l1 = rng.random() < 0.4
    desc = "crop %s %s %s rows %s" % (cr, prm, "NLTV-L1" if l1 else "NLTV-CSAD", rows or "auto")
    if l1:
        want = O.nltvl1_global(i0n, i1n, lab, u0, **prm)
        got = ctx.nltvl1_global(i0n, i1n, lab, u0, **prm)
    else:
        want = O.nltvcsad_global(i0n, i1n, lab, u0, **prm)
        got = ctx.nltvcsad_global(i0n, i1n, lab, u0, **prm)
    want = want[0] if isinstance(want, tuple) else want
    got = got[0] if isinstance(got, tuple) else got
    return util.bits_equal(got, want), desc


def same_up_to_nan_payload(x, y):
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    nx_, ny_ = np.isnan(x), np.isnan(y)
    return x.shape == y.shape and np.array_equal(nx_, ny_) and np.array_equal(x[~nx_].view(np.uint32), y[~ny_].view(np.uint32))


def case_growing_edge(rng):
    """Seeds the matcher would never give: none at all, everywhere, far outside the frame, fractional, one direction only."""
    w, h = int(rng.integers(6, 70)), int(rng.integers(6, 50))
    (c0, c1), cr = crop_pair(rng, w, h)
    i0n, i1n = O.preprocess_local(c0, c1)
    kind = int(rng.integers(0, 6))
    go, ba = np.full((2, h, w), np.nan, np.float32), np.full((2, h, w), np.nan, np.float32)
    if kind == 1:                                   # every pixel a seed
        go = rng.integers(-3, 4, (2, h, w)).astype(np.float32)
        ba = rng.integers(-3, 4, (2, h, w)).astype(np.float32)
    elif kind == 2:                                 # flows that leave the frame
        for s in (go, ba):
            m = rng.random((h, w)) < 0.05
            s[:, m] = rng.integers(-200, 200, (2, int(m.sum()))).astype(np.float32)
    elif kind == 3:                                 # fractional seeds, forward only
        m = rng.random((h, w)) < 0.05
        go[:, m] = rng.normal(0, 2, (2, int(m.sum()))).astype(np.float32)
    elif kind == 4:                                 # a single seed per direction
        go[:, h // 2, w // 2] = 0
        ba[:, 0, 0] = 1
    elif kind == 5:                                 # one component finite, the other NaN: not a seed (:770)
        m = rng.random((h, w)) < 0.05
        go[0, m] = 1.0
        ba[:, m] = 0.0
    prm = dict(wr=int(rng.integers(1, 6)), loc_it=int(rng.integers(0, 4)), max_iter_patch=int(rng.integers(1, 9)),
               fb_thresh=float(rng.choice([2.0, 0.05, 50.0])), method=int(rng.integers(0, 2)))
    desc = "crop %s kind %d %s" % (cr, kind, prm)
    want, wene, st = O.local_faldoi(i0n, i1n, go, ba, **prm)
    got, gene, gst = ctx.match_growing(i0n, i1n, go, ba, preprocessed=1, **prm)
    ok = (same_up_to_nan_payload(got, want) and same_up_to_nan_payload(gene, wene) and list(gst.solves) == [st[1], st[4]]
          and list(gst.pops) == [st[0], st[3]] and list(gst.fills) == [st[2], st[5]])
    if not ok:
        desc += " | pops/windows/fills device %s checker %s" % ([list(gst.pops), list(gst.solves), list(gst.fills)], [int(v) for v in st])
    return ok, desc


def case_batch(rng):
    """Several pairs in one call (RGB frames, device preprocessing), default and register-capped builds."""
    w, h = int(rng.integers(16, 64)), int(rng.integers(16, 48))
    k = int(rng.integers(2, 7))
    i0s, i1s, gos, bas, wants = [], [], [], [], []
    prm = dict(wr=int(rng.integers(2, 6)), loc_it=int(rng.integers(1, 4)), method=int(rng.integers(0, 2)))
    for _ in range(k):
        (c0, c1), _cr = crop_pair(rng, w, h)
        a_, b_ = O.preprocess_local(c0, c1)
        go, ba = seed_flows(rng, a_, b_, 0.03, 0)
        i0s.append(c0); i1s.append(c1); gos.append(go); bas.append(ba)
        wants.append(O.local_faldoi(a_, b_, go, ba, **prm)[0])
    dense = rng.random() < 0.5
    os.environ["FALDOI_GROW_DENSE"] = "1" if dense else "0"
    os.environ["FALDOI_RESEED"] = "device" if rng.random() < 0.5 else "host"
    outs, _, _ = ctx.match_growing_batch(i0s, i1s, gos, bas, **prm)
    os.environ["FALDOI_GROW_DENSE"] = "0"
    os.environ.pop("FALDOI_RESEED")
    return all(util.bits_equal(o, w_) for o, w_ in zip(outs, wants)), "%d pairs %dx%d %s dense=%s" % (k, w, h, prm, dense)


def case_batch_split(rng):
    """Several pairs, each on its own image partitions; some pairs fall back to the whole image (no backward seed)."""
    w, h = int(rng.integers(40, 100)), int(rng.integers(40, 80))
    k = int(rng.integers(2, 6))
    prm = dict(wr=int(rng.integers(2, 6)), loc_it=int(rng.integers(1, 4)), method=int(rng.integers(0, 2)),
               fb_thresh=float(rng.choice([2.0, 0.3])))
    split = dict(split_img=int(rng.integers(1, 3)), h_parts=int(rng.integers(1, 4)), v_parts=int(rng.integers(1, 4)))
    i0s, i1s, gos, bas, wants = [], [], [], [], []
    for q in range(k):
        (c0, c1), _cr = crop_pair(rng, w, h)
        a_, b_ = O.preprocess_local(c0, c1)
        go, ba = seed_flows(rng, a_, b_, float(rng.choice([0.01, 0.04])), 0)
        if rng.random() < 0.25:
            ba = np.full_like(ba, np.nan)          # this pair's partitions never all get a candidate: whole-image fall-back
        i0s.append(a_); i1s.append(b_); gos.append(go); bas.append(ba)
        wants.append(O.local_faldoi(a_, b_, go, ba, **prm, **split))
    outs, enes, gst = ctx.match_growing_batch(i0s, i1s, gos, bas, preprocessed=1, **prm, **split)
    ok = all(util.bits_equal(o, w_[0]) and util.bits_equal(e, w_[1]) for o, e, w_ in zip(outs, enes, wants))
    ok = ok and list(gst.solves) == [sum(int(w_[2][1]) for w_ in wants), sum(int(w_[2][4]) for w_ in wants)]
    return ok, "%d pairs %dx%d %s %s" % (k, w, h, prm, split)


def case_prims(rng):
    """Stencils, warps, preprocessing, resampling and the pruning test on random sizes and values."""
    w, h = int(rng.integers(2, 150)), int(rng.integers(2, 110))
    img = rng.uniform(0, 255, (h, w)).astype(np.float32)
    v1 = rng.normal(0, 1, (h, w)).astype(np.float32)
    v2 = rng.normal(0, 1, (h, w)).astype(np.float32)
    L = O.lib()
    bad = []
    dx, dy = O.centered_gradient(img)
    gx, gy = ctx.centered_gradient(img)
    if not (util.bits_equal(gx, dx) and util.bits_equal(gy, dy)):
        bad.append("centered_gradient")
    fx, fy = np.empty_like(img), np.empty_like(img)
    L.orc_forward_gradient(img, fx, fy, w, h)
    if not all(util.bits_equal(x, y) for x, y in zip(ctx.forward_gradient(img), (fx, fy))):
        bad.append("forward_gradient")
    dv = np.empty_like(img)
    L.orc_divergence(v1, v2, dv, w, h)
    if not util.bits_equal(ctx.divergence(v1, v2), dv):
        bad.append("divergence")
    scale = float(rng.choice([0.5, 3.0, 40.0, 400.0]))
    uu = (rng.normal(0, scale, (h, w))).astype(np.float32)
    vv = (rng.normal(0, scale, (h, w))).astype(np.float32)
    if rng.random() < 0.3:
        uu[rng.random((h, w)) < 0.02] = np.nan
    for bo in (True, False):
        wo = np.empty_like(img)
        L.orc_bicubic_warp(img, uu, vv, wo, w, h, int(bo))
        if not same_up_to_nan_payload(ctx.bicubic_warp(img, uu, vv, bo), wo):
            bad.append("bicubic_warp border_out=%s" % bo)
    if w >= 6 and h >= 6:
        f0 = np.stack([uu, vv]) * np.float32(0.05)
        f1 = -np.stack([vv, uu]) * np.float32(0.05)
        eps = float(rng.choice([2.0, 0.3, 0.01]))
        tr = np.empty((h, w), np.int32)
        L.orc_fb_consistency_check(f0, f1, tr, w, h, eps)
        if not np.array_equal(ctx.fb_consistency_check(f0, f1, eps), tr):
            bad.append("fb_consistency_check")
    if w >= 8 and h >= 8:
        pd = int(rng.choice([1, 3]))
        c0 = rng.uniform(0, 255, (pd, h, w)).astype(np.float32)
        c1 = rng.uniform(0, 255, (pd, h, w)).astype(np.float32)
        if not all(util.bits_equal(x, y) for x, y in zip(ctx.preprocess_global(c0, c1), O.preprocess_global(c0, c1))):
            bad.append("preprocess_global")
        if not all(util.bits_equal(x, y) for x, y in zip(ctx.preprocess_local(c0, c1), O.preprocess_local(c0, c1))):
            bad.append("preprocess_local")
        fac = float(rng.choice([0.5, 0.75, 0.8]))
        zo = O.zoom_out(img, fac)
        if not util.bits_equal(ctx.zoom_out(img, fac), zo):
            bad.append("zoom_out %g" % fac)
        if not util.bits_equal(ctx.zoom_in(zo, w, h), O.zoom_in(zo, w, h)):
            bad.append("zoom_in")
    return not bad, "%dx%d %s" % (w, h, bad)


def case_tvl1_nan(rng):
    """NaN or absurdly large (finite) entries in the initial flow, NaN in a frame: same NaN set, same finite bits.
    (Infinities and magnitudes near FLT_MAX are outside the range of the exact-quotient sequences)"""
    w, h = int(rng.integers(8, 120)), int(rng.integers(8, 90))
    i0n, i1n, gt = util.synthetic_pair(w, h, seed=int(rng.integers(1 << 30)), amp=2.0)
    u0 = gt.copy()
    for _ in range(int(rng.integers(1, 4))):
        u0[int(rng.integers(0, 2)), int(rng.integers(0, h)), int(rng.integers(0, w))] = float(rng.choice([np.nan, np.nan, 1e4, -3e5]))
    if rng.random() < 0.3:
        i1n = i1n.copy()
        i1n[int(rng.integers(0, h)), int(rng.integers(0, w))] = np.nan
    prm = dict(warps=int(rng.integers(1, 4)), max_inner=int(rng.integers(1, 25)), tol_stop=bool(rng.random() < 0.5))
    chunk = int(rng.choice([0, 1, 4, 10]))
    want, wit, _, _ = O.tvl1_global(i0n, i1n, u0, **prm)
    got, git, _, _, _ = ctx.tvl1_global(i0n, i1n, u0, warps=prm["warps"], max_inner=prm["max_inner"], tol_stop=int(prm["tol_stop"]), chunk=chunk)
    ok = same_up_to_nan_payload(got, want) and np.array_equal(git, wit)
    desc = "%dx%d %s chunk=%d nan=%d" % (w, h, prm, chunk, np.isnan(want).sum())
    if not ok:
        bad = np.argwhere((np.isnan(got) != np.isnan(want)) | (~np.isnan(want) & (got.view(np.uint32) != want.view(np.uint32))))
        desc += " | injected %s | iters %s vs %s | NaN device %d | %d entries differ, first %s: device %r checker %r" % (
            [(tuple(ix), float(u0[tuple(ix)])) for ix in np.argwhere(~np.isfinite(u0) | (np.abs(u0) > 1e20))], list(git), list(wit),
            np.isnan(got).sum(), len(bad), bad[0] if len(bad) else None,
            float(got[tuple(bad[0])]) if len(bad) else None, float(want[tuple(bad[0])]) if len(bad) else None)
    return ok, desc


def case_tvl1_batch(rng):
    """Several pairs of one size resident at once: every pair stops on its own (device-side loop control), the batch
    is solved twice (each solve restarts from the uploaded state), FALDOI_TILE picks the tile shape."""
    w, h = int(rng.integers(8, 260)), int(rng.integers(8, 180))
    k = int(rng.integers(2, 7))
    prm = dict(warps=int(rng.integers(1, 5)), max_inner=int(rng.integers(1, 60)), tol_stop=bool(rng.random() < 0.7),
               tol=float(rng.choice([0.01, 0.05, 0.002])))
    chunk = int(rng.choice([0, 1, 3, 5, 10]))
    tile = str(rng.choice(["", "4,16", "3,12", "4,12", "3,16", "4,8", "3,8", "2,16", "8,8"]))
    if tile:
        os.environ["FALDOI_TILE"] = tile
    else:
        os.environ.pop("FALDOI_TILE", None)
    batch = F.Tvl1Batch(ctx, k, w, h)
    ok, frames, state, notes = True, [], [], []
    try:
        for i in range(k):
            i0n, i1n, gt = util.synthetic_pair(w, h, seed=int(rng.integers(1 << 30)), amp=float(rng.uniform(0, 5)))
            u0 = (gt * np.float32(rng.choice([0.0, 1.0])) + rng.normal(0, 0.2, gt.shape)).astype(np.float32)
            batch.upload(i, i0n, i1n, u0)
            frames.append((i0n, i1n))
            state.append((u0, None))
        for rep in range(2):                           # every solve starts from the uploaded state (include/faldoi_hip.h)
            iters, errs, _, _ = batch.solve(warps=prm["warps"], max_inner=prm["max_inner"], tol_stop=int(prm["tol_stop"]),
                                           tol=prm["tol"], chunk=chunk)
            for i in range(k):
                want, wit, werr, wxi = O.tvl1_global(frames[i][0], frames[i][1], state[i][0], xi=state[i][1], **prm)
                got, gxi = batch.download(i, want_xi=True)
                parts = (util.bits_equal(got, want), np.array_equal(iters[i], wit), util.bits_equal(errs[i], werr), util.bits_equal(gxi, wxi))
                if not all(parts):
                    ok = False
                    notes.append("solve %d pair %d flow/iters/err/duals %s iters %s vs %s" % (rep, i, parts, list(iters[i]), list(wit)))
    finally:
        batch.close()
        os.environ.pop("FALDOI_TILE", None)
    return ok, "%d pairs %dx%d %s chunk=%d tile=%s %s" % (k, w, h, prm, chunk, tile or "auto", notes[:3])


def case_growing_big(rng):
    """Larger frames: deep queues, many relaunches, weighted functional, partitions."""
    w, h = int(rng.integers(120, 320)), int(rng.integers(90, 220))
    (c0, c1), cr = crop_pair(rng, w, h)
    i0n, i1n = O.preprocess_local(c0, c1)
    go, ba = seed_flows(rng, i0n, i1n, float(rng.choice([0.002, 0.01])), 0)
    prm = dict(wr=int(rng.integers(3, 6)), loc_it=int(rng.integers(1, 4)), method=int(rng.integers(0, 2)))
    split = {}
    if rng.random() < 0.5:
        split = dict(split_img=int(rng.integers(1, 3)), h_parts=int(rng.integers(2, 6)), v_parts=int(rng.integers(2, 5)))
    budget = int(rng.choice([16384, 1000, 40000]))
    want, wene, st = O.local_faldoi(i0n, i1n, go, ba, **prm, **split)
    got, gene, gst = ctx.match_growing(i0n, i1n, go, ba, preprocessed=1, budget=budget, **prm, **split)
    ok = (util.bits_equal(got, want) and util.bits_equal(gene, wene) and list(gst.solves) == [st[1], st[4]]
          and list(gst.pops) == [st[0], st[3]] and list(gst.fills) == [st[2], st[5]])
    return ok, "crop %s %s %s budget=%d windows=%d" % (cr, prm, split, budget, st[1] + st[4])


def case_growing_m(rng):
    """The other local functionals (-m 2..7): random crops (sometimes with frame pixels of either example pair), radii,
    iteration counts, seed densities; raw RGB frames (the Lab weights need them)."""
    method = int(rng.integers(2, 8))
    big = method < 4                                  # the CPU checker's CSAD step sorts 97 floats per pixel-iteration
    w, h = int(rng.integers(12, 72 if big else 44)), int(rng.integers(12, 56 if big else 34))
    (c0, c1), cr = crop_pair(rng, w, h)
    a_, b_ = O.preprocess_local(c0, c1)
    go, ba = seed_flows(rng, a_, b_, float(rng.choice([0.01, 0.03, 0.1])), float(rng.choice([0, 0, 1.5])))
    prm = dict(wr=int(rng.integers(1, 6)), loc_it=int(rng.integers(0, 3)), max_iter_patch=int(rng.integers(1, 6)),
               fb_thresh=float(rng.choice([2.0, 0.5, 5.0])))
    budget = int(rng.choice([16384, 257, 3000]))
    desc = "m=%d crop %s %s seeds=%d/%d budget=%d" % (method, cr, prm, np.isfinite(go[0]).sum(), np.isfinite(ba[0]).sum(), budget)
    want, wene, st = O.local_faldoi_raw(c0, c1, go, ba, method, **prm)
    got, gene, gst = ctx.match_growing(c0, c1, go, ba, budget=budget, method=method, **prm)
    ok = (util.bits_equal(got, want) and util.bits_equal(gene, wene) and list(gst.solves) == [st[1], st[4]]
          and list(gst.pops) == [st[0], st[3]] and list(gst.fills) == [st[2], st[5]])
    if not ok:
        desc += " | flow: %s | energy: %s | counts device %s checker %s" % (
            util.mismatch_report(got, want), util.mismatch_report(gene, wene),
            [list(gst.pops), list(gst.solves), list(gst.fills)], [int(v) for v in st])
    return ok, desc


def case_persist(rng):
    """A static schedule of one pair on the persistent kernel (random tile geometry, ragged last tiles, many tiles) against
    the launch-per-chunk path: same flow, duals, errors bit for bit; where a NaN was injected, the same NaN set (a NaN's payload
    is not pinned anywhere: the two kernels are compiled separately and may commute the operands of an addition)."""
    w, h = int(rng.integers(2, 1100)), int(rng.integers(2, 480))
    if rng.random() < 0.4:
        w, h = int(rng.integers(2, 200)), int(rng.integers(2, 140))
    i0n, i1n, gt = util.synthetic_pair(max(w, 4), max(h, 4), seed=int(rng.integers(1 << 30)), amp=float(rng.uniform(0, 8)))
    i0n, i1n, gt = np.ascontiguousarray(i0n[:h, :w]), np.ascontiguousarray(i1n[:h, :w]), np.ascontiguousarray(gt[:, :h, :w])
    u0 = (gt * float(rng.choice([0.0, 0.5, 1.0])) + rng.normal(0, float(rng.choice([0, 0.5, 20.0])), (2, h, w))).astype(np.float32)
    if rng.random() < 0.15:
        u0[int(rng.integers(0, 2)), int(rng.integers(0, h)), int(rng.integers(0, w))] = np.nan
    xi = (rng.normal(0, float(rng.choice([0.3, 2.0])), (4, h, w))).astype(np.float32) if rng.random() < 0.5 else None
    prm = dict(warps=int(rng.integers(1, 6)), max_inner=int(rng.integers(1, 24)), tol_stop=0,
               lambda_=float(rng.choice([40.0, 10.0, 0.5, 120.0])), theta=float(rng.choice([0.3, 0.1, 0.8])),
               tau=float(rng.choice([0.125, 0.25, 0.05])), chunk=int(rng.choice([0, 0, 0, 1, 2, 3, 4, 6, 8])))
    cfg = str(rng.choice(["", "", "3,16,5", "4,13,5", "4,11,4", "4,16,8", "3,14,3", "3,16,2", "4,16,4"]))
    if rng.random() < 0.5:                            # the tolerance stop, decided by the resident tiles (64x48 tiles only)
        prm.update(tol_stop=1, tol=float(rng.choice([0.3, 0.1, 0.03, 0.01, 0.001])), max_inner=int(rng.integers(1, 90)))
        cfg = str(rng.choice(["", "", "3,16,5", "3,16,2", "3,16,7", "3,16,4"]))
    res = []
    for persist in ("0", "1"):
        os.environ["FALDOI_PERSIST"] = persist
        if cfg and persist == "1":
            os.environ["FALDOI_PERSIST_CFG"] = cfg
        else:
            os.environ.pop("FALDOI_PERSIST_CFG", None)
        b = F.Tvl1Batch(ctx, 1, w, h)
        try:
            b.upload(0, i0n, i1n, u0, xi)
            its, errs, _, launches = b.solve(**prm)
            u, x = b.download(0, want_xi=True)
            res.append((u, x, its.copy(), errs.copy(), launches, b.last_persistent()))
        finally:
            b.close()
    os.environ.pop("FALDOI_PERSIST", None)
    os.environ.pop("FALDOI_PERSIST_CFG", None)
    r, n = res
    ok = same_up_to_nan_payload(r[0], n[0]) and same_up_to_nan_payload(r[1], n[1]) and np.array_equal(r[2], n[2]) and \
        same_up_to_nan_payload(r[3], n[3]) and not r[5]
    desc = "%dx%d cfg=%r persistent=%s launches %d->%d iters %s %s xi=%s" % (w, h, cfg, n[5], r[4], n[4], list(n[2][0]), prm, xi is not None)
    return ok, desc


FAMILIES = {"persist": case_persist, "growing_m": case_growing_m, "tvl1_batch": case_tvl1_batch, "growing_big": case_growing_big, "tvl1": case_tvl1, "growing": case_growing, "tvcsad": case_tvcsad, "nltv": case_nltv,
            "growing_edge": case_growing_edge, "batch": case_batch, "batch_split": case_batch_split, "prims": case_prims, "tvl1_nan": case_tvl1_nan}
fams = [f for f in a.family.split(",") if f]
t_end = time.time() + a.minutes * 60
bad = 0
refused = {}
count = {f: 0 for f in fams}
for n in (range(a.cases) if a.only < 0 else [a.only]):
    if time.time() > t_end:
        break
    fam = fams[n % len(fams)]
    rng = np.random.default_rng([a.seed, n])
    try:
        ok, desc = FAMILIES[fam](rng)
    except F.FaldoiError as e:                       # a refused configuration is fine if it says why
        ok, desc = True, "refused: %s" % e
        refused[str(e)[:90]] = refused.get(str(e)[:90], 0) + 1
    except Exception:
        ok, desc = False, traceback.format_exc()
    count[fam] += 1
    if not ok:
        bad += 1
        print("MISMATCH case %d (%s, --seed %d): %s" % (n, fam, a.seed, desc), flush=True)
    elif n % 20 == 0 or a.only >= 0:
        print("case %d ok (%s): %s" % (n, fam, desc[:150]), flush=True)
print("done: %s cases, %d mismatches; refused configurations: %s" % (count, bad, refused), flush=True)
sys.exit(1 if bad else 0)
