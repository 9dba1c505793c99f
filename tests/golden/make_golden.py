"""Regenerates tests/golden/ref_primitives.npz and tests/golden/ref_models.npz from the REFERENCE's own compiled objects.

Run in the build container only (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_golden.py
The arrays are data (seeded inputs + what the reference objects returned for them); no reference
source text is stored.  The GPU box and CPU-only CI read the .npz and never need the reference.

Also committed next to it:
  sintel_clean_easy_frame_000{2,3}.png  -- the reference's example_data/clean/easy frames (inputs
                                           of BASELINE configs 1/2), copied verbatim as data files.
  sintel_final_hard_frame_000{2,3}.png  -- example_data/final/hard frames (a large-motion pair for the
                                           GPU == oracle test; no reference output recorded for it).
  sintel_{clean,final}_{easy,medium,hard}_frame_000{1..4}.png -- (round 4) all six example sequences, frames 1-4: the 18 frame pairs
                                           of SURVEY 8d's configs[3] input (example_frames() below; data files)
  known_answers.json                    -- what the compiled reference produced for whole solves
                                           (iteration counts, sha256 of the .flo), SURVEY.md App. D.
  sintel_{clean_easy,final_hard}_gt_frame_0002.npz -- the ground-truth flows the reference ships beside those frames
                                           (example_data/*/*/gt/frame_0002.flo; data files, re-packed planar (2,h,w) by
                                           ground_truth() below): the accuracy anchors of the suite (EPE vs GT).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402


def main():
    R = O.ref()
    if R is None:
        raise SystemExit("oracle/_ref/libfaldoi_ref.so missing: run `make -C oracle ref` first")
    rng = np.random.default_rng(20261003)
    h, w = 23, 29
    out = {}
    img = (rng.random((h, w)) * 255).astype(np.float32)
    img2 = (rng.random((h, w)) * 255).astype(np.float32)
    img3 = (rng.random((h, w)) * 200 + 7).astype(np.float32)
    out["img"], out["img2"], out["img3"] = img, img2, img3

    dx, dy = np.empty_like(img), np.empty_like(img)
    R.ref_centered_gradient(img, dx, dy, w, h)
    out["cg_dx"], out["cg_dy"] = dx.copy(), dy.copy()
    R.ref_forward_gradient(img, dx, dy, w, h)
    out["fg_fx"], out["fg_fy"] = dx.copy(), dy.copy()
    R.ref_divergence(img, img2, dx, w, h)
    out["div"] = dx.copy()

    g = img.copy()
    R.ref_gaussian(g, w, h, np.float32(0.9))
    out["gauss09"] = g

    a, b, c = np.empty_like(img), np.empty_like(img), np.empty_like(img)
    R.ref_image_normalization_3(img, img2, img3, a, b, c, h * w)
    out["norm3_a"], out["norm3_b"], out["norm3_c"] = a.copy(), b.copy(), c.copy()
    R.ref_image_normalization(img, img2, a, b, h * w)
    out["norm2_a"], out["norm2_b"] = a.copy(), b.copy()

    # flow with small, large, negative and far-out-of-range displacements
    u = (rng.standard_normal((h, w)) * 3).astype(np.float32)
    v = (rng.standard_normal((h, w)) * 3).astype(np.float32)
    u[0, :5] = [-0.5, -1.5, 0.25, 100.0, -100.0]
    v[1, :5] = [-0.5, -1.5, 0.25, 100.0, -100.0]
    u[5, 5], v[5, 5] = 0.0, 0.0
    u[6, 6], v[6, 6] = 1.0, -1.0
    out["u"], out["v"] = u, v
    for bo in (0, 1):
        o = np.empty_like(img)
        R.ref_bicubic_warp(img, u, v, o, w, h, bo)
        out["warp_bo%d" % bo] = o
    # point samples, incl. the (-1,0) band and the sx-for-sy quirk region
    pts = np.array([[0.0, 0.0], [-0.25, 3.0], [3.0, -0.25], [-0.25, -0.25], [1.5, 1.5],
                    [w - 1.5, h - 1.5], [w - 2.01, 2.0], [w + 3.0, 2.0], [2.0, h + 3.0],
                    [-3.7, 4.2], [4.2, -3.7], [0.999, 0.999], [w - 3.0, h - 3.0],
                    [w - 2.999, h - 2.999], [7.25, 0.5], [0.5, 7.25]], np.float32)
    out["pts"] = pts
    for bo in (0, 1):
        out["at_bo%d" % bo] = np.array(
            [R.ref_bicubic_at(img, float(p[0]), float(p[1]), w, h, bo) for p in pts], np.float32)
    # src/zoom.c (compiled from the reference with mask.c): down- and up-sampling
    import ctypes as C
    for tag, f in (("050", 0.5), ("075", 0.75)):
        ww, hh = C.c_int(), C.c_int()
        R.ref_zoom_size(w, h, C.byref(ww), C.byref(hh), f)
        zo = np.empty((hh.value, ww.value), np.float32)
        R.ref_zoom_out(img, zo, w, h, f)
        zi = np.empty((h, w), np.float32)
        R.ref_zoom_in(zo, zi, ww.value, hh.value, w, h)
        out["zoom_out_" + tag], out["zoom_in_" + tag] = zo, zi
    # src/elap_recsep.c (harmonic window filling): windows with holes -> filled windows
    for tag, (eh, ew, frac) in (("a", (11, 11, 0.4)), ("b", (11, 11, 0.93)), ("c", (7, 11, 0.6)), ("d", (3, 3, 0.5)), ("e", (2, 2, 0.5))):
        x = (rng.standard_normal((2, eh, ew)) * 3).astype(np.float32)
        x[:, rng.random((eh, ew)) < frac] = np.nan
        o = np.empty_like(x)
        R.ref_elap_recursive_separable(o, x.copy(), ew, eh, 2, 0.4, 3, 7)
        out["elap_in_" + tag], out["elap_out_" + tag] = x, o
    # local window operators on a sequence of overlapping windows (state carries over): plain and weighted
    lh, lw = 30, 36
    yy, xx = np.mgrid[0:lh, 0:lw].astype(np.float64)
    l0 = (0.5 + 0.25 * np.sin(xx / 3.1) * np.cos(yy / 4.3) + 0.2 * np.sin((xx + 2 * yy) / 7.0)).astype(np.float32)
    l1 = (0.5 + 0.25 * np.sin((xx - 0.8) / 3.1) * np.cos((yy + 0.5) / 4.3) + 0.2 * np.sin((xx - 0.8 + 2 * yy + 1.0) / 7.0)).astype(np.float32)
    l1x, l1y = np.empty_like(l1), np.empty_like(l1)
    R.ref_centered_gradient(l1, l1x, l1y, lw, lh)
    wins = np.array([[5, 5, 5], [0, 0, 5], [lw - 1, lh - 1, 5], [20, 17, 5], [22, 18, 5], [1, lh - 2, 1], [0, 0, 1], [30, 2, 1],
                     [lw - 1, 12, 5], [16, 12, 5], [17, 13, 5]], np.int32)            # centre column, row, radius
    out["loc_i0"], out["loc_i1"], out["loc_wins"] = l0, l1, wins
    u0 = (rng.standard_normal((2, lh, lw)) * 0.7).astype(np.float32)
    out["loc_u0"] = u0
    for tag in ("plain", "weighted"):
        u = u0.copy()
        ref = O.ref_weighted_operator(lw, lh, 5) if tag == "weighted" else R.ref_local_tvl1_create(lw, lh)
        eners = []
        for ci, cj, wc in wins:
            ii, ij, ei, ej = O.patch_index(int(wc), lw, lh, int(ci), int(cj))
            if tag == "weighted":
                e = R.ref_local_tvl1w_estimate(ref, l0, l1, l1x, l1y, u[0], u[1], ii, ij, ei, ej, 4, max(0, int(wc - ci)),
                                               max(0, int(wc - cj)), int(wc))
            else:
                e = R.ref_local_tvl1_estimate(ref, l0, l1, l1x, l1y, u[0], u[1], ii, ij, ei, ej, 4)
            eners.append(e)
        (R.ref_local_tvl1w_destroy if tag == "weighted" else R.ref_local_tvl1_destroy)(ref)
        out["loc_u_" + tag], out["loc_e_" + tag] = u, np.array(eners, np.float32)
    for r in (1, 3, 5):
        wv = np.zeros(2 * r + 1, np.float32)
        R.ref_gaussian1d_weight_c(wv, r)            # src/mask.c:360, the copy prepare_stuff binds
        out["gw_%d" % r] = wv
    np.savez_compressed(os.path.join(HERE, "ref_primitives.npz"), **out)
    print("wrote ref_primitives.npz with", len(out), "arrays")


def partition_case(P, rng_seed, w, h):
    """The same seeded candidate pushes (heavy ties) into a Parts object, then the image -> partitions transfer."""
    rng = np.random.default_rng(rng_seed)
    for d in (0, 1):
        for _ in range(300):
            i, j = int(rng.integers(0, w)), int(rng.integers(0, h))
            sim = float(rng.integers(0, 6)) * 0.25
            u, v = float(rng.standard_normal()), float(rng.standard_normal())
            P.push(d, i, j, u, v, sim)
    P.transfer(True)


def models():
    """tests/golden/ref_models.npz: prepare_stuff + of_estimation of the reference AS CALLED for every functional (-m 0..7) on
    a crop of the example pair, the Lab image, and the partition split of a candidate queue (tests/models_replay.py)."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    import models_replay as MR
    from util import sintel_pair
    R = O.ref()
    if R is None or not hasattr(R, "ref_op_create"):
        raise SystemExit("oracle/_ref without the model objects: run `make -C oracle ref` first")
    i0, i1 = sintel_pair(crop=MR.CROP)
    out = {"crop": np.array(MR.CROP, np.int32)}
    for method in range(8):
        for k, v in MR.replay_reference(O, method, i0, i1).items():
            out["m%d_%s" % (method, k)] = v
    lab = np.empty_like(i0)
    R.ref_rgb_to_lab(i0, i0.shape[1] * i0.shape[2], lab)
    out["lab"] = lab
    # partitions: geometry of three grids on an odd-sized crop, and where every candidate of a queue with many ties goes
    j0, j1 = sintel_pair(crop=(300, 100, 53, 37))
    j0n, j1n = O.preprocess_local(j0, j1)
    for cols, rows in ((3, 2), (2, 3), (4, 1)):
        A = O.Parts("ref", j0, j1, j0n, j1n, cols, rows)
        out["parts_%dx%d_geometry" % (cols, rows)] = np.array([A.geometry(p) for p in range(A.count())], np.int32)
        partition_case(A, 77, 53, 37)
        for p in range(A.count()):
            for d in (0, 1):
                out["parts_%dx%d_queue_%d_%d" % (cols, rows, p, d)] = A.drain(p, d)
        out["parts_%dx%d_any_empty" % (cols, rows)] = np.array([int(A.any_empty())], np.int32)
    np.savez_compressed(os.path.join(HERE, "ref_models.npz"), **out)
    print("wrote ref_models.npz with", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "ref_models.npz")), "bytes")


def ground_truth():
    """The reference's ground-truth flows of its two example pairs we keep frames of, planar (2, h, w) float32."""
    import faldoi_ipol_amd as F
    for sub, tag in (("clean/easy", "sintel_clean_easy"), ("final/hard", "sintel_final_hard")):
        gt = F.flo.read_flo("/root/reference/example_data/%s/gt/frame_0002.flo" % sub)
        dst = os.path.join(HERE, "%s_gt_frame_0002.npz" % tag)
        np.savez_compressed(dst, gt=gt)
        print("wrote", os.path.basename(dst), gt.shape, os.path.getsize(dst), "bytes")


def example_frames():
    """The reference's six example sequences, frames 1..4 (example_data/{clean,final}/{easy,medium,hard}/frame_000N.png), copied
    verbatim as data files: SURVEY 8d's configs[3] input is their 18 consecutive frame pairs (+ 14 synthetic).  1024x436 RGB PNGs,
    12.9 MB in all."""
    import shutil
    root = "/root/reference/example_data"
    for render in ("clean", "final"):
        for level in ("easy", "medium", "hard"):
            for k in range(1, 5):
                src = os.path.join(root, render, level, "frame_%04d.png" % k)
                dst = os.path.join(HERE, "sintel_%s_%s_frame_%04d.png" % (render, level, k))
                if not os.path.exists(dst):
                    shutil.copyfile(src, dst)
                    print("copied", os.path.basename(dst), os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
    models()
    ground_truth()
    example_frames()
