"""Regenerates tests/golden/ref_primitives.npz from the REFERENCE's own compiled primitives.

Run in the build container only (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_golden.py
The arrays are data (seeded inputs + what the reference objects returned for them); no reference
source text is stored.  The GPU box and CPU-only CI read the .npz and never need the reference.

Also committed next to it:
  sintel_clean_easy_frame_000{2,3}.png  -- the reference's example_data/clean/easy frames (inputs
                                           of BASELINE configs 1/2), copied verbatim as data files.
  sintel_final_hard_frame_000{2,3}.png  -- example_data/final/hard frames (a large-motion pair for the
                                           GPU == oracle test; no reference output recorded for it).
  known_answers.json                    -- what the compiled reference produced for whole solves
                                           (iteration counts, sha256 of the .flo), SURVEY.md App. D.
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
    np.savez_compressed(os.path.join(HERE, "ref_primitives.npz"), **out)
    print("wrote ref_primitives.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
