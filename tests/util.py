"""Shared test inputs: the Sintel example frames (committed fixtures) and seeded synthetic pairs."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)


def load_rgb_planar(path, crop=None):
    """PNG -> (3,h,w) float32 in 0..255, the layout iio_read_image_float_split gives the reference."""
    from PIL import Image
    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)
    if crop is not None:
        x0, y0, w, h = crop
        a = a[y0:y0 + h, x0:x0 + w]
    return np.ascontiguousarray(a.transpose(2, 0, 1))


def sintel_pair(crop=None):
    i0 = load_rgb_planar(os.path.join(GOLDEN, "sintel_clean_easy_frame_0002.png"), crop)
    i1 = load_rgb_planar(os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png"), crop)
    return i0, i1


def reference_pairs():
    """The 18 frame pairs of SURVEY 8d's configs[3] input: the reference's six example sequences (clean / final x easy / medium /
    hard), frames 1->2, 2->3, 3->4 -- committed fixtures.  The headline's pair (clean/easy 2->3) comes first.
    Returns [(name, path of I0, path of I1)]."""
    out = []
    for render in ("clean", "final"):
        for level in ("easy", "medium", "hard"):
            for k in (2, 1, 3):
                out.append(("%s/%s %d->%d" % (render, level, k, k + 1),
                            os.path.join(GOLDEN, "sintel_%s_%s_frame_%04d.png" % (render, level, k)),
                            os.path.join(GOLDEN, "sintel_%s_%s_frame_%04d.png" % (render, level, k + 1))))
    return out


def flo_sha256(flow):
    import faldoi_ipol_amd as F
    return hashlib.sha256(F.flo.flo_bytes(flow)).hexdigest()


def synthetic_pair(w, h, seed=1234, amp=3.0):
    """Smooth random texture I0, a smooth flow, and I1 = I0 sampled at x - flow (bilinear; the
    exact warp model does not matter, both sides of a parity test see the same planes).
    Returns preprocessed-looking planes in [0,1]: (i0n, i1n, flow_gt)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w))
    for _ in range(24):
        lam = rng.uniform(6, 80)
        th = rng.uniform(0, np.pi)
        ph = rng.uniform(0, 2 * np.pi)
        img += lam * np.sin(2 * np.pi * (xx * np.cos(th) + yy * np.sin(th)) / lam + ph)
    img = (img - img.min()) / (img.max() - img.min())
    fu = amp * np.sin(2 * np.pi * yy / (1.7 * h) + 0.3) * np.cos(2 * np.pi * xx / (1.3 * w))
    fv = 0.6 * amp * np.cos(2 * np.pi * xx / (2.1 * w) + 1.0)

    def sample(im, x, y):
        x = np.clip(x, 0, w - 1.001)
        y = np.clip(y, 0, h - 1.001)
        x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
        ax, ay = x - x0, y - y0
        return ((1 - ax) * (1 - ay) * im[y0, x0] + ax * (1 - ay) * im[y0, x0 + 1]
                + (1 - ax) * ay * im[y0 + 1, x0] + ax * ay * im[y0 + 1, x0 + 1])

    i1 = sample(img, xx - fu, yy - fv)
    return img.astype(np.float32), i1.astype(np.float32), np.stack([fu, fv]).astype(np.float32)


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def mismatch_report(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    ne = a.view(np.uint32) != b.view(np.uint32)
    idx = np.argwhere(ne)
    return "%d of %d values differ; max |d| = %g; first at %s" % (
        ne.sum(), ne.size, float(np.nanmax(np.abs(a.astype(np.float64) - b))), idx[0] if len(idx) else None)
