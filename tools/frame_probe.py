"""Development aid: per-tile time of a warp's ten iterations in the persistent TV-L1 kernel (FALDOI_STAMPS=2 grids) on inputs whose
statistics are the same everywhere -- what remains of the image-frame tiles' excess is then the code they run, not their data.
usage: python tools/frame_probe.py [real|noise|both]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import faldoi_ipol_amd as F
from util import GOLDEN, load_rgb_planar

W, H = 1024, 436


def run(ctx, name, i0n, i1n, u0):
    b = F.Tvl1Batch(ctx, 1, W, H)
    b.upload(0, i0n, i1n, u0)
    kw = dict(warps=300, max_inner=10, tol_stop=0)
    for _ in range(2):
        b.solve(**kw)
    t = min(b.solve(**kw)[2] for _ in range(5))
    sys.stderr.write("=== %s: %.3f ms per solve, persistent %s\n" % (name, t, b.last_persistent()))
    sys.stderr.flush()
    os.environ["FALDOI_STAMPS"] = "2"
    b.solve(**kw)
    os.environ.pop("FALDOI_STAMPS")
    b.close()


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "both"
    with F.Context(0) as ctx:
        if mode in ("real", "both"):
            i0n, i1n = ctx.preprocess_global(load_rgb_planar(os.path.join(GOLDEN, "sintel_clean_easy_frame_0002.png")),
                                             load_rgb_planar(os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png")))
            run(ctx, "example pair", i0n, i1n, np.zeros((2, H, W), np.float32))
        if mode in ("noise", "both"):
            rng = np.random.default_rng(5)
            # white noise blurred over ~3 px, the second frame an independent one: no tile converges, every dual saturates
            def tex():
                a = rng.random((H + 8, W + 8)).astype(np.float32)
                k = np.ones(5, np.float32) / 5
                a = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, a)
                a = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, a)
                return np.ascontiguousarray(a[4:-4, 4:-4])
            run(ctx, "stationary noise (independent frames)", tex(), tex(), np.zeros((2, H, W), np.float32))


if __name__ == "__main__":
    main()
