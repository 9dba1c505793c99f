"""Times TV-CSAD (global_faldoi -m 4) on the device; optionally against the CPU checker on a crop."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import faldoi_ipol_amd as F  # noqa: E402
import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--crop", default="")
ap.add_argument("--warps", type=int, default=5)
ap.add_argument("--cpu", action="store_true")
a = ap.parse_args()
crop = tuple(int(v) for v in a.crop.split(",")) if a.crop else None
i0, i1 = util.sintel_pair(crop)
ctx = F.Context()
i0n, i1n = ctx.preprocess_global(i0, i1)
h, w = i0n.shape
u0 = np.zeros((2, h, w), np.float32)
ctx.tvcsad_global(i0n, i1n, u0, warps=1, max_inner=5)
t = time.time()
u, its, errs, xi, st = ctx.tvcsad_global(i0n, i1n, u0, warps=a.warps)
dt = time.time() - t
print("%dx%d: iterations %s, device %.1f ms (wall %.1f ms) => %.1f Mpx*iter/s, %d of %d stop tests needed the serial sum"
      % (w, h, list(its), st.solve_ms, dt * 1e3, w * h * int(its.sum()) / st.solve_ms / 1e3, st.chunk_launches, int(its.sum())))
if a.cpu:
    from oracle import oracle as O
    t = time.time()
    want, wits, _, _ = O.tvcsad_global(i0n, i1n, u0, warps=a.warps)
    dt = time.time() - t
    print("CPU checker: %.1f s, iterations %s, bit-identical: %s" % (dt, list(wits), util.bits_equal(u, want)))
