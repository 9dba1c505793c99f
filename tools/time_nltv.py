import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import faldoi_ipol_amd as F
from util import sintel_pair
ctx = F.Context(0); i0, i1 = sintel_pair()
i0n, i1n = ctx.preprocess_global(i0, i1)
t = time.perf_counter(); lab = F.capi.image_to_lab(i0); print("lab host %.1f ms" % ((time.perf_counter() - t) * 1e3))
l1 = os.environ.get("NLTV_L1") == "1"            # NLTV-L1 (the same regulariser, TV-L1's data term) instead of NLTV-CSAD
for _ in range(2):
    t = time.perf_counter()
    u, ms = (ctx.nltvl1_global if l1 else ctx.nltvcsad_global)(i0n, i1n, lab, np.zeros((2, 436, 1024), np.float32), warps=1, max_inner=400)
    print(("NLTV-L1 " if l1 else "NLTV-CSAD") + " 1024x436, 1 warp x 400 its: device %.2f ms (%.1f us/iter, %.2f Gpx*it/s), wall %.1f ms"
          % (ms, ms * 1e3 / 400, 1024 * 436 * 400 / ms / 1e6, (time.perf_counter() - t) * 1e3))
if os.environ.get("TVCSAD") == "1":              # TV-CSAD, the reference's defaults for -m 4 (5 warps, <= 400 iterations)
    for _ in range(2):
        t = time.perf_counter()
        u, its, errs, xi, st = ctx.tvcsad_global(i0n, i1n, np.zeros((2, 436, 1024), np.float32), warps=5)
        print("TV-CSAD 1024x436, 5 warps, iterations %s: device %.2f ms (%.1f us/iter), wall %.1f ms"
              % (list(its), st.solve_ms, st.solve_ms * 1e3 / max(1, int(its.sum())), (time.perf_counter() - t) * 1e3))
