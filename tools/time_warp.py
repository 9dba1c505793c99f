import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import faldoi_ipol_amd as F
from util import sintel_pair
ctx = F.Context(0); i0, i1 = sintel_pair()
b = F.Tvl1Batch(ctx, 1, 1024, 436); b.upload_images(0, i0, i1, np.zeros((2, 436, 1024), np.float32))
for rnd in range(3):
    for v in ("4", "6", "8"):
        os.environ["FALDOI_WARP_WPS"] = v
        _, _, ms, nl = b.solve(warps=300, max_inner=10, tol_stop=0)
        # warps x 1 inner: isolates warp kernel + fixed chunk cost
        _, _, ms1, nl1 = b.solve(warps=300, max_inner=1, tol_stop=0)
        if rnd: print("wps", v, "solve %.3f ms  (1-iter schedule: %.3f ms -> %.2f us per warp+chunk(1 it))" % (ms, ms1, ms1 * 1e3 / 300))
