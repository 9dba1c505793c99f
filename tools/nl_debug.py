"""Debug aid: NLTV-CSAD fused kernel against the sorting-network streaming path on one crop; prints where they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import faldoi_ipol_amd as F
from oracle import oracle as O
from util import sintel_pair
ctx = F.Context(0)
full0, full1 = sintel_pair()
for (w, h, x0, y0) in [(65, 17, 100, 50), (130, 35, 200, 100), (64, 17, 10, 10), (66, 30, 5, 5)]:
    c0 = np.ascontiguousarray(full0[:, y0:y0 + h, x0:x0 + w]); c1 = np.ascontiguousarray(full1[:, y0:y0 + h, x0:x0 + w])
    lab = F.capi.image_to_lab(c0)
    i0n, i1n = O.preprocess_global(c0, c1)
    u0 = np.random.default_rng(1).normal(0, 0.5, (2, h, w)).astype(np.float32)
    for inner in (1, 2):
        os.environ["FALDOI_CSAD_SORT"] = "1"
        ref = ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=1, max_inner=inner)[0]
        del os.environ["FALDOI_CSAD_SORT"]
        for rows in (8, 12, 14, 16):
            os.environ["FALDOI_NL_ROWS"] = str(rows)
            got = ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=1, max_inner=inner)[0]
            ne = got.view(np.uint32) != ref.view(np.uint32)
            idx = np.argwhere(ne)
            print(w, h, "inner", inner, "rows", rows, "differ", int(ne.sum()), "cols", sorted(set(idx[:, 2].tolist()))[:12], "rows", sorted(set(idx[:, 1].tolist()))[:20])
