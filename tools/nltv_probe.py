import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import faldoi_ipol_amd as F
from util import sintel_pair
ctx = F.Context(0)
i0, i1 = sintel_pair()
i0n, i1n = ctx.preprocess_global(i0, i1)
lab = F.capi.image_to_lab(i0)
u0 = np.zeros((2, 436, 1024), np.float32)
ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=1, max_inner=20)
res = {}
for mode in ("0", "1", "0", "1"):
    os.environ["FALDOI_NL_PRIMAL1"] = mode
    u, ms = ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=1, max_inner=400)
    print("NLTV-CSAD 1x400, one-pixel primal kernel forced=%s: %.2f ms" % (mode, ms), flush=True)
    res.setdefault(mode, u)
    u2, ms2 = ctx.nltvl1_global(i0n, i1n, lab, u0, warps=1, max_inner=400)
    print("NLTV-L1   1x400, forced=%s: %.2f ms" % (mode, ms2), flush=True)
    res.setdefault("l" + mode, u2)
print("same bits CSAD:", np.array_equal(res["0"].view(np.uint32), res["1"].view(np.uint32)), " L1:", np.array_equal(res["l0"].view(np.uint32), res["l1"].view(np.uint32)))
