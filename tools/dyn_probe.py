"""Development aid: the reference-default solve (5 warps, tolerance stop) of the example pair on the launch-per-chunk path and on
the persistent kernel with the stop test inside -- time of both, and that counts, errors and flow are identical."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import faldoi_ipol_amd as F
from util import GOLDEN, load_rgb_planar
ctx = F.Context(0)
i0n, i1n = ctx.preprocess_global(load_rgb_planar(os.path.join(GOLDEN, "sintel_clean_easy_frame_0002.png")), load_rgb_planar(os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png")))
b = F.Tvl1Batch(ctx, 1, 1024, 436)
b.upload(0, i0n, i1n, np.zeros((2, 436, 1024), np.float32))
res = {}
for persist in ("0", "1"):
    os.environ["FALDOI_PERSIST"] = persist
    for _ in range(2): b.solve()
    t = min(b.solve()[2] for _ in range(5))
    it, er, _, nl = b.solve()
    u = b.download(0)
    res[persist] = (u, it.copy(), er.copy())
    print("persist", persist, "%.3f ms" % t, "launches", nl, "persistent", b.last_persistent(), it)
print("identical:", np.array_equal(res["0"][0].view(np.uint32), res["1"][0].view(np.uint32)), np.array_equal(res["0"][1], res["1"][1]), np.array_equal(res["0"][2].view(np.uint32), res["1"][2].view(np.uint32)))
b.close()
