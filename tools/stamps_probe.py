import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import faldoi_ipol_amd as F
from util import GOLDEN, load_rgb_planar
ctx = F.Context(0)
f2 = os.path.join(GOLDEN, "sintel_clean_easy_frame_0002.png"); f3 = os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png")
i0n, i1n = ctx.preprocess_global(load_rgb_planar(f2), load_rgb_planar(f3))
b = F.Tvl1Batch(ctx, 1, 1024, 436)
b.upload(0, i0n, i1n, np.zeros((2, 436, 1024), np.float32))
for _ in range(3): b.solve(warps=300, max_inner=10, tol_stop=0)
os.environ["FALDOI_STAMPS"] = "2"
for k in (10,):
    print("chunk", k, flush=True)
    b.solve(warps=300, max_inner=k, tol_stop=0, chunk=k)
