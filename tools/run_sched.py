import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import faldoi_ipol_amd as F
from util import sintel_pair
ctx = F.Context(0); i0, i1 = sintel_pair()
b = F.Tvl1Batch(ctx, 1, 1024, 436); b.upload_images(0, i0, i1, np.zeros((2, 436, 1024), np.float32))
inner = int(sys.argv[1]); chunk = int(sys.argv[2]); warps = int(sys.argv[3])
for _ in range(2):
    print(b.solve(warps=warps, max_inner=inner, tol_stop=0, chunk=chunk)[2:])
