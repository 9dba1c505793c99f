#!/usr/bin/env python3
"""Tuning aid: times BASELINE configs[0]/[1]-style solves (reference defaults, tolerance stop on)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import faldoi_ipol_amd as F
from util import sintel_pair
ctx = F.Context(0)
i0, i1 = sintel_pair()
b = F.Tvl1Batch(ctx, 1, 1024, 436)
b.upload_images(0, i0, i1, np.zeros((2, 436, 1024), np.float32))
for grp in (os.environ.get("FALDOI_SYNC_GROUP", "8"),):
    for rep in range(3):
        t = time.perf_counter()
        iters, errs, ms, nl = b.solve()
        wall = (time.perf_counter() - t) * 1e3
        tot = int(iters.sum())
        print("config1: iters", list(iters[0]), "device ms %.3f wall ms %.3f launches %d -> %.1f Gpx*it/s (%.2f us/iter)"
              % (ms, wall, nl, 1024 * 436 * tot / ms / 1e6, ms * 1e3 / tot))
