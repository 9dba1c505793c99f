#!/usr/bin/env python3
"""Study (CPU only): how much of a whole-image growing is inherently sequential?

Runs the CPU checker's local step on the example pair, records the order in which ONE growing (one direction, every
local iteration) minimises its windows, and computes the length of the longest dependency chain when a window depends
on every earlier window it shares a pixel with (what a speculative / out-of-order executor would have to respect).
windows / chain length = the upper bound of any order-preserving parallel execution of that growing."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import util  # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    i0, i1 = util.sintel_pair()
    if scale < 1.0:
        h, w = int(436 * scale), int(1024 * scale)
        i0, i1 = i0[:, :h, :w].copy(), i1[:, :h, :w].copy()
    h, w = i0.shape[1:]
    a, b = O.preprocess_local(i0, i1)
    rng = np.random.default_rng(7)
    sd = []
    for src, dst in ((a, b), (b, a)):
        u = O.tvl1_global(src, dst, np.zeros((2, h, w), np.float32), warps=3, max_inner=100)[0]
        s = np.full_like(u, np.nan)
        m = rng.random((h, w)) < 0.011
        s[:, m] = np.round(u[:, m])
        sd.append(s)
    L = O.lib()
    cap = 3 * 8 * h * w
    buf = np.zeros(cap, np.int32)
    L.orc_trace_windows.argtypes = [np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"), C.c_long]
    L.orc_trace_length.restype = C.c_long
    L.orc_trace_windows(buf, cap)
    t = time.time()
    O.local_faldoi(a, b, sd[0], sd[1])
    n = L.orc_trace_length() // 3
    L.orc_trace_windows(buf, 0)
    print("local step %.1f s; one growing direction minimised %d windows" % (time.time() - t, n))
    tr = buf[:3 * n].reshape(n, 3)
    wr = 5
    for it in sorted(set(tr[:, 2].tolist())):
        seq = tr[tr[:, 2] == it]
        depth = np.zeros((h, w), np.int32)
        longest = 0
        hist = np.zeros(len(seq), np.int32)
        for q, (i, j, _) in enumerate(seq):
            y0, y1, x0, x1 = max(j - wr, 0), min(j + wr + 1, h), max(i - wr, 0), min(i + wr + 1, w)
            d = int(depth[y0:y1, x0:x1].max()) + 1
            depth[y0:y1, x0:x1] = d
            hist[q] = d
            longest = max(longest, d)
        # how far apart (in pop order) are consecutive dependent windows: the lookahead a speculative executor needs
        print("  iteration %d: %d windows, longest chain %d  ->  parallelism bound %.1fx" % (it, len(seq), longest, len(seq) / longest))


if __name__ == "__main__":
    main()
