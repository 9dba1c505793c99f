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
    nsolvers = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    L.orc_spec_model.argtypes = [C.c_int]
    L.orc_spec_report.argtypes = [np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")]
    L.orc_spec_model(nsolvers)
    L.orc_set_num_threads(1)                     # the model observes ONE growing; keep the run single-threaded so it is the forward one
    t = time.time()
    O.local_faldoi(a, b, sd[0], sd[1])
    if nsolvers:
        rep = np.zeros(8)
        L.orc_spec_report(rep)
        print("speculation model, %d solver(s) beside the committer: %.2f G cycles instead of %.2f G (x%.2f); of %d windows: %d hits, "
              "%d waited for, %d solved by the committer" % (nsolvers, rep[0] / 1e9, rep[1] / 1e9, rep[1] / rep[0], rep[5], rep[2], rep[4], rep[3]))
        L.orc_spec_model(0)
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
        dx = np.abs(np.diff(seq[:, 0])); dy = np.abs(np.diff(seq[:, 1]))
        indep = (dx > 2 * wr) | (dy > 2 * wr)
        # runs of consecutive windows each overlapping its predecessor: what a 2-window speculation cannot split
        print("  iteration %d: consecutive windows that do NOT overlap: %.1f %%; adjacent (4-neighbour) successors: %.1f %%"
              % (it, 100.0 * indep.mean(), 100.0 * ((dx + dy) == 1).mean()))
        for look in (2, 4, 8, 16):
            # window q is 'free' if it overlaps none of the previous `look` windows
            free = np.ones(len(seq), bool)
            for d in range(1, look + 1):
                ddx = np.abs(seq[d:, 0] - seq[:-d, 0]); ddy = np.abs(seq[d:, 1] - seq[:-d, 1])
                free[d:] &= (ddx > 2 * wr) | (ddy > 2 * wr)
            print("     overlaps none of the previous %2d windows: %.1f %%" % (look, 100.0 * free.mean()))
        print("  iteration %d: %d windows, longest chain %d  ->  parallelism bound %.1fx" % (it, len(seq), longest, len(seq) / longest))


if __name__ == "__main__":
    main()
