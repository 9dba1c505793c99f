"""Development aid: the persistent TV-L1 kernel against the per-launch path -- identical bits, and the time of both.

usage: python tools/persist_probe.py [check|time|sweep] ...
  check  several image sizes / schedules / geometries, asserts bit-identical flow, duals, iteration counts and errors
  time   the 1024x436 headline schedule (300 warps x 10 iterations) on both paths, with in-kernel stamps
  sweep  the geometries given as R,NW,k arguments on the headline schedule
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import faldoi_ipol_amd as F
from util import synthetic_pair


def solve(b, persist, cfg=None, **kw):
    os.environ["FALDOI_PERSIST"] = "1" if persist else "0"
    if cfg:
        os.environ["FALDOI_PERSIST_CFG"] = cfg
    else:
        os.environ.pop("FALDOI_PERSIST_CFG", None)
    it, er, ms, launches = b.solve(tol_stop=0, **kw)
    u, xi = b.download(0, want_xi=True)
    return u, xi, it.copy(), er.copy(), ms, launches


def same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def check(ctx):
    bad = 0
    cases = [(192, 128, 3, 10, None), (192, 128, 2, 7, "4,16,8"), (300, 200, 4, 10, "3,16,5"), (300, 200, 3, 12, "4,13,5"),
             (257, 131, 2, 9, "4,11,4"), (640, 360, 3, 10, "3,16,5"), (1024, 436, 5, 10, "3,16,5"), (1024, 436, 3, 10, "4,13,5"),
             (1024, 436, 2, 16, "4,16,8"), (64, 48, 2, 5, "3,14,3")]
    for (w, h, warps, inner, cfg) in cases:
        i0, i1, gt = synthetic_pair(w, h, seed=w + h, amp=2.0)
        u0 = (0.5 * gt).astype(np.float32)
        b = F.Tvl1Batch(ctx, 1, w, h)
        b.upload(0, i0, i1, u0)
        ref = solve(b, False, warps=warps, max_inner=inner)
        new = solve(b, True, cfg, warps=warps, max_inner=inner)
        ok = same(ref[0], new[0]) and same(ref[1], new[1]) and np.array_equal(ref[2], new[2]) and same(ref[3], new[3])
        print("%4dx%-4d warps %d inner %2d cfg %-8s launches %4d -> %d: %s   (%.3f ms vs %.3f ms)"
              % (w, h, warps, inner, cfg, ref[5], new[5], "identical" if ok else "DIFFERENT", ref[4], new[4]), flush=True)
        if new[5] != 1:
            print("   (the persistent path was not taken)")
        if not ok:
            bad += 1
            d = np.argwhere(ref[0].view(np.uint32) != new[0].view(np.uint32))
            print("   first differing flow entries:", d[:5].tolist(), "of", len(d))
        b.close()
    return bad


def headline(ctx, cfgs, stamps):
    from util import GOLDEN, load_rgb_planar
    f2 = os.path.join(GOLDEN, "sintel_clean_easy_frame_0002.png")
    f3 = os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png")
    i0n, i1n = ctx.preprocess_global(load_rgb_planar(f2), load_rgb_planar(f3))
    b = F.Tvl1Batch(ctx, 1, 1024, 436)
    b.upload(0, i0n, i1n, np.zeros((2, 436, 1024), np.float32))
    kw = dict(warps=300, max_inner=10)
    ref = None
    for _ in range(3):
        ref = solve(b, False, **kw)
    t = min(solve(b, False, **kw)[4] for _ in range(5))
    print("per-launch path: %.3f ms  (%.1f Gpx.iter/s)" % (t, 1024 * 436 * 3000 / t / 1e6), flush=True)
    for cfg in cfgs:
        try:
            for _ in range(2):
                new = solve(b, True, cfg, **kw)
            t = min(solve(b, True, cfg, **kw)[4] for _ in range(5))
            ok = same(ref[0], new[0]) and same(ref[1], new[1]) and same(ref[3], new[3])
            print("persistent %-8s: %.3f ms  (%.1f Gpx.iter/s)  launches %d  %s"
                  % (cfg, t, 1024 * 436 * 3000 / t / 1e6, new[5], "identical" if ok else "DIFFERENT"), flush=True)
            if stamps:
                os.environ["FALDOI_STAMPS"] = "1"
                solve(b, True, cfg, **kw)
                os.environ.pop("FALDOI_STAMPS")
        except RuntimeError as e:
            print("persistent %-8s: %s" % (cfg, e), flush=True)
    b.close()


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    with F.Context(0) as ctx:
        if mode == "check":
            sys.exit(1 if check(ctx) else 0)
        if mode == "time":
            headline(ctx, ["3,16,5"], True)
        if mode == "sweep":
            headline(ctx, sys.argv[2:], True)
