#!/usr/bin/env python3
"""Tuning aid (not part of the product): times the BASELINE configs[1] solve under different
tile shapes (FALDOI_TILE=R,NW) and chunk sizes in ONE process, interleaved rounds."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import faldoi_ipol_amd as F  # noqa: E402
from util import synthetic_pair  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="4,16;8,8;4,8;2,16;6,8;3,16")
    ap.add_argument("--chunks", default="10,5")
    ap.add_argument("--inner", default="10")
    ap.add_argument("--warps", type=int, default=300)
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--size", default="1024x436")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "sintel"])
    a = ap.parse_args()
    w, h = map(int, a.size.split("x"))
    ctx = F.Context(0)
    b = F.Tvl1Batch(ctx, a.pairs, w, h)
    for i in range(a.pairs):
        if a.data == "sintel":          # tuning tool only: preprocessing borrowed from the checker
            from oracle import oracle as O
            from util import sintel_pair
            i0, i1 = sintel_pair()
            i0n, i1n = O.preprocess_global(i0, i1)
            gt = np.zeros((2, h, w), np.float32)
        else:
            i0n, i1n, gt = synthetic_pair(w, h, seed=1234 + i, amp=4.0)
        b.upload(i, i0n, i1n, np.zeros_like(gt))
    res = {}
    for rnd in range(a.rounds + 1):
        for tile in a.tiles.split(";"):
            os.environ["FALDOI_TILE"] = tile
            for ch in map(int, a.chunks.split(",")):
                for inner in map(int, a.inner.split(",")):
                    _, _, ms, nl = b.solve(warps=a.warps, max_inner=inner, tol_stop=0, chunk=ch)
                    if rnd:
                        res.setdefault((tile, ch, inner), []).append((ms, nl))
    print("%-7s %5s %5s %10s %9s %12s" % ("tile", "chunk", "inner", "ms/solve", "us/launch", "Gpx*it/s"))
    for (tile, ch, inner), v in res.items():
        ms = float(np.median([x[0] for x in v]))
        nl = v[0][1]
        print("%-7s %5d %5d %10.3f %9.2f %12.1f" % (tile, ch, inner, ms, ms * 1e3 / nl,
                                                    a.pairs * w * h * a.warps * inner / ms / 1e6))


if __name__ == "__main__":
    main()
