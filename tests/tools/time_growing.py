"""Times the local step (match growing) on the GPU and, optionally, the CPU checker on the same seeds.
usage: python tests/tools/time_growing.py [--crop x0,y0,w,h] [--frac 0.01] [--pairs 1] [--cpu] [--budget N]"""
import argparse
import sys
import time

import numpy as np

import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import faldoi_ipol_amd as F  # noqa: E402
import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--crop", default="")
ap.add_argument("--frac", type=float, default=0.011)
ap.add_argument("--pairs", type=int, default=1)
ap.add_argument("--cpu", action="store_true")
ap.add_argument("--budget", type=int, default=16384)
ap.add_argument("--loc_it", type=int, default=3)
ap.add_argument("--split", default="", help="H,V[,mode]: image partitions (-split_img mode -h_parts H -v_parts V; mode 1 or 2)")
ap.add_argument("--verbose", type=int, default=0)
ap.add_argument("--scale", type=int, default=1, help="enlarge the frames k times (bicubic-free pixel repetition + smoothing by the preprocessing)")
ap.add_argument("--hard", action="store_true", help="the final/hard example pair (large motion) instead of clean/easy")
a = ap.parse_args()
crop = tuple(int(v) for v in a.crop.split(",")) if a.crop else None
i0, i1 = util.sintel_pair(crop)
if a.hard:
    i0 = util.load_rgb_planar(os.path.join(util.GOLDEN, "sintel_final_hard_frame_0002.png"), crop)
    i1 = util.load_rgb_planar(os.path.join(util.GOLDEN, "sintel_final_hard_frame_0003.png"), crop)
if a.scale > 1:
    i0 = np.ascontiguousarray(np.repeat(np.repeat(i0, a.scale, axis=1), a.scale, axis=2))
    i1 = np.ascontiguousarray(np.repeat(np.repeat(i1, a.scale, axis=1), a.scale, axis=2))
ctx = F.Context()
i0n, i1n = ctx.preprocess_local(i0, i1)
h, w = i0n.shape
rng = np.random.default_rng(1)
seeds = []
for src, dst in ((i0n, i1n), (i1n, i0n)):
    u = ctx.tvl1_global(src, dst, np.zeros((2, h, w), np.float32), warps=5)[0]
    s = np.full_like(u, np.nan)
    m = rng.random((h, w)) < a.frac
    s[:, m] = np.round(u[:, m])
    seeds.append(s)
go, ba = seeds
print("size %dx%d, seeds %d / %d" % (w, h, np.isfinite(go[0]).sum(), np.isfinite(ba[0]).sum()), flush=True)
t = time.time()
sp = [int(v) for v in a.split.split(",")] if a.split else []
split = dict(split_img=sp[2] if len(sp) > 2 else 1, h_parts=sp[0], v_parts=sp[1]) if sp else {}
if a.pairs == 1:
    out, ene, st = ctx.match_growing(i0, i1, go, ba, budget=a.budget, iterations_of=a.loc_it, verbose=a.verbose, **split)
else:
    outs, enes, st = ctx.match_growing_batch([i0] * a.pairs, [i1] * a.pairs, [go] * a.pairs, [ba] * a.pairs,
                                            budget=a.budget, iterations_of=a.loc_it, verbose=a.verbose, **split)
    out = outs[0]
    assert all(util.bits_equal(o, out) for o in outs)
dt = time.time() - t
solves = sum(st.solves)
print("GPU: %.2f s wall, growing %.2f s, %d windows (%d fills, %d pops) => %.1f k windows/s, %d launches"
      % (dt, st.growing_ms / 1e3, solves, sum(st.fills), sum(st.pops), solves / st.growing_ms, st.launches), flush=True)
cy = list(st.cycles)
print("cycles per window: queue %.0f, fill %.0f (per fill %.0f), solve %.0f" % (cy[0] / solves, cy[1] / solves, cy[1] / max(1, sum(st.fills)), cy[2] / solves))
print("FB kept:", [round(v, 4) for v in st.fb_kept[:2 * a.loc_it]], "finite:", float(np.isfinite(out).mean()))
if a.cpu:
    from oracle import oracle as O
    t = time.time()
    want, wene, ost = O.local_faldoi(i0n, i1n, go, ba, loc_it=a.loc_it, **split)
    dt = time.time() - t
    print("CPU checker (forward || backward; partitions on up to 16 threads): %.2f s => %.1f k windows/s; bit-identical flow: %s, energy: %s"
          % (dt, (ost[1] + ost[4]) / dt / 1e3, util.bits_equal(out, want), util.bits_equal(ene if a.pairs == 1 else enes[0], wene)))
