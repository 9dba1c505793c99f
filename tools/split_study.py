"""How the local step's result moves with the image partitions (and what they cost in time).

For one 1024x436 example pair: whole-image growings (the reference's default), the reference's partition mode
(-split_img 1: from local iteration 1 on) and partitions from iteration 0 on (-split_img 2, ours), each followed
by the global TV-L1 minimisation with the reference's defaults.  Prints times and end-point differences against the
whole-image result; --save writes the flows (fp32) for an offline comparison with a ground-truth flow.

usage: python tools/split_study.py [--hard] [--configs "1:3,2;1:8,6;2:3,2;2:8,6;2:16,12"] [--save gpurun_out/split.npz]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import faldoi_ipol_amd as F  # noqa: E402
import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hard", action="store_true")
ap.add_argument("--configs", default="1:3,2;1:8,6;2:3,2;2:8,6;2:16,12")
ap.add_argument("--frac", type=float, default=0.011)
ap.add_argument("--save", default="")
ap.add_argument("--crop", default="")
a = ap.parse_args()
crop = tuple(int(v) for v in a.crop.split(",")) if a.crop else None
name = "final_hard" if a.hard else "clean_easy"
i0 = util.load_rgb_planar(os.path.join(util.GOLDEN, "sintel_%s_frame_0002.png" % name), crop)
i1 = util.load_rgb_planar(os.path.join(util.GOLDEN, "sintel_%s_frame_0003.png" % name), crop)
ctx = F.Context()
i0n, i1n = ctx.preprocess_local(i0, i1)
g0, g1 = ctx.preprocess_global(i0, i1)
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
print("%s %dx%d, seeds %d / %d" % (name, w, h, np.isfinite(go[0]).sum(), np.isfinite(ba[0]).sum()), flush=True)


def epe(x, y):
    return np.sqrt(((x - y) ** 2).sum(0))


runs = [("whole", {})]
for c in a.configs.split(";"):
    if c:
        mode, grid = c.split(":")[:2]
        hp, vp = (int(v) for v in grid.split(","))
        kw = dict(split_img=int(mode), h_parts=hp, v_parts=vp)
        label = "split%s_%dx%d" % (mode, hp, vp)
        if c.count(":") == 2:                      # mode:H,V:FH,FV = the grid of local iteration 0
            fh, fv = (int(v) for v in c.split(":")[2].split(","))
            kw.update(first_h_parts=fh, first_v_parts=fv)
            label += "_first%dx%d" % (fh, fv)
        runs.append((label, kw))
keep = {}
base_local = base_final = None
for label, kw in runs:
    t = time.time()
    loc, ene, st = ctx.match_growing(i0, i1, go, ba, **kw)
    dt = time.time() - t
    fin = ctx.tvl1_global(g0, g1, np.nan_to_num(loc), warps=5)[0]          # the reference's default schedule (tolerance stop on)
    if base_local is None:
        base_local, base_final = loc, fin
    dl, df = epe(loc, base_local), epe(fin, base_final)
    print("%-16s local step %6.2f s (growing %6.2f s, %d windows, %d launches) | vs whole image: local EPE mean %.4f "
          "(>1px: %.2f %%), after the global step mean %.4f (>1px: %.2f %%)"
          % (label, dt, st.growing_ms / 1e3, sum(st.solves), st.launches, np.nanmean(dl), 100 * np.nanmean(dl > 1),
             df.mean(), 100 * (df > 1).mean()), flush=True)
    keep["local_" + label] = loc
    keep["final_" + label] = fin
if a.save:
    os.makedirs(os.path.dirname(os.path.abspath(a.save)), exist_ok=True)
    np.savez_compressed(a.save, **keep)
    print("saved", a.save, flush=True)
