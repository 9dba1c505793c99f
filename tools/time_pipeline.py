"""BASELINE configs[2] shape: the whole chain (seeds -> local step -> global step) on the final/hard example pair with
the one-process driver, with and without image partitions.  The seeds stand in for the SIFT matches (the matcher
binaries cannot be run in this environment): ~1.1 % of a dense TV-L1 flow, rounded to integers."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import faldoi_ipol_amd as F  # noqa: E402
import util  # noqa: E402

f2 = os.path.join(util.GOLDEN, "sintel_final_hard_frame_0002.png")
f3 = os.path.join(util.GOLDEN, "sintel_final_hard_frame_0003.png")
i0, i1 = util.load_rgb_planar(f2), util.load_rgb_planar(f3)
ctx = F.Context()
a, b = ctx.preprocess_local(i0, i1)
h, w = a.shape
rng = np.random.default_rng(1)
tmp = tempfile.mkdtemp()
for name, (src, dst) in (("m0.txt", (a, b)), ("m1.txt", (b, a))):
    u = ctx.tvl1_global(src, dst, np.zeros((2, h, w), np.float32), warps=5)[0]
    m = rng.random((h, w)) < 0.011
    ys, xs = np.nonzero(m)
    with open(os.path.join(tmp, name), "w") as f:
        for y, x in zip(ys, xs):
            f.write("%d %d %d %d\n" % (x, y, x + int(round(float(u[0, y, x]))), y + int(round(float(u[1, y, x])))))
ctx.close()
open(os.path.join(tmp, "ims.txt"), "w").write(f2 + "\n" + f3 + "\n")
exe = F.build.build_host()["faldoi_flow"]
for extra in ([], ["-split_img", "1", "-h_parts", "3", "-v_parts", "2"], ["-split_img", "2", "-h_parts", "3", "-v_parts", "2"],
              ["-split_img", "2", "-h_parts", "8", "-v_parts", "6"],
              ["-split_img", "2", "-h_parts", "8", "-v_parts", "6", "-first_h_parts", "4", "-first_v_parts", "3"]):
    t = time.time()
    r = subprocess.run([exe, os.path.join(tmp, "ims.txt"), os.path.join(tmp, "m0.txt"), os.path.join(tmp, "m1.txt"),
                        os.path.join(tmp, "out.flo"), "-matcher", "cut", "-verbose", "1"] + extra, capture_output=True, text=True)
    print(" ".join(extra) or "(whole image)", "->", [l for l in r.stdout.splitlines() if l.startswith("faldoi_flow")], "wall %.1f s" % (time.time() - t), "rc", r.returncode)
