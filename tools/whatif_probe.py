"""Development aid: what the persistent TV-L1 kernel's time is made of.  Builds timing-only variants of the library with one
ingredient removed (FALDOI_WHATIF bits, faldoi_device.hpp; their results are WRONG by construction) under tools/_whatif/, and
times the 1024x436 headline schedule with each, in one process per variant.

usage: python tools/whatif_probe.py build          (CPU container: cross-compiles the variants)
       python tools/whatif_probe.py run [cfg]      (GPU box)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_whatif")
VARIANTS = {0: "product", 1: "no barriers in the iterations", 2: "duals never take the quotient path",
            4: "threshold quotient -> product", 8: "theta quotients -> products", 14: "no exact quotient at all",
            16: "no waiting for neighbours", 32: "no warp constants", 63: "all of the above"}
if os.environ.get("FALDOI_WHATIF_ONLY"):
    VARIANTS = {int(v): VARIANTS[int(v)] for v in os.environ["FALDOI_WHATIF_ONLY"].split(",")}


def build():
    sys.path.insert(0, ROOT)
    import faldoi_ipol_amd.build as B
    os.makedirs(OUT, exist_ok=True)
    for v in VARIANTS:
        so = os.path.join(OUT, "libfaldoi_whatif_%d.so" % v)
        cmd = [B.hipcc()] + B.HIPCC_FLAGS + ["-DFALDOI_WHATIF=%d" % v, "-I" + os.path.join(ROOT, "include"),
                                            "-I" + os.path.join(B.HERE, "csrc"), "-o", so] + B.SOURCES
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        print("built", so, flush=True)


def one(v, cfg):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import faldoi_ipol_amd.build as B
    B.SO = os.path.join(OUT, "libfaldoi_whatif_%d.so" % v)
    B.needs_build = lambda: False
    import numpy as np
    import faldoi_ipol_amd as F
    from util import GOLDEN, load_rgb_planar
    with F.Context(0) as ctx:
        i0n, i1n = ctx.preprocess_global(load_rgb_planar(os.path.join(GOLDEN, "sintel_clean_easy_frame_0002.png")),
                                         load_rgb_planar(os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png")))
        b = F.Tvl1Batch(ctx, 1, 1024, 436)
        b.upload(0, i0n, i1n, np.zeros((2, 436, 1024), np.float32))
        res = []
        for persist in ("0", "1"):
            os.environ["FALDOI_PERSIST"] = persist
            os.environ["FALDOI_PERSIST_CFG"] = cfg
            os.environ["FALDOI_PERSIST_SPINS"] = "100000"
            for _ in range(2):
                b.solve(warps=300, max_inner=10, tol_stop=0)
            res.append(min(b.solve(warps=300, max_inner=10, tol_stop=0)[2] for _ in range(5)))
        print("%-40s per-launch %7.3f ms   persistent %7.3f ms" % (VARIANTS[v], res[0], res[1]), flush=True)
        b.close()


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        cfg = sys.argv[2] if len(sys.argv) > 2 else "3,16,5"
        for v in VARIANTS:
            subprocess.call([sys.executable, os.path.abspath(__file__), "one", str(v), cfg])
    elif sys.argv[1] == "one":
        one(int(sys.argv[2]), sys.argv[3])
