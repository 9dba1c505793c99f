import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import faldoi_ipol_amd as F
    from util import GOLDEN, load_rgb_planar
    ctx = F.Context(0)
    f2 = os.path.join(GOLDEN, "sintel_clean_easy_frame_0002.png"); f3 = os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png")
    i0n, i1n = ctx.preprocess_global(load_rgb_planar(f2), load_rgb_planar(f3))
    b = F.Tvl1Batch(ctx, 1, 1024, 436)
    b.upload(0, i0n, i1n, np.zeros((2, 436, 1024), np.float32))
    for _ in range(3): b.solve(warps=300, max_inner=10, tol_stop=0)
    t1 = min(b.solve(warps=300, max_inner=10, tol_stop=0)[2] for _ in range(8))
    os.environ["FALDOI_WARP_REPEAT"] = "2"
    t2 = min(b.solve(warps=300, max_inner=10, tol_stop=0)[2] for _ in range(8))
    print("LDS", os.environ.get("FALDOI_WARP_LDS", "0"), "WPS", os.environ.get("FALDOI_WARP_WPS", "4"), "solve %.3f ms, warp kernel %.2f us" % (t1, (t2 - t1) * 1e3 / 300), flush=True)
else:
    for lds in ("0",):
        env = dict(os.environ, FALDOI_WARP_LDS=lds)
        subprocess.run([sys.executable, __file__, "child"], env=env)
