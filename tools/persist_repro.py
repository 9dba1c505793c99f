"""Development aid: re-run one case of the `persist` fuzz family and show what differs and where."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import faldoi_ipol_amd as F
import util
seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([seed, n])
ctx = F.Context()
w, h = int(rng.integers(2, 1100)), int(rng.integers(2, 480))
if rng.random() < 0.4:
    w, h = int(rng.integers(2, 200)), int(rng.integers(2, 140))
i0n, i1n, gt = util.synthetic_pair(max(w, 4), max(h, 4), seed=int(rng.integers(1 << 30)), amp=float(rng.uniform(0, 8)))
i0n, i1n, gt = np.ascontiguousarray(i0n[:h, :w]), np.ascontiguousarray(i1n[:h, :w]), np.ascontiguousarray(gt[:, :h, :w])
u0 = (gt * float(rng.choice([0.0, 0.5, 1.0])) + rng.normal(0, float(rng.choice([0, 0.5, 20.0])), (2, h, w))).astype(np.float32)
if rng.random() < 0.15:
    u0[int(rng.integers(0, 2)), int(rng.integers(0, h)), int(rng.integers(0, w))] = np.nan
xi = (rng.normal(0, float(rng.choice([0.3, 2.0])), (4, h, w))).astype(np.float32) if rng.random() < 0.5 else None
prm = dict(warps=int(rng.integers(1, 6)), max_inner=int(rng.integers(1, 24)), tol_stop=0,
           lambda_=float(rng.choice([40.0, 10.0, 0.5, 120.0])), theta=float(rng.choice([0.3, 0.1, 0.8])),
           tau=float(rng.choice([0.125, 0.25, 0.05])), chunk=int(rng.choice([0, 0, 0, 1, 2, 3, 4, 6, 8])))
cfg = str(rng.choice(["", "", "3,16,5", "4,13,5", "4,11,4", "4,16,8", "3,14,3", "3,16,2", "4,16,4"]))
print(w, h, prm, cfg, "nan in u0:", int(np.isnan(u0).sum()))
for warps in range(1, prm["warps"] + 1):
    res = []
    for persist in ("0", "1"):
        os.environ["FALDOI_PERSIST"] = persist
        if cfg and persist == "1":
            os.environ["FALDOI_PERSIST_CFG"] = cfg
        else:
            os.environ.pop("FALDOI_PERSIST_CFG", None)
        b = F.Tvl1Batch(ctx, 1, w, h)
        b.upload(0, i0n, i1n, u0, xi)
        p = dict(prm); p["warps"] = warps
        its, errs, _, launches = b.solve(**p)
        u, x = b.download(0, want_xi=True)
        res.append((u, x, its.copy(), errs.copy(), b.last_schedule()))
        b.close()
    r, m = res
    du = np.argwhere(r[0].view(np.uint32) != m[0].view(np.uint32))
    dx = np.argwhere(r[1].view(np.uint32) != m[1].view(np.uint32))
    print("warps", warps, "schedule", m[4], "flow diffs", len(du), "dual diffs", len(dx), "errs", r[3], m[3], "iters", r[2], m[2])
    if len(du):
        ys, xs = du[:, 1], du[:, 2]
        print("   flow diff rows %d..%d cols %d..%d, first %s: %r vs %r" % (ys.min(), ys.max(), xs.min(), xs.max(), du[0], r[0][tuple(du[0])], m[0][tuple(du[0])]))
