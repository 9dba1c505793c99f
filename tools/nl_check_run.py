"""Runs the fused NLTV kernel's geometry sweep on the range-checked build of the library (FALDOI_LIB must name
libfaldoi_hip_check.so: build.build_check()) and prints the report of faldoi_debug_nl_check as one JSON line.
Started as a child process by tests/test_gpu_round4.py; the results themselves are compared with the product build's by the caller
through the sha256 printed here."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import faldoi_ipol_amd as F
from util import sintel_pair

W, H = 1024, 436


def sweep(ctx):
    """The geometries of tests/test_gpu_round3.py's fused-kernel test (same generator, same draws) plus the full frame."""
    rng = np.random.default_rng(20261005)
    full0, full1 = sintel_pair()
    h256 = hashlib.sha256()
    for case in range(31):
        if case < 6:
            w, h = [(2, 2), (3, 7), (64, 16), (65, 17), (128, 33), (7, 40)][case]
        elif case == 30:
            w, h = W, H
        else:
            w, h = int(rng.integers(6, 420)), int(rng.integers(6, 150))
        x0, y0 = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        c0 = np.ascontiguousarray(full0[:, y0:y0 + h, x0:x0 + w])
        c1 = np.ascontiguousarray(full1[:, y0:y0 + h, x0:x0 + w])
        lab = F.capi.image_to_lab(c0)
        i0n, i1n = c0.mean(axis=0) / np.float32(255), c1.mean(axis=0) / np.float32(255)
        u0 = rng.normal(0, float(rng.choice([0, 0.5, 3.0])), (2, h, w)).astype(np.float32)
        prm = dict(warps=int(rng.integers(1, 3)), max_inner=int(rng.integers(1, 6)))
        for l1 in (False, True):
            for rows in (8, 12, 14, 16):
                os.environ["FALDOI_NL_ROWS"] = str(rows)
                u = (ctx.nltvl1_global if l1 else ctx.nltvcsad_global)(i0n, i1n, lab, u0, **prm)[0]
                h256.update(np.ascontiguousarray(u).tobytes())
    return h256.hexdigest()


def main():
    with F.Context(0) as ctx:
        ctx.nl_check_report(reset=True)
        digest = sweep(ctx)
        bits, count, worst, launches, compiled = ctx.nl_check_report(reset=True)
        out = dict(bits=bits, violations=count, worst=worst, launches=launches, compiled_in=compiled, sha256=digest,
                   library=F.capi.library_path())
        # the detector is not vacuous: one deliberately out-of-range index per kind
        os.environ["FALDOI_NL_CHECK_SELFTEST"] = "1"
        os.environ["FALDOI_NL_ROWS"] = "16"
        z = np.zeros((2, 40, 70), np.float32)
        ctx.nltvcsad_global(np.zeros((40, 70), np.float32) + 0.5, np.zeros((40, 70), np.float32) + 0.25, np.zeros((3, 40, 70), np.float32), z,
                            warps=1, max_inner=1)
        sb, sc, _, _, _ = ctx.nl_check_report(reset=True)
        out.update(selftest_bits=sb, selftest_violations=sc)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
