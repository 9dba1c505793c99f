"""Shader-clock cycles of one pop / one push of the device queue (one wavefront; FALDOI_REPLAY_CYCLES of the queue replay hook).
usage: python tools/heap_cycles.py [top_levels ...]      (levels of the queue kept in LDS; default 13 8 0)
The replay fills the queue with 45 k random energies, then alternates pops and pushes; the pops of the second half are timed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FALDOI_REPLAY_CYCLES"] = "1"
import faldoi_ipol_amd as F  # noqa: E402

levels = [int(v) for v in sys.argv[1:]] or [13, 8, 0]
rng = np.random.default_rng(3)
n = 65000
ops = np.ones(n, np.int32)
ops[45000:] = np.tile(np.array([0, 1], np.int32), (n - 45000) // 2)
sims = rng.random(n).astype(np.float32) * 100.0
sims[45000:] += 50.0                         # re-inserted elements sink: like a growing's candidates (worse than the one popped)
ctx = F.Context()
want = F.capi.queue_replay(ops, sims)
for lv in levels:
    os.environ["FALDOI_HEAP_TOP_LEVELS"] = str(lv)
    sys.stderr.write("top levels in LDS: %d   " % lv)
    sys.stderr.flush()
    got = ctx.queue_replay(ops, sims)
    assert np.array_equal(got, want)
