"""CPU suite, part 1c: the oracle against committed vectors of the reference's energy-model, table and partition objects
(tests/golden/ref_models.npz, written by tests/golden/make_golden.py from oracle/_ref in the build container).  Same pins as
tests/test_oracle_models_cpu.py, but alive wherever the repository is -- the reference itself cannot travel."""
import os

import numpy as np
import pytest

import models_replay as MR
from util import GOLDEN, bits_equal, mismatch_report, sintel_pair

@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "ref_models.npz"))


@pytest.mark.parametrize("method", list(range(8)))
def test_of_estimation_as_called_equals_the_committed_reference_vectors(oracle, G, method):
    """prepare_stuff's frames, every window's energy, the flow after the sequence, TV-l2's persistent divergence planes, the
    non-local weights and all 48 dual planes: what the reference's objects returned, bit for bit (both directions)."""
    assert tuple(G["crop"]) == MR.CROP
    i0, i1 = sintel_pair(crop=MR.CROP)
    got = MR.replay_oracle(oracle, method, i0, i1)
    key = lambda k: "m%d_%s" % (method, k)
    assert bits_equal(got["frame0"], G[key("frame0")]) and bits_equal(got["frame1"], G[key("frame1")])
    for d in (0, 1):
        e, r = got["energy%d" % d], G[key("energy%d" % d)]
        assert np.array_equal(e.view(np.uint32), r.view(np.uint32)), (method, d, np.flatnonzero(e.view(np.uint32) != r.view(np.uint32))[:5])
        assert bits_equal(got["flow%d" % d], G[key("flow%d" % d)]), mismatch_report(got["flow%d" % d], G[key("flow%d" % d)])
        if method in (0, 1):
            assert bits_equal(got["div%d" % d], G[key("div%d" % d)])
    if method in (2, 3, 6, 7):
        valid = G[key("dual_valid")]
        v2 = np.concatenate([valid, valid])
        assert bits_equal(np.where(v2, got["duals"], 0), G[key("duals")])
        assert bits_equal(got["nl_weights"], G[key("nl_weights")])


def test_lab_image_equals_the_committed_reference_vector(oracle, G):
    i0, _ = sintel_pair(crop=MR.CROP)
    lab = np.empty_like(i0)
    oracle.lib().orc_rgb_to_lab_local(i0, i0.shape[1] * i0.shape[2], lab)
    assert bits_equal(lab, G["lab"]), mismatch_report(lab, G["lab"])


@pytest.mark.parametrize("grid", [(3, 2), (2, 3), (4, 1)])
def test_partition_split_equals_the_committed_reference_vectors(oracle, G, grid):
    """Geometry of the partitions and where each candidate of a queue with many equal energies ends up, in pop order
    (src/aux_partitions.cpp: init_subimage_partitions, update_candidate_queues with its index quirk, anyEmptyQueues)."""
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden import partition_case
    cols, rows = grid
    j0, j1 = sintel_pair(crop=(300, 100, 53, 37))
    j0n, j1n = oracle.preprocess_local(j0, j1)
    B = oracle.Parts("orc", j0, j1, j0n, j1n, cols, rows)
    tag = "parts_%dx%d_" % (cols, rows)
    assert np.array_equal(np.array([B.geometry(p) for p in range(B.count())], np.int32), G[tag + "geometry"])
    partition_case(B, 77, 53, 37)
    for p in range(B.count()):
        for d in (0, 1):
            q, r = B.drain(p, d), G[tag + "queue_%d_%d" % (p, d)]
            assert q.shape == r.shape and np.array_equal(q.view(np.uint32), r.view(np.uint32)), (p, d)
    assert int(B.any_empty()) == int(G[tag + "any_empty"][0])
