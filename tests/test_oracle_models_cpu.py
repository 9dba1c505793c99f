"""CPU suite, part 1b: the oracle against the reference's OWN energy-model, table and partition objects
(oracle/_ref, built by oracle/Makefile from every translation unit of the path that has no boost include):

  * prepare_stuff + of_estimation exactly as match_growing_variational calls them (src/energy_model.cpp:276,691);
  * init_subimage_partitions / image_to_partitions / update_candidate_queues / anyEmptyQueues (src/aux_partitions.cpp).

These run where the reference sources were present at build time (this container); on a box where oracle/_ref did
not travel they are skipped, and the committed vectors of tests/golden keep the same pins alive.
"""
import numpy as np
import pytest

from util import bits_equal, mismatch_report, sintel_pair


@pytest.fixture(scope="module")
def R(oracle):
    r = oracle.ref()
    if r is None or not hasattr(r, "ref_op_create"):
        pytest.skip("oracle/_ref (reference objects) not built here")
    return r


def _windows(rng, w, h, count):
    """Centres as a growing visits them: corners, frame pixels, overlapping neighbours, interior."""
    pts = [(0, 0), (w - 1, 0), (0, h - 1), (w - 1, h - 1), (1, 1), (w // 2, 0), (0, h // 2), (w - 1, h // 2), (w // 2, h - 1)]
    i, j = w // 2, h // 2
    for _ in range(count):
        i = int(np.clip(i + rng.integers(-3, 4), 0, w - 1))
        j = int(np.clip(j + rng.integers(-3, 4), 0, h - 1))
        pts.append((i, j))
    pts += [(int(rng.integers(0, w)), int(rng.integers(0, h))) for _ in range(count)]
    return pts


@pytest.mark.parametrize("method", [0, 1])
def test_prepare_stuff_and_of_estimation_as_called(oracle, R, method):
    """The reference's operator surface, driven the way add_neighbors drives it (seeding radius 1, growing radius 5), against
    the oracle's preprocessing + window operator: frames, gradients, every window's energy, the flow and both persistent
    divergence planes -- bit for bit, forward and backward."""
    i0, i1 = sintel_pair(crop=(420, 170, 72, 56))
    h, w = i0.shape[1:]
    op = oracle.RefOp(method, w, h, wr=5)
    r0, r1 = op.prepare(i0, i1)
    o0, o1 = oracle.preprocess_local(i0, i1)
    assert bits_equal(r0, o0) and bits_equal(r1, o1)
    L = oracle.lib()
    rng = np.random.default_rng(3 + method)
    weight = np.zeros(11, np.float32)
    L.orc_gaussian1d_weight(weight, 5)
    for d, (src, dst) in enumerate(((o0, o1), (o1, o0))):
        gx, gy = oracle.centered_gradient(dst)
        assert bits_equal(op.plane(d, "I1x"), gx) and bits_equal(op.plane(d, "I1y"), gy)      # of the TARGET frame
        u = (rng.standard_normal((2, h, w)) * 1.5).astype(np.float32)
        op.set_flow(d, u)
        state = np.zeros((6, h, w), np.float32)
        import ctypes as C
        for k, (i, j) in enumerate(_windows(rng, w, h, 40)):
            wr = 1 if k % 3 == 0 else 5
            ii, ij, ei, ej = oracle.patch_index(wr, w, h, i, j)
            e_ref = op.estimate(d, i, j, wr)
            it = C.c_int(0)
            if method == 0:
                e = L.orc_local_tvl1_estimate(src, dst, gx, gy, u[0], u[1], state, w, h, ii, ij, ei, ej, 4, C.byref(it))
            else:
                iiw, ijw = max(0, -(i - wr)), max(0, -(j - wr))
                e = L.orc_local_tvl1w_estimate(src, dst, gx, gy, u[0], u[1], state, w, h, ii, ij, ei, ej, 4, C.byref(it), weight,
                                               iiw, ijw, wr)
            assert np.float32(e).view(np.uint32) == np.float32(e_ref).view(np.uint32), (d, k, i, j, wr, e, e_ref)
            assert bits_equal(op.get_flow(d), u), (d, k, mismatch_report(op.get_flow(d), u))
        assert bits_equal(op.plane(d, "div_xi1"), state[4]) and bits_equal(op.plane(d, "div_xi2"), state[5])
    op.close()


@pytest.mark.parametrize("grid", [(3, 2), (2, 3), (1, 3), (4, 1)])
def test_partition_bookkeeping_matches_the_reference(oracle, R, grid):
    """init_subimage_partitions (geometry, frames), image_to_partitions in both directions (every state plane the TV-l2
    functional owns, flow, energies, fixed / trust maps, output flow), update_candidate_queues (membership, the index
    adjustment quirk of src/aux_partitions.cpp:1992, pop order among equal energies) and anyEmptyQueues."""
    cols, rows = grid
    i0, i1 = sintel_pair(crop=(300, 100, 53, 37))            # odd sizes: remainders go to the last column / row
    h, w = i0.shape[1:]
    i0n, i1n = oracle.preprocess_local(i0, i1)
    A = oracle.Parts("ref", i0, i1, i0n, i1n, cols, rows)
    B = oracle.Parts("orc", i0, i1, i0n, i1n, cols, rows)
    assert A.count() == B.count() == cols * rows
    for p in range(A.count()):
        assert A.geometry(p) == B.geometry(p)
        for which in (0, 1):
            assert bits_equal(A.part_frame(p, which), B.part_frame(p, which))
    rng = np.random.default_rng(11)
    ids = list(range(15))
    for d in (0, 1):
        for pid in ids:
            v = rng.standard_normal((h, w)).astype(np.float32)
            if pid >= 13:
                v = (rng.random((h, w)) < 0.5).astype(np.float32)
            if pid in (11, 12):
                v[rng.random((h, w)) < 0.3] = np.nan
            if pid == 10:
                v = np.abs(v)
                v[rng.random((h, w)) < 0.2] = np.inf
            A.set_image(d, pid, v)
            B.set_image(d, pid, v)
        # candidates: heavy ties in the energy, positions all over the frame (also in the first partition, where the
        # reference's `i >= width ? i - off_x : i` leaves the index alone)
        for _ in range(400):
            i, j = int(rng.integers(0, w)), int(rng.integers(0, h))
            sim = float(rng.integers(0, 6)) * 0.25
            u, v = float(rng.standard_normal()), float(rng.standard_normal())
            A.push(d, i, j, u, v, sim)
            B.push(d, i, j, u, v, sim)
    A.transfer(True)
    B.transfer(True)
    assert A.any_empty() == B.any_empty()
    for p in range(A.count()):
        for d in (0, 1):
            for pid in ids:
                a, b = A.get_part(p, d, pid), B.get_part(p, d, pid)
                assert bits_equal(a, b), (p, d, pid, mismatch_report(a, b))
            qa, qb = A.drain(p, d), B.drain(p, d)
            assert qa.shape == qb.shape and np.array_equal(qa.view(np.uint32), qb.view(np.uint32)), (p, d)
    # the partitions change their planes; the merge must put every plane back where it belongs and empty the image queues
    for p in range(A.count()):
        pw, ph, _, _ = A.geometry(p)
        for d in (0, 1):
            for pid in ids:
                v = rng.standard_normal((ph, pw)).astype(np.float32)
                if pid >= 13:
                    v = (rng.random((ph, pw)) < 0.5).astype(np.float32)
                A.set_part(p, d, pid, v)
                B.set_part(p, d, pid, v)
    A.transfer(False)
    B.transfer(False)
    for d in (0, 1):
        assert A.image_queue_size(d) == B.image_queue_size(d) == 0
        for pid in ids:
            a, b = A.get_image(d, pid), B.get_image(d, pid)
            assert bits_equal(a, b), (d, pid, mismatch_report(a, b))
    # a grid whose partitions did not all receive candidates: anyEmptyQueues
    C1 = oracle.Parts("ref", i0, i1, i0n, i1n, cols, rows)
    C2 = oracle.Parts("orc", i0, i1, i0n, i1n, cols, rows)
    for P in (C1, C2):
        P.push(0, 1, 1, 0.0, 0.0, 1.0)
        P.push(1, 1, 1, 0.0, 0.0, 1.0)
        P.transfer(True)
    assert C1.any_empty() == C2.any_empty() == (cols * rows > 1)
