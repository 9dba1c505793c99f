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


def test_lab_and_nonlocal_tables_match_the_reference(oracle, R):
    """rgb_to_lab (src/energy_model.cpp:60), nltv_ini_dual_variables (src/aux_energy_model.cpp:112: weights, normaliser,
    neighbour and mirror-slot tables) and csad_ini_pos_nei (:216) -- the tables prepare_stuff builds for -m 2..7."""
    i0, _ = sintel_pair(crop=(500, 120, 41, 29))
    h, w = i0.shape[1:]
    rng = np.random.default_rng(2)
    imgs = [i0, np.floor(rng.random((3, h, w)) * 256).astype(np.float32), np.zeros((3, h, w), np.float32)]
    L = oracle.lib()
    n = h * w
    for rgb in imgs:
        a, b = np.empty_like(rgb), np.empty_like(rgb)
        R.ref_rgb_to_lab(rgb, n, a)
        L.orc_rgb_to_lab_local(rgb, n, b)
        assert bits_equal(a, b), mismatch_report(a, b)
        sc, wp = np.empty((24, h, w), np.float32), np.empty((24, h, w), np.float32)
        api, apj, rp = (np.empty((24, h, w), np.int32) for _ in range(3))
        wt = np.empty((h, w), np.float32)
        R.ref_nltv_table(a, 3, w, h, sc, wp, api, apj, rp, wt)
        owp, owt = np.empty((24, h, w), np.float32), np.empty((h, w), np.float32)
        L.orc_nltv_table_local(a, 3, w, h, owp, owt)
        valid = api >= 0
        assert bits_equal(np.where(valid, wp, 0), owp) and bits_equal(wt, owt)
        # the implicit tables: slot s <-> offset (k, l) in row-major order without the centre, mirror slot 23 - s, duals 0
        s = 0
        yy, xx = np.mgrid[0:h, 0:w]
        for k in range(-2, 3):
            for l in range(-2, 3):
                if k == 0 and l == 0:
                    continue
                inside = (xx + l >= 0) & (xx + l < w) & (yy + k >= 0) & (yy + k < h)
                assert np.array_equal(valid[s], inside)
                assert np.array_equal(api[s][inside], (xx + l)[inside]) and np.array_equal(apj[s][inside], (yy + k)[inside])
                assert (rp[s][inside] == 23 - s).all() and (sc[s][inside] == 0).all() and (sc[s][~inside] == -2).all()
                s += 1
    capi, capj, cnt = np.empty((48, h, w), np.int32), np.empty((48, h, w), np.int32), np.empty((h, w), np.int32)
    R.ref_csad_table(w, h, capi, capj, cnt)
    s = 0
    total = np.zeros((h, w), np.int32)
    for k in range(-3, 4):
        for l in range(-3, 4):
            if k == 0 and l == 0:
                continue
            inside = (xx + l >= 0) & (xx + l < w) & (yy + k >= 0) & (yy + k < h)
            assert np.array_equal(capi[s] >= 0, inside) and np.array_equal(capi[s][inside], (xx + l)[inside])
            assert np.array_equal(capj[s][inside], (yy + k)[inside])
            total += inside
            s += 1
    assert np.array_equal(cnt, total)


@pytest.mark.parametrize("method", [2, 3, 4, 5, 6, 7])
def test_other_local_functionals_as_called(oracle, R, method):
    """prepare_stuff + of_estimation of the reference for -m 2..7 (NLTV-L1, TV-CSAD, NLTV-CSAD and their weighted forms),
    driven like add_neighbors drives them, against the oracle: frames, window energies, flow and -- for the non-local
    functionals -- every dual plane after the sequence, forward and backward."""
    import ctypes as C
    i0, i1 = sintel_pair(crop=(430, 160, 60, 44))
    h, w = i0.shape[1:]
    n = h * w
    L = oracle.lib()
    op = oracle.RefOp(method, w, h, wr=5)
    r0, r1 = op.prepare(i0, i1)
    # the frames prepare_stuff hands on: -m 2 builds BOTH from i0 (src/energy_model.cpp:318-319), the others from i0 / i1
    o0, o1 = oracle.preprocess_local(i0, i0 if method == 2 else i1)
    assert bits_equal(r0, o0) and bits_equal(r1, o1)
    nl = method in (2, 3, 6, 7)
    lab = np.empty_like(i0)
    L.orc_rgb_to_lab_local(i0, n, lab)                       # both directions use the Lab image of i0 (:322-323 etc.)
    wp, wt = np.zeros((24, h, w), np.float32), np.zeros((h, w), np.float32)
    if nl:
        L.orc_nltv_table_local(lab, 3, w, h, wp, wt)
    weight = np.zeros(11, np.float32)
    L.orc_gaussian1d_weight(weight, 5)
    rng = np.random.default_rng(40 + method)
    for d, (src, dst) in enumerate(((o0, o1), (o1, o0))):
        gx, gy = oracle.centered_gradient(dst)
        assert bits_equal(op.plane(d, "I1x"), gx) and bits_equal(op.plane(d, "I1y"), gy)
        u = (rng.standard_normal((2, h, w)) * 1.2).astype(np.float32)
        op.set_flow(d, u)
        duals = np.zeros((48, h, w), np.float32)
        for k, (i, j) in enumerate(_windows(rng, w, h, 25)):
            wr = 1 if k % 3 == 0 else 5
            ii, ij, ei, ej = oracle.patch_index(wr, w, h, i, j)
            e_ref = op.estimate(d, i, j, wr)
            it = C.c_int(0)
            e = L.orc_local_nl_estimate(method, src, dst, gx, gy, u[0], u[1], duals.ctypes.data if nl else None,
                                        wp.ctypes.data if nl else None, w, h, ii, ij, ei, ej, 4, weight, max(0, -(i - wr)),
                                        max(0, -(j - wr)), wr, C.byref(it))
            assert np.float32(e).view(np.uint32) == np.float32(e_ref).view(np.uint32), (method, d, k, i, j, wr, e, e_ref)
            got = op.get_flow(d)
            assert bits_equal(got, u), (method, d, k, (i, j, wr), mismatch_report(got, u))
        if nl:
            for which in (0, 1):
                sc, rwp, _ = op.duals(d, which)
                valid = sc != -2.0                           # slots outside the image keep the reference's marker
                mine = duals[24 * which:24 * which + 24]
                assert bits_equal(np.where(valid, sc, 0), np.where(valid, mine, 0)), (method, d, which)
                assert bits_equal(np.where(valid, rwp, 0), wp)
    op.close()
