"""CPU suite, part 2: the C-ABI library builds for gfx950, loads, and exports every symbol
include/faldoi_hip.h declares.  No compute calls here (there is no GPU in CI), except to check
that they fail loudly -- the product has no CPU fallback."""
import ctypes
import os
import re

import numpy as np
import pytest

import faldoi_ipol_amd as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    L = F.capi.lib()
    header = open(os.path.join(ROOT, "include", "faldoi_hip.h")).read()
    declared = set(re.findall(r"\b(faldoi_[a-z0-9_]+)\s*\(", header))
    declared -= {"faldoi_ctx", "faldoi_tvl1_batch"}
    assert declared == set(F.capi.EXPORTS)
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert L.faldoi_abi_version() == 1


def test_schedule_selection_is_host_logic_and_picks_what_was_measured(monkeypatch):
    # which kernel path a solve takes depends on sizes and the CU count alone (faldoi_tvl1_schedule_query touches no device):
    # one pair that fits the chip -> ONE persistent launch, 64x48 tiles, 5 iterations between halo exchanges (both stop modes);
    # batches and frames with more tiles than CUs -> a launch per 5 iterations of 64x32 tiles; FALDOI_PERSIST=0 -> per launch
    monkeypatch.delenv("FALDOI_PERSIST", raising=False)
    monkeypatch.delenv("FALDOI_PERSIST_CFG", raising=False)
    monkeypatch.delenv("FALDOI_TILE", raising=False)
    q = F.capi.schedule_query
    assert q(1, 1024, 436, warps=300, max_inner=10, tol_stop=0) == (True, 3, 16, 5, 247)       # BASELINE configs[1]
    assert q(1, 1024, 436) == (True, 3, 16, 5, 247)                                              # the reference's defaults
    assert q(1, 1024, 436, compute_units=128, tol_stop=0) == (False, 3, 12, 5, 323)               # does not fit half a chip
    assert q(4, 1024, 436, tol_stop=0) == (False, 3, 12, 5, 4 * 323)                              # BASELINE configs[3] shape
    assert q(1, 1920, 1080, tol_stop=0)[:4] == (False, 3, 12, 5)                                 # more tiles than CUs
    assert q(1, 256, 192, tol_stop=0) == (True, 3, 16, 5, 5 * 6)
    assert q(1, 1024, 436, max_inner=3, tol_stop=0) == (True, 3, 16, 3, 18 * 11)                 # chunk = the iteration bound
    assert q(1, 1024, 436, chunk=16) == (False, 4, 16, 16, 32 * 14)                              # an explicit chunk the tiles cannot host
    monkeypatch.setenv("FALDOI_PERSIST", "0")
    assert q(1, 1024, 436, warps=300, max_inner=10, tol_stop=0) == (False, 4, 16, 10, 240)


def test_default_params_are_the_reference_constants():
    p = F.default_params()
    # src/parameters.h:20-22,31,43,51
    assert (p.lambda_, p.warps, p.max_inner, p.tol_stop) == (40.0, 5, 400, 1)
    assert p.theta == np.float32(0.3) and p.tau == np.float32(0.125) and p.tol == np.float32(0.01)


def _gpu_present():
    n = ctypes.c_int(0)
    return F.capi.lib().faldoi_device_count(ctypes.byref(n)) == 0 and n.value > 0


def test_no_device_is_a_loud_error_not_a_fallback():
    if _gpu_present():
        pytest.skip("a GPU is present")
    with pytest.raises(F.FaldoiError) as e:
        F.Context(0)
    assert e.value.code == F.capi.ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "faldoi-ipol_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.lower().replace("oracle-verified", ""), fn


def test_flo_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    f = rng.standard_normal((2, 5, 7)).astype(np.float32)
    f[0, 0, 0] = np.nan                                     # sparse flows use NaN for "no seed"
    p = tmp_path / "a.flo"
    F.flo.write_flo(str(p), f)
    raw = p.read_bytes()
    assert raw[:4] == b"PIEH" and len(raw) == 12 + 8 * 35
    g = F.flo.read_flo(str(p))
    assert np.array_equal(f.view(np.uint32), g.view(np.uint32))
    # interleaved layout: first pixel's (u, v)
    assert np.frombuffer(raw, "<f4", 2, 12)[1] == f[1, 0, 0]
    with pytest.raises(ValueError):
        (tmp_path / "bad.flo").write_bytes(b"PIEHxx")
        F.flo.read_flo(str(tmp_path / "bad.flo"))


def test_image_to_lab_host_function_matches_the_oracle(oracle):
    # host-side table code of the library (libm): same bits as the checker's restatement
    rng = np.random.default_rng(3)
    rgb = np.floor(rng.random((3, 19, 23)) * 256).astype(np.float32)
    rgb[:, 0, :3] = [[0, 255, 1], [0, 255, 2], [0, 255, 3]]
    a = F.capi.image_to_lab(rgb)
    b = oracle.image_to_lab(rgb)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_candidate_queue_pops_like_std_priority_queue(oracle):
    # host-only: the product's heap against the container the reference uses, with many equal energies
    import ctypes as C
    import numpy as np
    import faldoi_ipol_amd as F
    L = oracle.lib()
    rng = np.random.default_rng(5)
    for levels, n in ((3, 200), (8, 4000), (1000, 20000), (2, 65000)):
        ops = (rng.random(n) < 0.62).astype(np.int32)
        ops[:10] = 1
        sims = rng.integers(0, levels, n).astype(np.float32) * 0.25
        got = F.capi.queue_replay(ops, sims)
        q = L.orc_queue_new()
        want = []
        i, j, u, v, s = C.c_int(), C.c_int(), C.c_float(), C.c_float(), C.c_float()
        for k in range(n):
            if ops[k]:
                L.orc_queue_push(q, k, 0, 0.0, 0.0, float(sims[k]))
            else:
                want.append(i.value if L.orc_queue_pop(q, C.byref(i), C.byref(j), C.byref(u), C.byref(v), C.byref(s)) else -1)
        L.orc_queue_free(q)
        assert list(got) == want, (levels, n)


def test_binding_rejects_arrays_of_the_wrong_shape_before_the_library_sees_them():
    """The C ABI takes raw pointers and one (nx, ny): a frame, flow or Lab image of another size would be read past its end
    inside the library -- the host process dies of a segmentation fault without a message (the unexplained core dump of a
    round-3 scratch script, gpurun_out/nl_cases.log, fits exactly that: no GPU fault line, no exception text, at the script's
    first randomly cropped case).  The ctypes binding therefore checks every array against the frame's shape; no GPU needed."""
    i0 = np.zeros((20, 30), np.float32)
    lab = np.zeros((3, 20, 30), np.float32)
    u0 = np.zeros((2, 20, 30), np.float32)

    class Stub(F.capi.Context):                       # no device: the checks run before the library is called
        def __init__(self):
            self._h = None

        def close(self):
            pass
    c = Stub()
    with pytest.raises(ValueError, match="i1n has shape"):
        c.nltvcsad_global(i0, np.zeros((20, 31), np.float32), lab, u0)
    with pytest.raises(ValueError, match="lab has shape"):
        c.nltvcsad_global(i0, i0, np.zeros((3, 21, 30), np.float32), u0)
    with pytest.raises(ValueError, match="u0 has shape"):
        c.nltvl1_global(i0, i0, lab, np.zeros((2, 20, 29), np.float32))
    with pytest.raises(ValueError, match="u0 has shape"):
        c.tvl1_global(i0, i0, np.zeros((2, 19, 30), np.float32))
    with pytest.raises(ValueError, match="xi has shape"):
        c.tvcsad_global(i0, i0, u0, xi=np.zeros((4, 20, 31), np.float32))
    with pytest.raises(ValueError, match="i1 has shape"):
        c.preprocess_global(np.zeros((3, 20, 30), np.float32), np.zeros((3, 20, 31), np.float32))
    with pytest.raises(ValueError, match="planar RGB"):
        F.capi.image_to_lab(np.zeros((20, 30), np.float32))
