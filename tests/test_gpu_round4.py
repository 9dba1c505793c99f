"""GPU suite, round 4: the CSAD selection by the warp's rank table (six probes instead of a 48-input sort per pixel and
iteration) against the sorting-network form it replaces and against the CPU checker, which sorts like the reference
(src/global_faldoi.cpp:1733-1758, :1547-1571).  Everything goes through the C ABI."""
import numpy as np
import pytest

import faldoi_ipol_amd as F
from util import bits_equal, mismatch_report, sintel_pair

pytestmark = pytest.mark.gpu
W, H = 1024, 436


def _crop_inputs(oracle, rng, w, h):
    full0, full1 = sintel_pair()
    x0, y0 = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
    c0 = np.ascontiguousarray(full0[:, y0:y0 + h, x0:x0 + w])
    c1 = np.ascontiguousarray(full1[:, y0:y0 + h, x0:x0 + w])
    lab = F.capi.image_to_lab(c0)
    if min(w, h) >= 6:
        i0n, i1n = oracle.preprocess_global(c0, c1)
    else:                                        # smaller than the Gaussian's support: gray planes as they are
        i0n, i1n = c0.mean(axis=0) / np.float32(255), c1.mean(axis=0) / np.float32(255)
    return i0n, i1n, lab


def _degenerate(rng, kind, w, h):
    """Frames that make the CSAD terms tie: flat regions (every term of a pixel equal, b == r happens: -0 among the sorted
    values), two grey levels, a ramp (equal terms along rows), and frames with exact zeros."""
    if kind == 0:
        a = np.zeros((h, w), np.float32)
        a[:, w // 2:] = 1.0
        b = np.roll(a, 1, axis=1)
    elif kind == 1:
        a = np.tile(np.arange(w, dtype=np.float32) / np.float32(w), (h, 1))
        b = np.roll(a, 2, axis=1)
    elif kind == 2:
        a = (rng.integers(0, 2, (h, w))).astype(np.float32)
        b = np.roll(a, 1, axis=0)
    else:
        a = (rng.integers(0, 3, (h, w)) * 0.5).astype(np.float32)
        b = (rng.integers(0, 3, (h, w)) * 0.5).astype(np.float32)
    lab = np.stack([a * 100, a * 0, a * 0]).astype(np.float32)
    return np.ascontiguousarray(a), np.ascontiguousarray(b), lab


def test_csad_selection_by_ranks_equals_the_sorting_network_and_the_checker(ctx, oracle, monkeypatch):
    """NLTV-CSAD on the fused kernel and on the two streaming kernels, TV-CSAD: the rank-table selection (product path) returns
    the bits of the per-iteration sorting network (FALDOI_CSAD_SORT=1, the round-3 form) -- including the sign of a zero --
    on crops of the example pair (every tile height, ragged tiles, images smaller than the 7x7 window, several warps: the
    table is rebuilt per warp) and on frames built to make the 48 terms tie; small cases also equal the CPU checker."""
    rng = np.random.default_rng(40401)
    checked = 0
    for case in range(36):
        if case < 6:
            w, h = [(2, 2), (3, 7), (64, 16), (65, 17), (130, 35), (7, 40)][case]
        elif case < 14:
            w, h = int(rng.integers(8, 90)), int(rng.integers(8, 60))
        else:
            w, h = int(rng.integers(6, 420)), int(rng.integers(6, 150))
        if 6 <= case < 14:
            i0n, i1n, lab = _degenerate(rng, case % 4, w, h)
        else:
            i0n, i1n, lab = _crop_inputs(oracle, rng, w, h)
        sd = float(rng.choice([0, 0.5, 3.0]))
        u0 = rng.normal(0, sd, (2, h, w)).astype(np.float32)
        if case % 5 == 0:
            u0 = np.round(u0)                    # integer flows: exact zeros among the terms
        prm = dict(warps=int(rng.integers(1, 4)), max_inner=int(rng.integers(1, 14)))
        tv = case % 3 == 2
        if tv:
            def solve():
                return ctx.tvcsad_global(i0n, i1n, u0, warps=prm["warps"], max_inner=prm["max_inner"])[0]
        else:
            def solve():
                return ctx.nltvcsad_global(i0n, i1n, lab, u0, **prm)[0]
        monkeypatch.setenv("FALDOI_CSAD_SORT", "1")
        ref = solve()
        monkeypatch.delenv("FALDOI_CSAD_SORT")
        outs = []
        if tv:
            outs.append(("tvcsad", solve()))
        else:
            monkeypatch.setenv("FALDOI_NL_FUSED", "0")
            outs.append(("two kernels", solve()))
            monkeypatch.setenv("FALDOI_NL_FUSED", "1")
            rows = [None, 8, 12, 14, 16][case % 5]
            if rows:
                monkeypatch.setenv("FALDOI_NL_ROWS", str(rows))
            else:
                monkeypatch.delenv("FALDOI_NL_ROWS", raising=False)
            outs.append(("fused rows=%s" % rows, solve()))
            monkeypatch.delenv("FALDOI_NL_FUSED")
            monkeypatch.delenv("FALDOI_NL_ROWS", raising=False)
        for name, got in outs:
            assert bits_equal(got, ref), (case, name, w, h, prm, mismatch_report(got, ref))
        if w * h <= 6000 and not tv:
            want = oracle.nltvcsad_global(i0n, i1n, lab, u0, **prm)
            want = want[0] if isinstance(want, tuple) else want
            # (the checker's std::sort may order a tied -0 / +0 pair either way: compare values, and bits away from zeros)
            got = outs[-1][1]
            assert np.array_equal(got, want), (case, w, h, mismatch_report(got, want))
            checked += 1
    assert checked >= 8


def test_fused_nltv_kernel_range_checked_build_is_clean(ctx):
    """libfaldoi_hip_check.so (build.build_check(): the same sources with -DFALDOI_NL_CHECK=1) compares every LDS index and every slab
    offset the fused NLTV kernel forms with the extent it must stay inside (nltv_fused_kernel.hpp, nf_check).  The geometry sweep
    of the fused-kernel tests -- images smaller than a tile, ragged last tiles, one to seven tile columns, every tile height, both
    data terms -- and the full frame run on it ONCE, in a child process: no violation, the same flows as the product build, and the
    detector is shown not to be vacuous (a deliberate out-of-range index of each kind is reported)."""
    import json
    import os
    import subprocess
    import sys
    so = F.build.SO_CHECK
    assert os.path.exists(so), "range-checked library missing: __graft_entry__.build() makes it"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "nl_check_run.py")
    env = dict(os.environ, FALDOI_LIB=so)
    env.pop("FALDOI_NL_FUSED", None)
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["compiled_in"] and rep["library"].endswith("libfaldoi_hip_check.so")
    assert rep["launches"] > 500, rep
    assert rep["bits"] == 0 and rep["violations"] == 0, rep
    assert rep["selftest_violations"] >= 5 and bin(rep["selftest_bits"]).count("1") >= 4, rep
    # the same sweep on the product library (this process): identical flows
    r2 = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=900,
                        env={k: v for k, v in os.environ.items() if k != "FALDOI_LIB"})
    assert r2.returncode == 0, r2.stderr[-3000:]
    rep2 = json.loads(r2.stdout.strip().splitlines()[-1])
    assert not rep2["compiled_in"] and rep2["bits"] == 0 and rep2["sha256"] == rep["sha256"], (rep, rep2)


# ------------------------------------------------------------------ -partial_res (src/local_faldoi.cpp:942-1036)
def _partial_case(oracle, w, h, x0, y0, seed):
    c0, c1 = sintel_pair((x0, y0, w, h))
    a, b = oracle.preprocess_local(c0, c1)
    rng = np.random.default_rng(seed)
    go = np.full((2, h, w), np.nan, np.float32)
    ba = np.full((2, h, w), np.nan, np.float32)
    m = rng.random((h, w)) < 0.02
    go[:, m] = np.round(rng.normal(0, 1.5, (2, int(m.sum())))).astype(np.float32)
    m = rng.random((h, w)) < 0.02
    ba[:, m] = np.round(rng.normal(0, 1.5, (2, int(m.sum())))).astype(np.float32)
    return a, b, go, ba


def test_partial_results_of_the_forward_growings_match_the_cpu_path(ctx, oracle, tmp_path):
    """-partial_res 1: every forward growing hands out its flow the first time the share of pixels it has fixed passes 30 / 70 / 80 /
    95 % (the reference's float test, so thresholds can be skipped on small domains) and when its queue is empty.  The device
    growing stops at exactly those window counts (the launch budget is cut to land on them) -- the same set of dumps, bit for bit
    (NaN where nothing is fixed yet), as the CPU restatement of local_growing; the final flow is unchanged by the stops.  Whole
    image and the reference's partition mode; then the executable writes the reference's file names."""
    import subprocess
    for (w, h, x0, y0, split, budget) in ((96, 64, 300, 100, 0, 16384), (96, 64, 300, 100, 0, 700), (120, 80, 500, 200, 1, 16384),
                                         (13, 9, 40, 40, 0, 5)):
        a, b, go, ba = _partial_case(oracle, w, h, x0, y0, 5 + w)
        kw = dict(split_img=split, h_parts=3, v_parts=2) if split else {}
        want_flow, want_ene, _, want = oracle.local_faldoi(a, b, go, ba, partial_res=True, **kw)
        got = {}
        flow, ene, _ = ctx.match_growing(a, b, go, ba, preprocessed=1, budget=budget, partial_res=got, **kw)
        plain, _, _ = ctx.match_growing(a, b, go, ba, preprocessed=1, **kw)
        assert bits_equal(flow, want_flow) and bits_equal(ene, want_ene) and bits_equal(flow, plain)
        got = {k[1:]: v for k, v in got.items()}                     # one pair: drop the pair index
        assert sorted(got) == sorted(want), (w, h, split, sorted(got), sorted(want))
        assert any(k[0] == 100 for k in got) and any(k[0] < 100 for k in got)
        for k in want:
            assert bits_equal(got[k], want[k]), (w, h, split, k, mismatch_report(got[k], want[k]))
        if split:
            assert any(k[2] >= 0 and k[1] > 0 for k in got) and any(k[2] == -1 for k in got)
    # the executable: the reference's names under -partial_dir (additive; default ../Results/Partial_results/)
    from PIL import Image
    w, h, x0, y0 = 96, 64, 300, 100
    c0, c1 = sintel_pair((x0, y0, w, h))
    a, b, go, ba = _partial_case(oracle, w, h, x0, y0, 5 + w)
    for k, im in enumerate((c0, c1)):
        Image.fromarray(im.transpose(1, 2, 0).astype(np.uint8), "RGB").save(tmp_path / ("f%d.png" % k))
    (tmp_path / "ims.txt").write_text("%s\n%s\n" % (tmp_path / "f0.png", tmp_path / "f1.png"))
    for name, f in (("go.flo", go), ("ba.flo", ba)):
        (tmp_path / name).write_bytes(F.flo.flo_bytes(f))
    (tmp_path / "part").mkdir()
    exe = F.build.build_host()["local_faldoi"]
    r = subprocess.run([exe, str(tmp_path / "ims.txt"), str(tmp_path / "go.flo"), str(tmp_path / "ba.flo"), str(tmp_path / "out.flo"),
                        str(tmp_path / "sim.tiff"), "-partial_res", "1", "-partial_dir", str(tmp_path / "part")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a2, b2 = oracle.preprocess_local(c0, c1)
    _, _, _, want = oracle.local_faldoi(a2, b2, go, ba, partial_res=True)
    names = sorted(p.name for p in (tmp_path / "part").iterdir())
    assert names == sorted("partial_results_fwd_%d_iter_%d.flo" % (pc, it) for (pc, it, _) in want), names
    for (pc, it, _), flow in want.items():
        assert (tmp_path / "part" / ("partial_results_fwd_%d_iter_%d.flo" % (pc, it))).read_bytes() == F.flo.flo_bytes(flow)


def test_bench_per_rank_parity_path_rehearsed_on_one_gpu():
    """BASELINE configs[3] is 8 ranks x 4 pairs; no 8-GPU node has been available to the driver, so the per-rank evidence a SCALE
    line would carry (bench.py: per_rank_check -- a checksum of every rank's resident flow and a bounded bit comparison with the
    CPU checker) is exercised here on the one GPU: `bench.py --pairs-per-gpu 4` with FALDOI_BENCH_RANK_CHECK=1, as a child."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FALDOI_BENCH_RANK_CHECK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID", "FALDOI_LIB"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--pairs-per-gpu", "4", "--steps", "2", "--warmup", "1",
                        "--no-secondary"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["pairs_per_gpu"] == 4 and line["scaling"] == "weak"
    assert line["config"]["kernel_path"] == "one launch per chunk of iterations"
    chk = line["per_rank_check"]
    assert len(chk) == 1 and chk[0]["rank"] == 0 and len(chk[0]["flow_sha256_48"]) == 12
    assert chk[0]["bit_identical_to_cpu_first_30_warps"] is True
    assert line["epe_vs_cpu"]["bit_identical"] is True and line["value"] > 50000
    errs = {k: v for k, v in (line.get("secondary_errors") or {}).items() if k != "counters"}     # (a stale profile is not this test's matter)
    assert not errs, errs


def test_all_eighteen_reference_pairs_match_the_checker_reference_defaults(ctx, oracle):
    """SURVEY 8d's configs[3] input: the reference's six example sequences x frames 1->2, 2->3, 3->4 (committed fixtures), zero
    initial flow, the reference's default schedule (5 warps, <= 400 iterations, tol 0.01).  All eighteen 1024x436 pairs in ONE
    batch on the device (the pairs stop independently): per-pair iteration counts, errors, flow and duals bit-equal to the CPU
    checker from the RGB frames on (device-side preprocessing included)."""
    from util import load_rgb_planar, reference_pairs
    pairs = reference_pairs()
    assert len(pairs) == 18
    b = F.Tvl1Batch(ctx, len(pairs), W, H)
    try:
        frames = []
        for i, (name, fa, fb) in enumerate(pairs):
            i0, i1 = load_rgb_planar(fa), load_rgb_planar(fb)
            frames.append((i0, i1))
            b.upload_images(i, i0, i1, np.zeros((2, H, W), np.float32))
        iters, errs, ms, launches = b.solve()
        assert not b.last_persistent()
        counts = []
        for i, (name, _, _) in enumerate(pairs):
            i0n, i1n = oracle.preprocess_global(*frames[i])
            ou, oit, oerr, oxi = oracle.tvl1_global(i0n, i1n, np.zeros((2, H, W), np.float32))
            u, xi = b.download(i, want_xi=True)
            assert list(iters[i]) == list(oit), (name, iters[i], oit)
            assert bits_equal(errs[i], oerr), name
            assert bits_equal(u, ou), (name, mismatch_report(u, ou))
            assert bits_equal(xi, oxi), (name, mismatch_report(xi, oxi))
            counts.append(int(iters[i].sum()))
        print("18 reference pairs, reference defaults: %.1f ms on the device, iterations per pair %s" % (ms, counts))
        assert len(set(counts)) > 6
    finally:
        b.close()


def test_growing_with_several_windows_at_a_time_is_the_sequential_growing(ctx, oracle, monkeypatch):
    """grow_twin_kernel.hpp: one growing computes up to K windows side by side where the pop order allows it (the next candidates of
    the queue, guessed after a pop, held back, and committed only if they ARE the next pops once the earlier windows' candidates are
    in).  K = 2 and K = 4 against the plain kernel (FALDOI_GROW_TWIN=0) and the CPU checker: flow, energy map and the counts of
    pops / windows / fills identical -- whole image (plain and weighted functional), with saliency maps, with a tiny launch budget
    (the guesses must respect it), on a tiny image."""
    rng = np.random.default_rng(9)
    for (w, h, x0, y0, kw) in ((160, 96, 300, 100, {}), (120, 80, 600, 250, dict(method=1)), (96, 64, 100, 300, dict(budget=7)),
                               (9, 7, 50, 50, {}), (140, 90, 400, 60, dict(sal=True))):
        a, b, go, ba = _partial_case(oracle, w, h, x0, y0, 21 + w)
        okw = {k: v for k, v in kw.items() if k in ("method",)}
        gkw = {k: v for k, v in kw.items() if k in ("method", "budget")}
        sal = (rng.random((h, w)).astype(np.float32) + 0.5, rng.random((h, w)).astype(np.float32) + 0.5) if kw.get("sal") else (None, None)
        want_flow, want_ene, want_st = oracle.local_faldoi(a, b, go, ba, sal0=sal[0], sal1=sal[1], **okw)
        res = {}
        for k in ("0", "2", "4"):
            monkeypatch.setenv("FALDOI_GROW_TWIN", k)
            flow, ene, st = ctx.match_growing(a, b, go, ba, sal[0], sal[1], preprocessed=1, **gkw)
            res[k] = (flow, ene, (list(st.pops), list(st.solves), list(st.fills)))
            assert bits_equal(flow, want_flow), (w, h, kw, k, mismatch_report(flow, want_flow))
            assert bits_equal(ene, want_ene), (w, h, kw, k)
            assert [int(v) for v in want_st] == [st.pops[0], st.solves[0], st.fills[0], st.pops[1], st.solves[1], st.fills[1]], (k, list(want_st), res[k][2])
        monkeypatch.delenv("FALDOI_GROW_TWIN")
        assert res["0"][2] == res["2"][2] == res["4"][2]


@pytest.mark.gpu
@pytest.mark.parametrize("top_levels", [0, 2, 5, 9, 13])
def test_device_queue_rounds_of_six_levels_and_in_register_placement(ctx, monkeypatch, top_levels):
    # a pop walks six levels per trip (one vector comparison decides all 63 nodes, the path is marked lane-parallel; a second register
    # holds depth 6) and places the last element from the path in registers; every split between LDS and global levels, ties, long
    # runs of pops down to the empty queue, against the host queue (itself tested against std::priority_queue)
    monkeypatch.setenv("FALDOI_HEAP_TOP_LEVELS", str(top_levels))
    rng = np.random.default_rng(40 + top_levels)
    for levels, n, ppush in ((3, 300, 0.6), (100000, 65000, 0.75), (7, 65000, 0.66), (1000, 40000, 0.52), (100000, 3000, 0.5)):
        ops = (rng.random(n) < ppush).astype(np.int32)
        ops[: n // 3] |= (rng.random(n // 3) < 0.8).astype(np.int32)      # grow first, then drain: pops from every depth
        ops[-n // 4:] = 0
        sims = rng.integers(0, levels, n).astype(np.float32) * 0.25
        want = F.capi.queue_replay(ops, sims)
        got = ctx.queue_replay(ops, sims)
        assert np.array_equal(got, want), (top_levels, levels, n)
