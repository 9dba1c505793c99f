"""GPU suite: the HIP path, called through the C ABI, against the oracle on the same inputs.

Bar (BASELINE.json north_star): <= 1e-4 mean EPE for the floating-point solver.  This
implementation is held to the stricter bar the design allows -- BIT-IDENTICAL flow, duals,
per-warp iteration counts and errors -- and the EPE gate is asserted as well so a future
relaxation (e.g. fused multiply-add) has a written tolerance to answer to.
"""
import subprocess

import numpy as np
import pytest

import faldoi_ipol_amd as F
from util import (GOLDEN, bits_equal, flo_sha256, known_answers, mismatch_report, sintel_pair,
                  synthetic_pair)

pytestmark = pytest.mark.gpu
EPE_GATE = 1e-4      # mean end-point error, pixels (north_star)


def _rand(rng, h, w, s=1.0):
    return (rng.standard_normal((h, w)) * s).astype(np.float32)


# ------------------------------------------------------------------ primitives (SURVEY 8a A1,A2/3,A7,A9)
@pytest.mark.parametrize("h,w", [(2, 2), (5, 3), (23, 29), (64, 64), (97, 130)])
def test_stencil_kernels_bit_exact(ctx, oracle, h, w):
    L = oracle.lib()
    rng = np.random.default_rng(h * 1000 + w)
    a, b = _rand(rng, h, w, 30), _rand(rng, h, w, 30)
    o = [np.empty_like(a) for _ in range(2)]
    L.orc_centered_gradient(a, o[0], o[1], w, h)
    dx, dy = ctx.centered_gradient(a)
    assert bits_equal(dx, o[0]) and bits_equal(dy, o[1])
    L.orc_forward_gradient(a, o[0], o[1], w, h)
    fx, fy = ctx.forward_gradient(a)
    assert bits_equal(fx, o[0]) and bits_equal(fy, o[1])
    L.orc_divergence(a, b, o[0], w, h)
    assert bits_equal(ctx.divergence(a, b), o[0])


def test_primitives_match_reference_golden_vectors(ctx):
    g = np.load(GOLDEN + "/ref_primitives.npz")
    dx, dy = ctx.centered_gradient(g["img"])
    assert bits_equal(dx, g["cg_dx"]) and bits_equal(dy, g["cg_dy"])
    fx, fy = ctx.forward_gradient(g["img"])
    assert bits_equal(fx, g["fg_fx"]) and bits_equal(fy, g["fg_fy"])
    assert bits_equal(ctx.divergence(g["img"], g["img2"]), g["div"])
    for bo in (0, 1):
        assert bits_equal(ctx.bicubic_warp(g["img"], g["u"], g["v"], bo), g["warp_bo%d" % bo])


@pytest.mark.parametrize("border_out", [True, False])
def test_bicubic_warp_bit_exact_incl_out_of_range(ctx, oracle, border_out):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    h, w = 61, 83
    img = (rng.random((h, w)) * 255).astype(np.float32)
    u, v = _rand(rng, h, w, 6), _rand(rng, h, w, 6)
    u[0, :6] = [-0.5, -1.5, 0.25, 1e6, -1e6, 0.0]
    v[1, :6] = [-0.5, -1.5, 0.25, 1e6, -1e6, 0.0]
    u[10, 10], v[10, 10] = np.float32(1e-30), np.float32(-1e-30)     # denormal-ish offsets
    want = np.empty_like(img)
    L.orc_bicubic_warp(img, u, v, want, w, h, int(border_out))
    got = ctx.bicubic_warp(img, u, v, border_out)
    assert bits_equal(got, want), mismatch_report(got, want)


# ------------------------------------------------------------------ preprocessing
def test_preprocessing_bit_exact_sintel_and_random(ctx, oracle):
    i0, i1 = sintel_pair()
    g0, g1 = ctx.preprocess_global(i0, i1)
    o0, o1 = oracle.preprocess_global(i0, i1)
    assert bits_equal(g0, o0), mismatch_report(g0, o0)
    assert bits_equal(g1, o1), mismatch_report(g1, o1)
    rng = np.random.default_rng(5)
    for pd, h, w in ((1, 6, 6), (3, 17, 9), (1, 50, 300), (3, 64, 64)):
        a, b, c = (np.floor(rng.random((pd, h, w)) * 256).astype(np.float32) for _ in range(3))
        c += 3.0                                   # distinct minima: exercises the min-selection quirk
        g0, g1 = ctx.preprocess_global(a, b, c)
        o = [np.empty((h, w), np.float32) for _ in range(3)]
        oracle.lib().orc_preprocess_global(a, b, c, w, h, pd, o[0], o[1], o[2])
        assert bits_equal(g0, o[0]) and bits_equal(g1, o[1]), (pd, h, w)
    with pytest.raises(F.FaldoiError):
        ctx.preprocess_global(np.zeros((2, 8, 8), np.float32), np.zeros((2, 8, 8), np.float32))
    with pytest.raises(F.FaldoiError):
        ctx.preprocess_global(np.zeros((1, 4, 8), np.float32), np.zeros((1, 4, 8), np.float32))


def test_upload_images_then_solve_matches_known_answer(ctx):
    # the whole device-side data path of global_faldoi: RGB frames -> preprocessing -> solver
    case = known_answers()["sintel_clean_easy_f2_f3_crop_400_150_128x96_zero_init"]
    i0, i1 = sintel_pair((case["x0"], case["y0"], case["w"], case["h"]))
    b = F.Tvl1Batch(ctx, 1, case["w"], case["h"])
    b.upload_images(0, i0, i1, np.zeros((2, case["h"], case["w"]), np.float32))
    iters, errs, ms, nl = b.solve()
    assert list(iters[0]) == case["iters"]
    assert flo_sha256(b.download(0)).startswith(case["flo_sha256_prefix"])
    b.close()


def test_fp64_assisted_quotient_equals_ieee_divide(ctx):
    # 2^30 hashed samples x 2 call sites, all exponents incl. subnormals and tie makers
    assert ctx.selftest_division(count=1 << 30, seed=1) == 0
    assert ctx.selftest_division(count=1 << 26, seed=0xdeadbeef) == 0
    # the one-product form of the quotients by theta and |grad I|^2 (faldoi_device.hpp: div_once) is part of the same self-test: all
    # numerators for theta = 0.3f, hashed ones and tie makers for hashed divisors.  Its refusal of even-integer divisors is not
    # vacuous: the sample holds such divisors, and some of their tie quotients would indeed come out wrong.
    import re
    m = re.search(r"refused (\d+) quotients, (\d+) of them would differ", ctx.selftest_report)
    assert m and int(m.group(1)) > 1000 and int(m.group(2)) > 0, ctx.selftest_report


# ------------------------------------------------------------------ whole solver
def _compare_solve(ctx, oracle, i0n, i1n, u0, xi0=None, **prm):
    ou, oit, oerr, oxi = oracle.tvl1_global(i0n, i1n, u0, xi=xi0,
                                            lam=prm.get("lam", 40.0), theta=prm.get("theta", 0.3),
                                            tau=prm.get("tau", 0.125), tol=prm.get("tol", 0.01),
                                            warps=prm.get("warps", 5), max_inner=prm.get("max_inner", 400),
                                            tol_stop=prm.get("tol_stop", True))
    h, w = i0n.shape
    xin = np.zeros((4, h, w), np.float32) if xi0 is None else xi0
    gu, git, gerr, gxi, ms = ctx.tvl1_global(i0n, i1n, u0, xi=xin, **prm)
    assert list(git) == list(oit), (git, oit)
    mean_epe = float(F.flo.epe(gu, ou).mean())
    assert mean_epe <= EPE_GATE, mean_epe
    assert bits_equal(gu, ou), mismatch_report(gu, ou)
    assert bits_equal(gxi, oxi), mismatch_report(gxi, oxi)
    assert bits_equal(gerr, oerr), (gerr, oerr)
    return gu


@pytest.mark.parametrize("w,h", [(64, 48), (130, 97), (45, 200), (16, 16)])
@pytest.mark.parametrize("chunk", [0, 1, 7, 16])
def test_tvl1_reference_schedule_bit_exact_synthetic(ctx, oracle, w, h, chunk):
    i0n, i1n, gt = synthetic_pair(w, h, seed=w + h)
    rng = np.random.default_rng(w)
    u0 = (gt + rng.standard_normal(gt.shape) * 0.5).astype(np.float32)
    _compare_solve(ctx, oracle, i0n, i1n, u0, warps=3, max_inner=60, chunk=chunk)


def test_tvl1_fixed_schedule_bit_exact(ctx, oracle):
    # BASELINE config 2's schedule shape (many warps x 10 inner iterations, tolerance stop off)
    i0n, i1n, gt = synthetic_pair(256, 192, seed=5)
    _compare_solve(ctx, oracle, i0n, i1n, np.zeros_like(gt), warps=12, max_inner=10, tol_stop=0)
    _compare_solve(ctx, oracle, i0n, i1n, np.zeros_like(gt), warps=4, max_inner=23, tol_stop=0, chunk=10)


def _static_solve(ctx, i0n, i1n, u0, xi0, **prm):
    h, w = i0n.shape
    b = F.Tvl1Batch(ctx, 1, w, h)
    try:
        b.upload(0, i0n, i1n, u0, xi0)
        its, errs, _, launches = b.solve(tol_stop=0, **prm)
        u, xi = b.download(0, want_xi=True)
        return u, xi, its[0].copy(), errs[0].copy(), launches, b.last_persistent(), b.last_schedule()
    finally:
        b.close()


@pytest.mark.parametrize("w,h,warps,inner,cfg", [(192, 128, 3, 10, None), (300, 200, 4, 10, "3,16,5"), (300, 200, 3, 12, "4,13,5"),
                                                 (257, 131, 2, 9, "4,11,4"), (640, 360, 3, 10, None), (1024, 436, 4, 10, None),
                                                 (1024, 436, 2, 16, "4,16,8"), (64, 48, 2, 5, "3,14,3"), (70, 3, 2, 7, None),
                                                 (2, 2, 2, 4, None)])
def test_persistent_kernel_equals_per_launch_path_and_checker(ctx, oracle, monkeypatch, w, h, warps, inner, cfg):
    # a static schedule of one pair runs as ONE launch that keeps every tile in registers and only exchanges halos between
    # workgroups (tvl1_persist_kernel.hpp); FALDOI_PERSIST=0 is the launch-per-chunk path.  Same bits from both, and from the
    # CPU checker, for several tile geometries (R,NW,k), image sizes with ragged last tiles, and a non-zero start.
    i0n, i1n, gt = synthetic_pair(w, h, seed=w + h, amp=2.0)
    rng = np.random.default_rng(w * h)
    u0 = (0.5 * gt).astype(np.float32)
    xi0 = (rng.standard_normal((4, h, w)) * 0.6).astype(np.float32)
    ou, oit, oerr, oxi = oracle.tvl1_global(i0n, i1n, u0, xi=xi0, lam=40.0, theta=0.3, tau=0.125, tol=0.01, warps=warps,
                                            max_inner=inner, tol_stop=False)
    monkeypatch.setenv("FALDOI_PERSIST", "0")
    ref = _static_solve(ctx, i0n, i1n, u0, xi0, warps=warps, max_inner=inner)
    assert not ref[5]
    monkeypatch.setenv("FALDOI_PERSIST", "1")
    if cfg:
        monkeypatch.setenv("FALDOI_PERSIST_CFG", cfg)
    new = _static_solve(ctx, i0n, i1n, u0, xi0, warps=warps, max_inner=inner)
    assert new[5] and new[4] == 1, "the persistent path was not taken"
    for got in (ref, new):
        assert bits_equal(got[0], ou), mismatch_report(got[0], ou)
        assert bits_equal(got[1], oxi), mismatch_report(got[1], oxi)
        assert list(got[2]) == list(oit) and bits_equal(got[3], oerr)


@pytest.mark.parametrize("w,h,warps,inner,tol,chunk", [(300, 200, 4, 60, 0.05, 0), (1024, 436, 3, 45, 0.05, 0), (640, 360, 3, 80, 0.03, 0),
                                                      (192, 128, 5, 400, 0.01, 0), (257, 131, 3, 33, 0.05, 3), (257, 131, 3, 120, 0.02, 0),
                                                      (70, 3, 2, 20, 0.01, 7)])
def test_persistent_kernel_with_the_tolerance_stop(ctx, oracle, monkeypatch, w, h, warps, inner, tol, chunk):
    # the reference's stop test (max |du|^2 <= tol^2 over the whole image) decided by the resident tiles themselves, every chunk:
    # same iteration counts, errors, flow and duals as the launch-per-chunk path with its control kernel, and as the CPU checker;
    # stops in the middle of a chunk (exact-length re-run from the LDS snapshot), at its end, and never (iteration bound)
    i0n, i1n, gt = synthetic_pair(w, h, seed=w * 3 + h, amp=2.0)
    u0 = (0.7 * gt).astype(np.float32)
    ou, oit, oerr, oxi = oracle.tvl1_global(i0n, i1n, u0, lam=40.0, theta=0.3, tau=0.125, tol=tol, warps=warps, max_inner=inner,
                                            tol_stop=True)
    out = {}
    for persist in ("0", "1"):
        monkeypatch.setenv("FALDOI_PERSIST", persist)
        b = F.Tvl1Batch(ctx, 1, w, h)
        try:
            b.upload(0, i0n, i1n, u0)
            its, errs, _, launches = b.solve(warps=warps, max_inner=inner, tol=tol, chunk=chunk)
            u, xi = b.download(0, want_xi=True)
            out[persist] = (u, xi, its[0].copy(), errs[0].copy(), b.last_persistent())
        finally:
            b.close()
    assert not out["0"][4] and out["1"][4], "the persistent path was not taken"
    for got in out.values():
        assert list(got[2]) == list(oit), (got[2], oit)
        assert bits_equal(got[0], ou), mismatch_report(got[0], ou)
        assert bits_equal(got[1], oxi) and bits_equal(got[3], oerr)
    # (the cases stop after 24 6 5 5 | 23 9 6 | 80 32 14 | 285 73 56 55 55 | 24 8 5 | 120 89 14 | 20 20 iterations: inside a chunk
    #  of five, on its last iteration, and at the iteration bound)


def test_persistent_kernel_random_cases_equal_the_per_launch_path(ctx, monkeypatch):
    # a bounded slice of tests/tools/fuzz_parity.py --family persist (which has run over 10 000 cases): random sizes up to the
    # full frame, schedules, parameters, start flows and duals, both stop modes -- the two kernel paths must agree bit for bit
    rng = np.random.default_rng(20261004)
    took = 0
    for case in range(36):
        w, h = (int(rng.integers(2, 1100)), int(rng.integers(2, 480))) if case % 3 else (int(rng.integers(2, 200)), int(rng.integers(2, 140)))
        i0n, i1n, gt = synthetic_pair(max(w, 4), max(h, 4), seed=int(rng.integers(1 << 30)), amp=float(rng.uniform(0, 8)))
        i0n, i1n, gt = np.ascontiguousarray(i0n[:h, :w]), np.ascontiguousarray(i1n[:h, :w]), np.ascontiguousarray(gt[:, :h, :w])
        u0 = (gt * float(rng.choice([0.0, 0.5, 1.0])) + rng.normal(0, float(rng.choice([0, 0.5, 20.0])), (2, h, w))).astype(np.float32)
        xi = (rng.normal(0, 0.5, (4, h, w))).astype(np.float32) if rng.random() < 0.5 else None
        prm = dict(warps=int(rng.integers(1, 5)), max_inner=int(rng.integers(1, 24)), tol_stop=0,
                   lambda_=float(rng.choice([40.0, 10.0, 120.0])), theta=float(rng.choice([0.3, 0.1, 0.8])),
                   tau=float(rng.choice([0.125, 0.25, 0.05])), chunk=int(rng.choice([0, 0, 0, 2, 3, 4])))
        if case % 2:
            prm.update(tol_stop=1, tol=float(rng.choice([0.3, 0.1, 0.03, 0.01])), max_inner=int(rng.integers(1, 70)))
        res = []
        for persist in ("0", "1"):
            monkeypatch.setenv("FALDOI_PERSIST", persist)
            b = F.Tvl1Batch(ctx, 1, w, h)
            try:
                b.upload(0, i0n, i1n, u0, xi)
                its, errs, _, _ = b.solve(**prm)
                u, x = b.download(0, want_xi=True)
                res.append((u, x, its.copy(), errs.copy(), b.last_persistent()))
            finally:
                b.close()
        r, n = res
        assert not r[4]
        took += int(n[4])
        assert bits_equal(r[0], n[0]) and bits_equal(r[1], n[1]) and np.array_equal(r[2], n[2]) and bits_equal(r[3], n[3]), (case, w, h, prm)
    assert took >= 30                                  # (an explicit chunk the tiles cannot host stays on the per-launch path)


@pytest.mark.parametrize("tol_stop", [0, 1])
def test_persistent_kernel_gives_up_cleanly_and_the_per_launch_path_takes_over(ctx, monkeypatch, capfd, tol_stop):
    # every spin in the persistent kernel is bounded; with the bound set to one poll some tile gives up in the first exchange,
    # raises the abort word, all workgroups leave, and the solve is redone on the per-launch path: same result, no hang
    # (both forms of the kernel: static schedule, tolerance stop)
    i0n, i1n, gt = synthetic_pair(640, 360, seed=3)
    z = np.zeros_like(gt)

    def _static_solve(ctx, i0n, i1n, u0, xi0, **prm):
        b = F.Tvl1Batch(ctx, 1, 640, 360)
        try:
            b.upload(0, i0n, i1n, u0, xi0)
            its, errs, _, launches = b.solve(tol_stop=tol_stop, tol=0.05, **prm)
            u, xi = b.download(0, want_xi=True)
            return u, xi, its[0].copy(), errs[0].copy(), launches, b.last_persistent(), b.last_schedule()
        finally:
            b.close()
    monkeypatch.setenv("FALDOI_PERSIST", "0")
    ref = _static_solve(ctx, i0n, i1n, z, None, warps=3, max_inner=10)
    monkeypatch.setenv("FALDOI_PERSIST", "1")
    monkeypatch.setenv("FALDOI_PERSIST_SPINS", "1")
    new = _static_solve(ctx, i0n, i1n, z, None, warps=3, max_inner=10)
    assert bits_equal(ref[0], new[0]) and bits_equal(ref[1], new[1]) and bits_equal(ref[3], new[3])
    if not new[5]:                                   # (all neighbours ready at the first poll is possible, just unlikely)
        assert new[4] == ref[4]
        if tol_stop == 0:                            # the note is printed once per process
            assert "per-launch path" in capfd.readouterr().err


def test_tvl1_nonzero_duals_and_parameters(ctx, oracle):
    i0n, i1n, gt = synthetic_pair(96, 80, seed=11)
    rng = np.random.default_rng(1)
    xi0 = (rng.standard_normal((4, 80, 96)) * 0.8).astype(np.float32)
    _compare_solve(ctx, oracle, i0n, i1n, gt, xi0=xi0, warps=2, max_inner=35, lam=15.0, theta=0.25,
                   tau=0.2, tol=0.003)


def test_tvl1_flat_images_hit_the_zero_gradient_branch(ctx, oracle):
    # grad == 0 everywhere: TH takes the `grad < GRAD_IS_ZERO` branch (src/global_faldoi.cpp:705)
    h, w = 40, 70
    flat = np.full((h, w), 0.5, np.float32)
    rng = np.random.default_rng(2)
    u0 = (rng.standard_normal((2, h, w)) * 0.3).astype(np.float32)
    _compare_solve(ctx, oracle, flat, flat, u0, warps=2, max_inner=25)


def test_tvl1_large_flow_leaves_the_image(ctx, oracle):
    # flow pointing outside: bicubic returns 0 there (border_out = true)
    i0n, i1n, gt = synthetic_pair(128, 64, seed=9)
    u0 = np.zeros_like(gt)
    u0[0, :, :40] = -55.0
    u0[1, 50:, :] = 30.0
    _compare_solve(ctx, oracle, i0n, i1n, u0, warps=3, max_inner=30)


def test_sintel_crop_known_answer(ctx, oracle):
    case = known_answers()["sintel_clean_easy_f2_f3_crop_400_150_128x96_zero_init"]
    i0, i1 = sintel_pair((case["x0"], case["y0"], case["w"], case["h"]))
    i0n, i1n = oracle.preprocess_global(i0, i1)
    u = _compare_solve(ctx, oracle, i0n, i1n, np.zeros((2, case["h"], case["w"]), np.float32))
    assert flo_sha256(u).startswith(case["flo_sha256_prefix"])


def test_sintel_full_frame_known_answer(ctx, oracle):
    # BASELINE config 1 on the GPU: reference defaults, 1024x436, zero init.  The expected values
    # are the compiled reference's own, not the oracle's.
    case = known_answers()["sintel_clean_easy_f2_f3_zero_init"]
    i0, i1 = sintel_pair()
    i0n, i1n = oracle.preprocess_global(i0, i1)
    u, iters, errs, _, ms = ctx.tvl1_global(i0n, i1n, np.zeros((2, 436, 1024), np.float32))
    assert list(iters) == case["iters"]
    assert flo_sha256(u).startswith(case["flo_sha256_prefix"])
    assert np.allclose(errs, case["errs"], atol=1e-6)


def test_batch_pairs_are_independent_and_stop_separately(ctx, oracle):
    w, h = 160, 120
    pairs = [synthetic_pair(w, h, seed=s, amp=a) for s, a in ((1, 1.0), (2, 4.0), (3, 0.2))]
    b = F.Tvl1Batch(ctx, len(pairs), w, h)
    for i, (i0n, i1n, gt) in enumerate(pairs):
        b.upload(i, i0n, i1n, np.zeros_like(gt))
    iters, errs, ms, nl = b.solve(warps=3, max_inner=80)
    for i, (i0n, i1n, gt) in enumerate(pairs):
        ou, oit, oerr, _ = oracle.tvl1_global(i0n, i1n, np.zeros_like(gt), warps=3, max_inner=80)
        assert list(iters[i]) == list(oit)
        assert bits_equal(b.download(i), ou)
    # solving again restarts from the uploaded state and reproduces the result (idempotent)
    iters2, _, _, _ = b.solve(warps=3, max_inner=80)
    assert np.array_equal(iters, iters2)
    assert bits_equal(b.download(1), oracle.tvl1_global(pairs[1][0], pairs[1][1], np.zeros_like(gt), warps=3, max_inner=80)[0])
    b.close()


def test_full_size_fixed_schedule_properties(ctx, oracle):
    # BASELINE config 2 at full size: 1024x436, 10 inner iterations per warp, tolerance stop off.
    # The oracle finishes 20 warps in seconds; beyond that we rely on size-independent properties:
    # determinism (two runs bit-identical) and chunk-size independence.
    i0, i1 = sintel_pair()
    i0n, i1n = oracle.preprocess_global(i0, i1)
    z = np.zeros((2, 436, 1024), np.float32)
    _compare_solve(ctx, oracle, i0n, i1n, z, warps=20, max_inner=10, tol_stop=0)
    a = ctx.tvl1_global(i0n, i1n, z, warps=60, max_inner=10, tol_stop=0)[0]
    b = ctx.tvl1_global(i0n, i1n, z, warps=60, max_inner=10, tol_stop=0, chunk=5)[0]
    assert bits_equal(a, b)


def test_argument_errors_are_reported_not_fatal(ctx):
    with pytest.raises(F.FaldoiError) as e:
        F.Tvl1Batch(ctx, 1, 1, 5)
    assert e.value.code == F.capi.ERR_INVALID
    b = F.Tvl1Batch(ctx, 2, 32, 32)
    with pytest.raises(F.FaldoiError) as e:
        b.solve()
    assert e.value.code == F.capi.ERR_STATE
    z = np.zeros((32, 32), np.float32)
    b.upload(0, z, z, np.zeros((2, 32, 32), np.float32))
    b.upload(1, z, z, np.zeros((2, 32, 32), np.float32))
    with pytest.raises(F.FaldoiError):
        b.solve(chunk=99)
    with pytest.raises(F.FaldoiError):
        b.solve(tau=-1.0)
    b.close()


def test_global_faldoi_cli_reproduces_the_reference_flo(ctx, tmp_path):
    # The drop-in itself: `global_faldoi ims.txt zero.flo out.flo -w 5 -verbose 1` on the reference's
    # example frames must write the .flo the compiled reference wrote (sha256 recorded in SURVEY.md
    # App. D) and print the reference's per-warp iteration counts.
    import subprocess
    case = known_answers()["sintel_clean_easy_f2_f3_zero_init"]
    bins = F.build.build_host()
    ims = tmp_path / "ims.txt"
    ims.write_text(GOLDEN + "/sintel_clean_easy_frame_0002.png\n" + GOLDEN + "/sintel_clean_easy_frame_0003.png\n")
    z, out = tmp_path / "zero.flo", tmp_path / "out.flo"
    F.flo.write_flo(str(z), np.zeros((2, 436, 1024), np.float32))
    r = subprocess.run([bins["global_faldoi"], str(ims), str(z), str(out), "-w", "5", "-verbose", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    import hashlib
    assert hashlib.sha256(out.read_bytes()).hexdigest().startswith(case["flo_sha256_prefix"])
    lines = [l for l in r.stderr.splitlines() if l.startswith("Warping:")]
    assert [int(l.split("Iter:")[1].split()[0]) for l in lines] == case["iters"]


def test_batch_driver_over_rccl_single_rank(oracle, tmp_path):
    # batch.py end to end under torch.distributed.run with the nccl (= RCCL) backend, 1 rank on the
    # one GPU this box has: PNG decode -> scatter -> device preprocessing + batched solve -> gather -> .flo
    import subprocess
    import sys
    from PIL import Image
    crops = [(400, 150, 160, 96), (100, 40, 160, 96), (700, 300, 160, 96)]
    lines = []
    for i, c in enumerate(crops):
        i0, i1 = sintel_pair(c)
        for k, im in enumerate((i0, i1)):
            Image.fromarray(im.transpose(1, 2, 0).astype(np.uint8), "RGB").save(tmp_path / ("p%d_%d.png" % (i, k)))
        lines.append("%s %s zero %s" % (tmp_path / ("p%d_0.png" % i), tmp_path / ("p%d_1.png" % i), tmp_path / ("o%d.flo" % i)))
    (tmp_path / "jobs.txt").write_text("\n".join(lines) + "\n")
    root = GOLDEN.rsplit("/tests/", 1)[0]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", root + "/faldoi-ipol_amd/batch.py",
                        str(tmp_path / "jobs.txt"), "-w", "2", "-max_inner", "40"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for i, c in enumerate(crops):
        i0, i1 = sintel_pair(c)
        i0n, i1n = oracle.preprocess_global(i0, i1)
        ou, _, _, _ = oracle.tvl1_global(i0n, i1n, np.zeros((2, 96, 160), np.float32), warps=2, max_inner=40)
        got = F.flo.read_flo(str(tmp_path / ("o%d.flo" % i)))
        assert bits_equal(got, ou), i


# ------------------------------------------------------------------ NLTV-CSAD (BASELINE configs[4])
def test_nltvcsad_crop_known_answer(ctx, oracle):
    case = known_answers()["sintel_clean_easy_f2_f3_crop_400_150_128x96_nltvcsad_w1"]
    i0, i1 = sintel_pair((case["x0"], case["y0"], case["w"], case["h"]))
    i0n, i1n = ctx.preprocess_global(i0, i1)
    lab = F.capi.image_to_lab(i0)
    u, ms = ctx.nltvcsad_global(i0n, i1n, lab, np.zeros((2, case["h"], case["w"]), np.float32),
                                warps=1, max_inner=case["inner"])
    ou = oracle.nltvcsad_global(oracle.preprocess_global(i0, i1)[0], oracle.preprocess_global(i0, i1)[1],
                                oracle.image_to_lab(i0), np.zeros((2, case["h"], case["w"]), np.float32))
    epe = float(F.flo.epe(u, ou).mean())
    assert epe <= EPE_GATE, epe
    assert bits_equal(u, ou), mismatch_report(u, ou)
    assert flo_sha256(u).startswith(case["flo_sha256_prefix"])      # the compiled reference's own output


@pytest.mark.parametrize("w,h,warps,inner", [(40, 30, 2, 25), (97, 64, 1, 40), (7, 9, 2, 10)])
def test_nltvcsad_bit_exact_small(ctx, oracle, w, h, warps, inner):
    rng = np.random.default_rng(w * h)
    i0n, i1n, gt = synthetic_pair(max(w, 16), max(h, 16), seed=w)
    i0n, i1n, gt = i0n[:h, :w].copy(), i1n[:h, :w].copy(), gt[:, :h, :w].copy()
    rgb = np.floor(rng.random((3, h, w)) * 256).astype(np.float32)
    rgb[:, : h // 2] = np.floor(i0n[: h // 2] * 255)[None]          # half structured, half noise
    lab = F.capi.image_to_lab(rgb)
    assert bits_equal(lab, oracle.image_to_lab(rgb))
    u0 = (gt + rng.standard_normal(gt.shape) * 0.3).astype(np.float32)
    u, ms = ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=warps, max_inner=inner)
    ou = oracle.nltvcsad_global(i0n, i1n, lab, u0, warps=warps, max_inner=inner)
    assert float(F.flo.epe(u, ou).mean()) <= EPE_GATE
    assert bits_equal(u, ou), mismatch_report(u, ou)


def test_global_faldoi_cli_method6_reproduces_the_reference_flo(tmp_path):
    # `global_faldoi ims.txt zero.flo out.flo -m 6 -w 1` on the 128x96 crop: byte-identical to the
    # compiled reference's output
    import hashlib
    import subprocess
    from PIL import Image
    case = known_answers()["sintel_clean_easy_f2_f3_crop_400_150_128x96_nltvcsad_w1"]
    i0, i1 = sintel_pair((case["x0"], case["y0"], case["w"], case["h"]))
    for k, im in enumerate((i0, i1)):
        Image.fromarray(im.transpose(1, 2, 0).astype(np.uint8), "RGB").save(tmp_path / ("c%d.png" % k))
    (tmp_path / "ims.txt").write_text("%s\n%s\n" % (tmp_path / "c0.png", tmp_path / "c1.png"))
    F.flo.write_flo(str(tmp_path / "zero.flo"), np.zeros((2, case["h"], case["w"]), np.float32))
    bins = F.build.build_host()
    r = subprocess.run([bins["global_faldoi"], str(tmp_path / "ims.txt"), str(tmp_path / "zero.flo"),
                        str(tmp_path / "out.flo"), "-m", "6", "-w", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert hashlib.sha256((tmp_path / "out.flo").read_bytes()).hexdigest().startswith(case["flo_sha256_prefix"])


@pytest.mark.parametrize("w,h,warps,inner", [(64, 48, 2, 30), (33, 21, 1, 50)])
def test_nltvl1_bit_exact(ctx, oracle, w, h, warps, inner):
    # NLTV-L1 (method 2): non-local regulariser + TV-L1 data term, src/global_faldoi.cpp:1177-1327
    rng = np.random.default_rng(w + h)
    i0n, i1n, gt = synthetic_pair(w, h, seed=h)
    rgb = np.floor(np.stack([i0n, np.roll(i0n, 3, 1), i0n[::-1]]) * 255).astype(np.float32)
    lab = F.capi.image_to_lab(rgb)
    u0 = (gt + rng.standard_normal(gt.shape) * 0.3).astype(np.float32)
    u, ms = ctx.nltvl1_global(i0n, i1n, lab, u0, warps=warps, max_inner=inner)
    ou = oracle.nltvl1_global(i0n, i1n, lab, u0, warps=warps, max_inner=inner)
    assert float(F.flo.epe(u, ou).mean()) <= EPE_GATE
    assert bits_equal(u, ou), mismatch_report(u, ou)


# ------------------------------------------------------------------ shapes the tiling must survive
@pytest.mark.parametrize("w,h", [(2, 2), (3, 200), (300, 2), (65, 65), (43, 45), (44, 44), (129, 7), (1, 1)])
def test_tvl1_awkward_shapes(ctx, oracle, w, h):
    if w < 2 or h < 2:
        with pytest.raises(F.FaldoiError):
            ctx.tvl1_global(np.zeros((h, w), np.float32), np.zeros((h, w), np.float32), np.zeros((2, h, w), np.float32))
        return
    rng = np.random.default_rng(w * 1000 + h)
    i0n = rng.random((h, w)).astype(np.float32)
    i1n = np.clip(i0n + rng.standard_normal((h, w)).astype(np.float32) * 0.05, 0, 1).astype(np.float32)
    u0 = (rng.standard_normal((2, h, w)) * 0.7).astype(np.float32)
    for chunk in (0, 3):
        _compare_solve(ctx, oracle, i0n, i1n, u0, warps=2, max_inner=17, chunk=chunk)


def test_tvl1_large_image_many_tiles(ctx, oracle):
    # 2048x1100: 47 x 25 tiles, several workgroups per CU; stop test on
    i0n, i1n, gt = synthetic_pair(2048, 1100, seed=77, amp=5.0)
    _compare_solve(ctx, oracle, i0n, i1n, np.zeros_like(gt), warps=2, max_inner=25)


def test_batch_of_many_small_pairs(ctx, oracle):
    n, w, h = 37, 70, 50
    b = F.Tvl1Batch(ctx, n, w, h)
    pairs = [synthetic_pair(w, h, seed=100 + i, amp=0.5 + 0.1 * i) for i in range(n)]
    for i, (a, c, gt) in enumerate(pairs):
        b.upload(i, a, c, np.zeros_like(gt))
    iters, errs, ms, nl = b.solve(warps=2, max_inner=60)
    for i in (0, 5, 17, 36):
        ou, oit, oerr, _ = oracle.tvl1_global(pairs[i][0], pairs[i][1], np.zeros((2, h, w), np.float32), warps=2,
                                             max_inner=60)
        assert list(iters[i]) == list(oit)
        assert bits_equal(b.download(i), ou) and bits_equal(errs[i], oerr)
    b.close()


def test_zoom_resampler_bit_exact(ctx, oracle):
    # src/zoom.c on the GPU against the oracle (itself pinned to the reference's zoom.c objects)
    g = np.load(GOLDEN + "/ref_primitives.npz")
    h, w = g["img"].shape
    for tag, f in (("050", 0.5), ("075", 0.75)):
        zo = ctx.zoom_out(g["img"], f)
        assert bits_equal(zo, g["zoom_out_" + tag])
        assert bits_equal(ctx.zoom_in(zo, w, h), g["zoom_in_" + tag])
    i0, _ = sintel_pair()
    gray = oracle.preprocess_global(i0, i0)[0]
    for f in (0.5, 0.35):
        a = ctx.zoom_out(gray, f)
        b = oracle.zoom_out(gray, f)
        assert bits_equal(a, b), f
        assert bits_equal(ctx.zoom_in(a, 1024, 436), oracle.zoom_in(b, 1024, 436))
    with pytest.raises(F.FaldoiError):
        ctx.zoom_out(gray, 1.5)


def test_second_sintel_pair_large_motion(ctx, oracle):
    # example_data/final/hard frames 2->3 (motion blur, displacements up to ~39 px, never meets the
    # tolerance: 5 x 400 iterations).  No reference output was recorded for this pair, so the check is
    # GPU == oracle (bit for bit), through the device-side preprocessing as well.
    from util import load_rgb_planar
    i0 = load_rgb_planar(GOLDEN + "/sintel_final_hard_frame_0002.png")
    i1 = load_rgb_planar(GOLDEN + "/sintel_final_hard_frame_0003.png")
    o0, o1 = oracle.preprocess_global(i0, i1)
    g0, g1 = ctx.preprocess_global(i0, i1)
    assert bits_equal(g0, o0) and bits_equal(g1, o1)
    u = _compare_solve(ctx, oracle, g0, g1, np.zeros((2, 436, 1024), np.float32))
    assert float(np.abs(u).max()) > 30.0


# ------------------------------------------------------------------ local patch operator
def test_local_patch_operator_sequence_bit_exact(ctx, oracle):
    # one patch per call, overlapping windows, 3x3 and 11x11, image corners: the sequential use the
    # reference makes of of_estimation.  Oracle (pinned to the reference objects) vs GPU: energies,
    # iteration counts and the whole flow plane after every call.
    i0, i1 = sintel_pair((400, 150, 96, 64))
    i0n, i1n = oracle.preprocess_global(i0, i1)
    h, w = i0n.shape
    rng = np.random.default_rng(0)
    u_o = (rng.standard_normal((2, h, w)) * 0.5).astype(np.float32)
    op = oracle.LocalTvl1(i0n, i1n)
    g = F.LocalTvl1(ctx, i0n, i1n)
    g.set_flow(u_o)
    centers = [(5, 5), (7, 6), (20, 30), (21, 30), (0, 0), (95, 63), (50, 1), (48, 3), (94, 30), (2, 40), (30, 30),
               (30, 31), (60, 20)] * 2
    for n, (ci, cj) in enumerate(centers):
        pi = oracle.patch_index(1 if n % 5 == 0 else 5, w, h, ci, cj)
        e_o, it_o = op.estimate(u_o, pi, 4)
        e_g, it_g = g.estimate([pi], 4)
        assert int(it_g[0]) == it_o, n
        assert np.float32(e_o).view(np.uint32) == e_g.view(np.uint32)[0], (n, e_o, e_g)
        assert bits_equal(g.get_flow(), u_o), n
    g.close()


def test_tvl1_uhd_frame_bit_exact(ctx, oracle):
    # a 3840x2160 frame (8.3 Mpx, 4 000+ tiles: the many-tiles schedule, 64-bit plane offsets): same bits as the CPU path
    w, h = 3840, 2160
    i0n, i1n, gt = synthetic_pair(w, h, seed=77, amp=3.0)
    u0 = (gt * np.float32(0.5)).astype(np.float32)
    want, wit, werr, wxi = oracle.tvl1_global(i0n, i1n, u0, warps=2, max_inner=12, tol_stop=True, tol=0.05)
    got, git, gerr, gxi, _ = ctx.tvl1_global(i0n, i1n, u0, xi=np.zeros((4, h, w), np.float32), warps=2, max_inner=12, tol_stop=1, tol=0.05)
    assert np.array_equal(git, wit) and bits_equal(gerr, werr)
    assert bits_equal(got, want), mismatch_report(got, want)
    assert bits_equal(gxi, wxi)


def _same_up_to_nan_payload(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32))


def test_nan_flow_spreads_like_in_the_reference(ctx, oracle):
    # A NaN in the flow (windows over pruned pixels that no fill touched; a NaN in an initial flow) makes the norm of
    # that pixel's duals NaN one iteration later, and the reference's MAX(1, xi_N) macro then yields NaN for all
    # four duals of the pixel, not only the NaN ones: the set of NaN pixels after k iterations must be the same
    # (found by tests/tools/fuzz_parity.py: one window energy finite instead of NaN, two more candidates in a queue).
    i0n, i1n, gt = synthetic_pair(96, 64, seed=9)
    rng = np.random.default_rng(4)
    u0 = (gt + rng.standard_normal(gt.shape) * 0.3).astype(np.float32)
    u0[0, 30, 40] = np.nan
    u0[1, 12, 70] = np.nan
    u_o = u0.copy()
    op = oracle.LocalTvl1(i0n, i1n)
    g = F.LocalTvl1(ctx, i0n, i1n)
    g.set_flow(u0)
    for pi in (oracle.patch_index(5, 96, 64, 41, 32), oracle.patch_index(2, 96, 64, 69, 13), oracle.patch_index(5, 96, 64, 20, 50)):
        e_o, it_o = op.estimate(u_o, pi, 6)
        e_g, it_g = g.estimate([pi], 6)
        assert int(it_g[0]) == it_o
        assert _same_up_to_nan_payload(e_g[0], e_o), (e_g, e_o)
        assert _same_up_to_nan_payload(g.get_flow(), u_o), pi
    assert np.isnan(u_o).sum() > 40 and np.isfinite(u_o).sum() > 1000
    g.close()
    # the global solver shares the step
    want, wit, _, _ = oracle.tvl1_global(i0n, i1n, u0, warps=2, max_inner=7, tol_stop=False)
    got, git, _, _, _ = ctx.tvl1_global(i0n, i1n, u0, warps=2, max_inner=7, tol_stop=0)
    assert _same_up_to_nan_payload(got, want) and np.isnan(want).sum() > 100


def test_local_patch_operator_batch_of_disjoint_patches(ctx, oracle):
    # many non-interacting windows in ONE launch == the same windows one after the other
    i0n, i1n, gt = synthetic_pair(256, 160, seed=21)
    rng = np.random.default_rng(5)
    u0 = (gt + rng.standard_normal(gt.shape) * 0.4).astype(np.float32)
    u_o = u0.copy()
    op = oracle.LocalTvl1(i0n, i1n)
    patches = [oracle.patch_index(5, 256, 160, ci, cj) for cj in range(20, 150, 24) for ci in range(20, 240, 24)]
    e_o = [op.estimate(u_o, pi, 4) for pi in patches]
    g = F.LocalTvl1(ctx, i0n, i1n)
    g.set_flow(u0)
    e_g, it_g = g.estimate(patches, 4)
    assert [int(x) for x in it_g] == [it for _, it in e_o]
    assert np.array_equal(e_g.view(np.uint32), np.array([e for e, _ in e_o], np.float32).view(np.uint32))
    assert bits_equal(g.get_flow(), u_o)
    with pytest.raises(F.FaldoiError):
        g.estimate([(0, 0, 12, 5)], 4)            # wider than 11
    with pytest.raises(F.FaldoiError):
        g.estimate([(250, 0, 257, 5)], 4)         # outside the image
    g.close()


def test_interpolate_poisson_bit_exact(ctx, oracle):
    # interpolate_poisson on windows with random fixed / free pixels (incl. none fixed, all fixed, corners)
    h, w = 48, 64
    rng = np.random.default_rng(8)
    i0n, i1n, _ = synthetic_pair(w, h, seed=2)
    u = (rng.standard_normal((2, h, w)) * 2).astype(np.float32)
    fixed = (rng.random((h, w)) < 0.35).astype(np.int32)
    fixed[20:31, 40:51] = 0            # a window with no fixed pixel at all
    fixed[0:11, 0:11] = 1              # and one with every pixel fixed
    patches = [oracle.patch_index(5, w, h, ci, cj) for ci, cj in ((5, 5), (45, 25), (63, 47), (0, 47), (30, 10), (12, 40))]
    patches.append(oracle.patch_index(1, w, h, 33, 33))
    g = F.LocalTvl1(ctx, i0n, i1n)
    u_o = u.copy()
    g.set_flow(u)
    g.set_fixed(fixed)
    L = oracle.lib()
    for pi in patches:                  # overlapping windows: one call each, like the reference
        L.orc_interpolate_poisson(u_o[0], u_o[1], fixed, w, pi[0], pi[1], pi[2], pi[3])
        g.interpolate([pi])
        assert bits_equal(g.get_flow(), u_o), pi
    g.close()


def _seed_flows(oracle, a, b, frac, seed):
    """Sparse integer seeds sampled from the oracle's dense TV-L1 flows (forward and backward)."""
    h, w = a.shape
    L = oracle.lib()
    rng = np.random.default_rng(seed)
    res = []
    for src, dst in ((a, b), (b, a)):
        u = np.zeros((2, h, w), np.float32)
        xi = np.zeros((4, h, w), np.float32)
        its, errs = np.zeros(2, np.int32), np.zeros(2, np.float32)
        L.orc_tvl1_global(src, dst.copy(), u[0], u[1], xi[0], xi[1], xi[2], xi[3], 40, 0.3, 0.125, 0.01, w, h, 2, 60, 1, its, errs)
        s = np.full_like(u, np.nan)
        m = rng.random((h, w)) < frac
        s[:, m] = np.round(u[:, m])
        res.append(s)
    return res


@pytest.mark.parametrize("crop,frac", [((400, 150, 96, 64), 0.01), ((0, 0, 70, 50), 0.03), ((950, 380, 74, 56), 0.004)])
def test_match_growing_bit_exact(ctx, oracle, crop, frac):
    # the whole local step: seeds, 3 x (growing fwd/bwd, pruning, re-seeding), last growing
    i0, i1 = sintel_pair(crop)
    a, b = oracle.preprocess_local(i0, i1)
    ga, gb = ctx.preprocess_local(i0, i1)
    assert bits_equal(ga, a) and bits_equal(gb, b)
    go, ba = _seed_flows(oracle, a, b, frac, seed=crop[0])
    want, want_ene, st = oracle.local_faldoi(a, b, go, ba)
    got, got_ene, gst = ctx.match_growing(i0, i1, go, ba, budget=3000)
    assert list(gst.pops) == [st[0], st[3]] and list(gst.solves) == [st[1], st[4]] and list(gst.fills) == [st[2], st[5]]
    assert bits_equal(got, want), mismatch_report(got, want)
    assert bits_equal(got_ene, want_ene)
    assert np.isfinite(got).all()
    if crop[0] == 400:                      # what very large batches use (the register-capped build of the growing
        import os                           # kernel, re-seeding by a device kernel instead of the host): same bits
        os.environ["FALDOI_GROW_DENSE"] = "1"
        os.environ["FALDOI_RESEED"] = "device"
        try:
            got2, ene2, _ = ctx.match_growing(i0, i1, go, ba)
        finally:
            os.environ.pop("FALDOI_GROW_DENSE")
            os.environ.pop("FALDOI_RESEED")
        assert bits_equal(got2, want) and bits_equal(ene2, want_ene)


def test_three_cli_pipeline_matches_cpu_path(ctx, oracle, tmp_path):
    # sparse_flow -> local_faldoi -> global_faldoi (the stage order of scripts_python/faldoi_sift.py:274-318) on a crop,
    # from match lists; every stage byte-identical to the CPU checker's
    from PIL import Image
    bins = F.build.build_host()
    crop = (380, 140, 112, 80)
    i0, i1 = sintel_pair(crop)
    h, w = i0.shape[1:]
    for k, im in enumerate((i0, i1)):
        Image.fromarray(im.transpose(1, 2, 0).astype(np.uint8), "RGB").save(tmp_path / ("f%d.png" % k))
    (tmp_path / "ims.txt").write_text("%s\n%s\n" % (tmp_path / "f0.png", tmp_path / "f1.png"))
    a, b = oracle.preprocess_local(i0, i1)
    go, ba = _seed_flows(oracle, a, b, 0.012, seed=11)
    for name, s in (("m0.txt", go), ("m1.txt", ba)):                 # match lists "x0 y0 x1 y1"
        ys, xs = np.nonzero(np.isfinite(s[0]))
        with open(tmp_path / name, "w") as f:
            for y, x in zip(ys, xs):
                f.write("%d %d %d %d\n" % (x, y, x + int(s[0, y, x]), y + int(s[1, y, x])))
    run = lambda *cmd: subprocess.run([str(c) for c in cmd], capture_output=True, text=True)
    for m, o in (("m0.txt", "s0.flo"), ("m1.txt", "s1.flo")):
        r = run(bins["sparse_flow"], tmp_path / m, w, h, tmp_path / o)
        assert r.returncode == 0, r.stderr
    go, ba = F.flo.read_flo(str(tmp_path / "s0.flo")), F.flo.read_flo(str(tmp_path / "s1.flo"))
    assert np.isfinite(go[0]).sum() > 50 and np.isfinite(ba[0]).sum() > 50
    r = run(bins["local_faldoi"], tmp_path / "ims.txt", tmp_path / "s0.flo", tmp_path / "s1.flo", tmp_path / "local.flo",
            tmp_path / "sim.tiff", "-m", 0, "-wr", 5, "-loc_it", 3, "-max_pch_it", 4, "-split_img", 0, "-h_parts", 3,
            "-v_parts", 2, "-fb_thresh", 2, "-partial_res", 0, "-verbose", 0)
    assert r.returncode == 0, r.stderr + r.stdout
    want_local, want_ene, _ = oracle.local_faldoi(a, b, go, ba)
    got_local = F.flo.read_flo(str(tmp_path / "local.flo"))
    assert bits_equal(got_local, want_local), mismatch_report(got_local, want_local)
    sim = open(tmp_path / "sim.tiff").read().split()
    assert sim[:4] == ["P2", str(w), str(h), "65535"]
    assert np.array_equal(np.array([float.fromhex(v) for v in sim[4:]], np.float32), want_ene.ravel())
    r = run(bins["global_faldoi"], tmp_path / "ims.txt", tmp_path / "local.flo", tmp_path / "final.flo", "-m", 0, "-w", 5)
    assert r.returncode == 0, r.stderr
    # CPU path of the global step from the same local flow (2-line list: I-1 := I1 in the normalisation)
    g0, g1 = oracle.preprocess_global(i0, i1, None)
    u = want_local.copy()
    xi = np.zeros((4, h, w), np.float32)
    its, errs = np.zeros(5, np.int32), np.zeros(5, np.float32)
    oracle.lib().orc_tvl1_global(g0, g1.copy(), u[0], u[1], xi[0], xi[1], xi[2], xi[3], 40, 0.3, 0.125, 0.01, w, h, 5, 400, 1, its, errs)
    got = F.flo.read_flo(str(tmp_path / "final.flo"))
    assert bits_equal(got, u), mismatch_report(got, u)
    # partitioned growing through the executable
    r = run(bins["local_faldoi"], tmp_path / "ims.txt", tmp_path / "s0.flo", tmp_path / "s1.flo", tmp_path / "local_split.flo",
            tmp_path / "sim_split.tiff", "-split_img", 1, "-h_parts", 2, "-v_parts", 2)
    assert r.returncode == 0, r.stderr + r.stdout
    want_split, _, _ = oracle.local_faldoi(a, b, go, ba, split_img=1, h_parts=2, v_parts=2)
    assert bits_equal(F.flo.read_flo(str(tmp_path / "local_split.flo")), want_split)
    # the one-process driver (no intermediate files) gives the same bytes; "sift" layout = match_cli columns 2,1,6,5
    for name in ("m0.txt", "m1.txt"):
        with open(tmp_path / name) as f, open(tmp_path / ("raw_" + name), "w") as g:
            for line in f:
                x0, y0, x1, y1 = line.split()
                g.write("%s %s 1.5 0.3 %s %s 1.6 0.2 77\n" % (y0, x0, y1, x1))
    r = run(bins["faldoi_flow"], tmp_path / "ims.txt", tmp_path / "raw_m0.txt", tmp_path / "raw_m1.txt", tmp_path / "flow.flo",
            "-local_out", tmp_path / "local2.flo")
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "flow.flo").read_bytes() == (tmp_path / "final.flo").read_bytes()
    assert (tmp_path / "local2.flo").read_bytes() == (tmp_path / "local.flo").read_bytes()


def test_match_growing_options_saliency_and_empty_direction(ctx, oracle):
    # non-default radius / iteration counts, saliency maps, and a backward direction without any seed
    i0, i1 = sintel_pair((600, 200, 64, 48))
    a, b = oracle.preprocess_local(i0, i1)
    go, ba = _seed_flows(oracle, a, b, 0.02, seed=4)
    rng = np.random.default_rng(9)
    sal0 = rng.uniform(0.5, 1.5, a.shape).astype(np.float32)
    sal1 = rng.uniform(0.5, 1.5, a.shape).astype(np.float32)
    want, wene, st = oracle.local_faldoi(a, b, go, ba, wr=3, loc_it=2, max_iter_patch=2, fb_thresh=1.0, sal0=sal0, sal1=sal1)
    got, gene, gst = ctx.match_growing(a, b, go, ba, sal0, sal1, preprocessed=1, wr=3, loc_it=2, max_iter_patch=2, fb_thresh=1.0,
                                       budget=777)
    assert bits_equal(got, want), mismatch_report(got, want)
    assert bits_equal(gene, wene) and list(gst.solves) == [st[1], st[4]]
    none = np.full_like(ba, np.nan)
    want, wene, st = oracle.local_faldoi(a, b, go, none, loc_it=1)
    got, gene, gst = ctx.match_growing(a, b, go, none, preprocessed=1, loc_it=1)
    assert st[4] == 0 and list(gst.solves) == [st[1], 0]
    assert bits_equal(got, want), mismatch_report(got, want)
    want, wene, st = oracle.local_faldoi(a, b, go, ba, loc_it=0)             # a single forward growing
    got, gene, gst = ctx.match_growing(a, b, go, ba, preprocessed=1, loc_it=0)
    assert bits_equal(got, want) and bits_equal(gene, wene)


def test_match_growing_batch_of_different_pairs(ctx, oracle):
    # several pairs in one launch sequence (one workgroup per growing): each equals its own sequential CPU run
    crops = [(100, 50, 56, 40), (500, 300, 56, 40), (900, 10, 56, 40)]
    ins = []
    for k, crop in enumerate(crops):
        i0, i1 = sintel_pair(crop)
        a, b = oracle.preprocess_local(i0, i1)
        go, ba = _seed_flows(oracle, a, b, 0.02, seed=20 + k)
        ins.append((i0, i1, a, b, go, ba))
    outs, enes, st = ctx.match_growing_batch([x[0] for x in ins], [x[1] for x in ins], [x[4] for x in ins],
                                             [x[5] for x in ins], budget=2000)
    total = 0
    for (i0, i1, a, b, go, ba), out, ene in zip(ins, outs, enes):
        want, wene, ost = oracle.local_faldoi(a, b, go, ba)
        total += ost[1] + ost[4]
        assert bits_equal(out, want), mismatch_report(out, want)
        assert bits_equal(ene, wene)
    assert sum(st.solves) == total


def test_fb_consistency_check_at_the_threshold(ctx, oracle):
    # |u + v(x+u)| within a few ulps of epsilon: the C library's hypotf decides on the CPU, sqrt(x*x + y*y) in
    # double on the device; plus a generic case with sub-pixel warps, borders and NaN holes
    h, w = 40, 64
    rng = np.random.default_rng(2)
    ang = rng.uniform(0, 2 * np.pi, (h, w))
    r = np.float32(2.0) * (1 + rng.integers(-3, 4, (h, w)) * np.float32(2.0 ** -23))
    go = np.zeros((2, h, w), np.float32)                       # zero forward flow: the warp is the identity
    ba = np.stack([(r * np.cos(ang)), (r * np.sin(ang))]).astype(np.float32)
    trust = np.zeros((h, w), np.int32)
    oracle.lib().orc_fb_consistency_check(go, ba, trust, w, h, 2.0)
    assert 0.2 < trust[3:-3, 3:-3].mean() < 0.8
    assert np.array_equal(ctx.fb_consistency_check(go, ba, 2.0), trust)
    go = (rng.standard_normal((2, h, w)) * 2).astype(np.float32)
    ba = (-go + rng.standard_normal((2, h, w)) * 1.5).astype(np.float32)
    ba[:, 5:9, 7:12] = np.nan
    go[:, 20:22, 30:33] = np.nan
    oracle.lib().orc_fb_consistency_check(go, ba, trust, w, h, 2.0)
    assert np.array_equal(ctx.fb_consistency_check(go, ba, 2.0), trust)


@pytest.mark.parametrize("crop,warps,tol", [((400, 150, 96, 64), 3, 0.01), ((30, 20, 61, 45), 2, 0.02)])
def test_tvcsad_bit_exact(ctx, oracle, crop, warps, tol):
    # TV-CSAD (method 4): CSAD data term + TV regulariser, stop test = the reference's one-thread serial mean
    i0, i1 = sintel_pair(crop)
    a, b = oracle.preprocess_global(i0, i1, None)
    u0 = np.zeros((2,) + a.shape, np.float32)
    want, wits, werr, wxi = oracle.tvcsad_global(a, b, u0, warps=warps, tol=tol, max_inner=150)
    got, gits, gerr, gxi, st = ctx.tvcsad_global(a, b, u0, warps=warps, tol=tol, max_inner=150)
    assert list(gits) == list(wits), (gits, wits)
    assert bits_equal(got, want), mismatch_report(got, want)
    assert bits_equal(gxi, wxi)
    assert min(wits) < 150                 # at least one warp ends through the tolerance test
    # the fast decision (order-independent double sum outside the rounding band) and the reference's serial
    # fp32 sum must agree: force the serial path for every stop test and compare
    import os
    os.environ["FALDOI_TVCSAD_SERIAL_SUM"] = "1"
    try:
        got2, gits2, gerr2, _, st2 = ctx.tvcsad_global(a, b, u0, warps=warps, tol=tol, max_inner=150)
    finally:
        os.environ.pop("FALDOI_TVCSAD_SERIAL_SUM")
    assert st2.chunk_launches == int(gits2.sum()) and list(gits2) == list(wits)
    assert bits_equal(got2, want) and np.array_equal(gerr2.view(np.uint32), werr.view(np.uint32))


def test_global_faldoi_cli_method4_matches_cpu_path(oracle, tmp_path):
    # `global_faldoi ims.txt zero.flo out.flo -m 4 -w 2` (TV-CSAD) on a crop against the CPU path
    from PIL import Image
    i0, i1 = sintel_pair((420, 160, 80, 56))
    h, w = i0.shape[1:]
    for k, im in enumerate((i0, i1)):
        Image.fromarray(im.transpose(1, 2, 0).astype(np.uint8), "RGB").save(tmp_path / ("c%d.png" % k))
    (tmp_path / "ims.txt").write_text("%s\n%s\n" % (tmp_path / "c0.png", tmp_path / "c1.png"))
    F.flo.write_flo(str(tmp_path / "zero.flo"), np.zeros((2, h, w), np.float32))
    bins = F.build.build_host()
    r = subprocess.run([bins["global_faldoi"], str(tmp_path / "ims.txt"), str(tmp_path / "zero.flo"), str(tmp_path / "out.flo"),
                        "-m", "4", "-w", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    a, b = oracle.preprocess_global(i0, i1, None)
    want, _, _, _ = oracle.tvcsad_global(a, b, np.zeros((2, h, w), np.float32), warps=2)
    got = F.flo.read_flo(str(tmp_path / "out.flo"))
    assert bits_equal(got, want), mismatch_report(got, want)


def test_match_growing_weighted_functional(ctx, oracle):
    # -m 1 (M_TVL1_W): per-pixel Gaussian weights on the data term, lambda rescaled by the (sic) off-centre weight
    i0, i1 = sintel_pair((640, 100, 72, 52))
    a, b = oracle.preprocess_local(i0, i1)
    go, ba = _seed_flows(oracle, a, b, 0.02, seed=31)
    want, wene, st = oracle.local_faldoi(a, b, go, ba, method=1)
    got, gene, gst = ctx.match_growing(a, b, go, ba, preprocessed=1, method=1)
    assert list(gst.solves) == [st[1], st[4]] and list(gst.pops) == [st[0], st[3]]
    assert bits_equal(got, want), mismatch_report(got, want)
    assert bits_equal(gene, wene)
    plain, _, _ = oracle.local_faldoi(a, b, go, ba, method=0)
    assert not bits_equal(plain, want)                       # the weights do change the result


def test_batch_driver_pipeline_jobs_over_rccl(oracle, tmp_path):
    # batch.py with pipeline jobs (seed flows -> local step -> global step) under torch.distributed.run / RCCL
    import sys
    from PIL import Image
    crops = [(400, 150, 64, 48), (100, 40, 64, 48)]
    lines, want = [], []
    for i, c in enumerate(crops):
        i0, i1 = sintel_pair(c)
        for k, im in enumerate((i0, i1)):
            Image.fromarray(im.transpose(1, 2, 0).astype(np.uint8), "RGB").save(tmp_path / ("p%d_%d.png" % (i, k)))
        a, b = oracle.preprocess_local(i0, i1)
        go, ba = _seed_flows(oracle, a, b, 0.02, seed=50 + i)
        F.flo.write_flo(str(tmp_path / ("s%d_0.flo" % i)), go)
        F.flo.write_flo(str(tmp_path / ("s%d_1.flo" % i)), ba)
        lines.append("%s %s %s %s %s" % (tmp_path / ("p%d_0.png" % i), tmp_path / ("p%d_1.png" % i), tmp_path / ("s%d_0.flo" % i),
                                         tmp_path / ("s%d_1.flo" % i), tmp_path / ("o%d.flo" % i)))
        local, _, _ = oracle.local_faldoi(a, b, go, ba)
        g0, g1 = oracle.preprocess_global(i0, i1)
        want.append(oracle.tvl1_global(g0, g1, local, warps=2, max_inner=40)[0])
    (tmp_path / "jobs.txt").write_text("\n".join(lines) + "\n")
    root = GOLDEN.rsplit("/tests/", 1)[0]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29543", root + "/faldoi-ipol_amd/batch.py",
                        str(tmp_path / "jobs.txt"), "-w", "2", "-max_inner", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for i in range(len(crops)):
        got = F.flo.read_flo(str(tmp_path / ("o%d.flo" % i)))
        assert bits_equal(got, want[i]), i


def test_match_growing_edge_cases(ctx, oracle):
    # tiny frame, dense seeds (every pixel a seed: the queue's worst case), small radii, one seed only
    i0n, i1n, gt = synthetic_pair(40, 28, seed=5, amp=1.5)
    rng = np.random.default_rng(12)
    dense = np.round(gt + rng.standard_normal(gt.shape) * 0.4).astype(np.float32)
    want, wene, st = oracle.local_faldoi(i0n, i1n, dense, -dense, loc_it=2)
    got, gene, gst = ctx.match_growing(i0n, i1n, dense, -dense, preprocessed=1, loc_it=2, budget=500)
    assert bits_equal(got, want) and bits_equal(gene, wene) and list(gst.pops) == [st[0], st[3]]
    for wr in (1, 2):
        sparse = np.full_like(dense, np.nan)
        sparse[:, ::7, ::9] = dense[:, ::7, ::9]
        want, wene, st = oracle.local_faldoi(i0n, i1n, sparse, -sparse, wr=wr, loc_it=1)
        got, gene, gst = ctx.match_growing(i0n, i1n, sparse, -sparse, preprocessed=1, wr=wr, loc_it=1)
        assert bits_equal(got, want), wr
    one = np.full_like(dense, np.nan)
    one[:, 14, 20] = (1.0, -1.0)
    want, wene, st = oracle.local_faldoi(i0n, i1n, one, one)
    got, gene, gst = ctx.match_growing(i0n, i1n, one, one, preprocessed=1)
    assert bits_equal(got, want) and bits_equal(gene, wene)
    t0, t1, tg = synthetic_pair(9, 7, seed=6, amp=0.5)                      # windows clipped on every side
    s = np.full_like(tg, np.nan)
    s[:, 3, 4] = 0.0
    s[:, 0, 0] = (1.0, 0.0)
    want, wene, st = oracle.local_faldoi(t0, t1, s, s)
    got, gene, gst = ctx.match_growing(t0, t1, s, s, preprocessed=1)
    assert bits_equal(got, want) and bits_equal(gene, wene)
    with pytest.raises(F.FaldoiError):
        ctx.match_growing(t0, t1, s, s, preprocessed=1, wr=6)               # window larger than the kernel's 11x11


def test_device_queue_pops_like_the_host_queue(ctx):
    # the wavefront-wide heap walks against the shared host definition (itself tested against std::priority_queue)
    rng = np.random.default_rng(6)
    for levels, n, ppush in ((3, 300, 0.6), (16, 5000, 0.55), (100000, 30000, 0.7), (2, 65000, 0.62), (50, 4000, 0.5)):
        ops = (rng.random(n) < ppush).astype(np.int32)
        ops[:8] = 1
        sims = rng.integers(0, levels, n).astype(np.float32) * 0.25
        want = F.capi.queue_replay(ops, sims)
        got = ctx.queue_replay(ops, sims)
        assert np.array_equal(got, want), (levels, n, int(np.argmax(got != want)) if len(got) == len(want) else (len(got), len(want)))


@pytest.mark.parametrize("crop,hp,vp,loc_it,method", [((400, 150, 100, 70), 3, 2, 3, 0), ((100, 60, 91, 66), 2, 2, 2, 0),
                                                       ((700, 250, 72, 54), 1, 3, 3, 0), ((250, 300, 80, 60), 2, 3, 2, 1)])
def test_match_growing_with_image_partitions(ctx, oracle, crop, hp, vp, loc_it, method):
    # -split_img 1: iteration 0 on the whole image, then growings on an hp x vp grid of partitions (odd iterations),
    # on its transpose (even ones) and on the grid again for the last growing; remainders and the reference's index
    # adjustment quirk (odd width with two columns) included
    i0, i1 = sintel_pair(crop)
    a, b = oracle.preprocess_local(i0, i1)
    go, ba = _seed_flows(oracle, a, b, 0.015, seed=hp * 10 + vp)
    want, wene, st = oracle.local_faldoi(a, b, go, ba, loc_it=loc_it, split_img=1, h_parts=hp, v_parts=vp, method=method)
    got, gene, gst = ctx.match_growing(a, b, go, ba, preprocessed=1, loc_it=loc_it, split_img=1, h_parts=hp, v_parts=vp, method=method)
    assert list(gst.solves) == [st[1], st[4]] and list(gst.pops) == [st[0], st[3]] and list(gst.fills) == [st[2], st[5]]
    assert bits_equal(got, want), mismatch_report(got, want)
    assert bits_equal(gene, wene)
    plain, _, _ = oracle.local_faldoi(a, b, go, ba, loc_it=loc_it, method=method)
    assert not bits_equal(plain, want)                     # partitioning changes the result, in the reference too
    sal = np.ones_like(a)
    with pytest.raises(F.FaldoiError):
        ctx.match_growing(a, b, go, ba, sal, sal, preprocessed=1, split_img=1)


@pytest.mark.parametrize("crop,hp,vp,method", [((400, 150, 100, 70), 3, 2, 0), ((250, 300, 80, 60), 2, 3, 1)])
def test_match_growing_partitions_from_the_first_iteration(ctx, oracle, crop, hp, vp, method):
    # -split_img 2 (additive, not in the reference): local iteration 0 runs on the transposed grid as well, so the
    # 2 * hp * vp growings are in flight from the start; same bits as the CPU restatement of that schedule, and a
    # result different from both the reference's partition mode and the whole-image run
    i0, i1 = sintel_pair(crop)
    a, b = oracle.preprocess_local(i0, i1)
    go, ba = _seed_flows(oracle, a, b, 0.03, seed=5)
    want, wene, st = oracle.local_faldoi(a, b, go, ba, split_img=2, h_parts=hp, v_parts=vp, method=method)
    got, gene, gst = ctx.match_growing(a, b, go, ba, preprocessed=1, split_img=2, h_parts=hp, v_parts=vp, method=method)
    assert list(gst.solves) == [st[1], st[4]] and list(gst.pops) == [st[0], st[3]] and list(gst.fills) == [st[2], st[5]]
    assert bits_equal(got, want), mismatch_report(got, want)
    assert bits_equal(gene, wene)
    ref_mode, _, _ = oracle.local_faldoi(a, b, go, ba, split_img=1, h_parts=hp, v_parts=vp, method=method)
    assert not bits_equal(ref_mode, want)
    # a grid of its own for iteration 0 (coarser: keeps most of the accuracy)
    first = dict(split_img=2, h_parts=hp, v_parts=vp, first_h_parts=2, first_v_parts=1, method=method)
    want1, wene1, st1 = oracle.local_faldoi(a, b, go, ba, **first)
    got1, gene1, gst1 = ctx.match_growing(a, b, go, ba, preprocessed=1, **first)
    assert bits_equal(got1, want1) and bits_equal(gene1, wene1) and list(gst1.pops) == [st1[0], st1[3]]
    assert not bits_equal(want1, want)
    with pytest.raises(F.FaldoiError):
        ctx.match_growing(a, b, go, ba, preprocessed=1, split_img=1, first_h_parts=2, first_v_parts=1)
    with pytest.raises(F.FaldoiError):
        ctx.match_growing(a, b, go, ba, preprocessed=1, split_img=3)


def test_match_growing_batch_with_partitions(ctx, oracle):
    # several pairs in one call, each on its own partitions, all growings in one launch; pair 1 has no backward seed,
    # so it alone falls back to whole-image growings while the others stay partitioned
    crops = [(400, 150, 90, 60), (100, 60, 90, 60), (700, 250, 90, 60)]
    i0s, i1s, gos, bas, wants = [], [], [], [], []
    for q, crop in enumerate(crops):
        i0, i1 = sintel_pair(crop)
        a, b = oracle.preprocess_local(i0, i1)
        go, ba = _seed_flows(oracle, a, b, 0.02, seed=30 + q)
        if q == 1:
            ba = np.full_like(ba, np.nan)
        i0s.append(a); i1s.append(b); gos.append(go); bas.append(ba)
        wants.append(oracle.local_faldoi(a, b, go, ba, split_img=1, h_parts=3, v_parts=2))
    outs, enes, gst = ctx.match_growing_batch(i0s, i1s, gos, bas, preprocessed=1, split_img=1, h_parts=3, v_parts=2)
    for out, ene, (want, wene, st) in zip(outs, enes, wants):
        assert bits_equal(out, want), mismatch_report(out, want)
        assert bits_equal(ene, wene)
    assert list(gst.solves) == [sum(int(w[2][1]) for w in wants), sum(int(w[2][4]) for w in wants)]
    plain, _, _ = oracle.local_faldoi(i0s[1], i1s[1], gos[1], bas[1])
    assert bits_equal(plain, wants[1][0])                  # the fall-back pair is the unpartitioned run


def test_match_growing_partitions_fall_back_to_the_whole_image(ctx, oracle):
    # no backward seed at all: some partition queue never receives a candidate, so every partitioned iteration reverts
    # to whole-image growings (src/local_faldoi.cpp:1349-1370) while the partitions' forward queues pile up unused
    i0, i1 = sintel_pair((520, 210, 84, 60))
    a, b = oracle.preprocess_local(i0, i1)
    go, _ = _seed_flows(oracle, a, b, 0.02, seed=77)
    none = np.full_like(go, np.nan)
    want, wene, st = oracle.local_faldoi(a, b, go, none, split_img=1, h_parts=2, v_parts=2)
    got, gene, gst = ctx.match_growing(a, b, go, none, preprocessed=1, split_img=1, h_parts=2, v_parts=2)
    assert list(gst.solves) == [st[1], st[4]] and list(gst.pops) == [st[0], st[3]]
    assert bits_equal(got, want), mismatch_report(got, want)
    plain, _, _ = oracle.local_faldoi(a, b, go, none)
    assert bits_equal(plain, want)                          # with the fall-back it is the unpartitioned run


# ------------------------------------------------------------------ BASELINE configs at their own size
def test_full_size_pipeline_final_hard_reference_defaults(ctx, oracle, tmp_path):
    """BASELINE configs[2]: the whole chain on the Sintel final/hard example pair at 1024x436 with the reference's
    defaults (-m 0 -wr 5 -loc_it 3 -max_pch_it 4 -split_img 0 -fb_thresh 2, then global -w 5): `faldoi_flow` from match
    lists; the local flow, the energy map and the final flow are bit-identical to the CPU checker's.  (Slow: the two
    whole-image growings are sequential on both sides.  The matcher binaries cannot run here, so the match lists are
    ~1.1 % of a dense TV-L1 flow rounded to integers.)"""
    import os
    import time
    from util import GOLDEN, load_rgb_planar
    bins = F.build.build_host()
    f2 = os.path.join(GOLDEN, "sintel_final_hard_frame_0002.png")
    f3 = os.path.join(GOLDEN, "sintel_final_hard_frame_0003.png")
    i0, i1 = load_rgb_planar(f2), load_rgb_planar(f3)
    h, w = i0.shape[1:]
    assert (w, h) == (1024, 436)
    a, b = ctx.preprocess_local(i0, i1)
    oa, ob = oracle.preprocess_local(i0, i1)
    assert bits_equal(a, oa) and bits_equal(b, ob)
    rng = np.random.default_rng(1)
    for name, (src, dst) in (("m0.txt", (a, b)), ("m1.txt", (b, a))):
        u = ctx.tvl1_global(src, dst, np.zeros((2, h, w), np.float32), warps=5)[0]
        ys, xs = np.nonzero(rng.random((h, w)) < 0.011)
        with open(tmp_path / name, "w") as f:
            for y, x in zip(ys, xs):
                f.write("%d %d %d %d\n" % (x, y, x + int(round(float(u[0, y, x]))), y + int(round(float(u[1, y, x])))))
    (tmp_path / "ims.txt").write_text(f2 + "\n" + f3 + "\n")
    t0 = time.time()
    r = subprocess.run([bins["faldoi_flow"], str(tmp_path / "ims.txt"), str(tmp_path / "m0.txt"), str(tmp_path / "m1.txt"),
                        str(tmp_path / "out.flo"), "-matcher", "cut", "-local_out", str(tmp_path / "local.flo"), "-sim_out",
                        str(tmp_path / "sim.pgm"), "-verbose", "1"], capture_output=True, text=True)
    gpu_s = time.time() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    for m, o in (("m0.txt", "s0.flo"), ("m1.txt", "s1.flo")):
        rr = subprocess.run([bins["sparse_flow"], str(tmp_path / m), str(w), str(h), str(tmp_path / o)], capture_output=True, text=True)
        assert rr.returncode == 0, rr.stderr
    go, ba = F.flo.read_flo(str(tmp_path / "s0.flo")), F.flo.read_flo(str(tmp_path / "s1.flo"))
    assert np.isfinite(go[0]).sum() > 4000 and np.isfinite(ba[0]).sum() > 4000
    t0 = time.time()
    want_local, want_ene, st = oracle.local_faldoi(oa, ob, go, ba)
    cpu_s = time.time() - t0
    got_local = F.flo.read_flo(str(tmp_path / "local.flo"))
    assert bits_equal(got_local, want_local), mismatch_report(got_local, want_local)
    sim = open(tmp_path / "sim.pgm").read().split()
    assert sim[:4] == ["P2", str(w), str(h), "65535"]
    got_ene = np.array([float.fromhex(v) for v in sim[4:]], np.float32)
    assert bits_equal(got_ene.reshape(h, w), want_ene)
    g0, g1 = oracle.preprocess_global(i0, i1, None)
    u = want_local.copy()
    xi = np.zeros((4, h, w), np.float32)
    its, errs = np.zeros(5, np.int32), np.zeros(5, np.float32)
    oracle.lib().orc_tvl1_global(g0, g1.copy(), u[0], u[1], xi[0], xi[1], xi[2], xi[3], 40, 0.3, 0.125, 0.01, w, h, 5, 400, 1, its, errs)
    got = F.flo.read_flo(str(tmp_path / "out.flo"))
    assert float(F.flo.epe(got, u).mean()) <= EPE_GATE
    assert bits_equal(got, u), mismatch_report(got, u)
    # accuracy anchor: the ground truth the reference ships for this pair (example_data/final/hard/gt, committed fixture).  With
    # these matcher-free seeds (1.1 % of a dense TV-L1 flow of a 39-px-motion pair) the chain lands at ~24.6 px -- between the zero
    # flow's 31.99 and what real matches would give; the bound catches a broken chain, the bits above catch everything else
    gt = np.load(GOLDEN + "/sintel_final_hard_gt_frame_0002.npz")["gt"]
    epe_gt = float(F.flo.epe(got, gt).mean())
    assert abs(float(F.flo.epe(np.zeros_like(gt), gt).mean()) - 31.9898) < 1e-3
    assert 20.0 < epe_gt < 28.0, epe_gt
    print("configs[2] 1024x436 -split_img 0: EPE vs ground truth %.4f px (zero flow 31.99)" % epe_gt)
    print("configs[2] 1024x436 -split_img 0: faldoi_flow %.1f s wall (%s); CPU checker local step %.1f s on 2 threads; "
          "%d windows" % (gpu_s, "; ".join(l for l in r.stdout.splitlines() if l.startswith("faldoi_flow")), cpu_s, st[1] + st[4]))


def test_nltvcsad_full_size_bounded_schedule_and_determinism(ctx, oracle):
    """BASELINE configs[4] at its own size (Sintel clean/easy 1024x436, -m 6): a bounded schedule (1 warp x 40 iterations)
    bit-identical to the CPU checker, and the full 1 x 400 schedule twice with identical bits (no atomics, no
    order-dependent reduction anywhere in the two streaming kernels)."""
    i0, i1 = sintel_pair()
    h, w = i0.shape[1:]
    assert (w, h) == (1024, 436)
    i0n, i1n = ctx.preprocess_global(i0, i1)
    lab = F.capi.image_to_lab(i0)
    assert bits_equal(lab, oracle.image_to_lab(i0))
    u0 = np.zeros((2, h, w), np.float32)
    u, _ = ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=1, max_inner=40)
    ou = oracle.nltvcsad_global(i0n, i1n, lab, u0, warps=1, max_inner=40)
    assert float(F.flo.epe(u, ou).mean()) <= EPE_GATE
    assert bits_equal(u, ou), mismatch_report(u, ou)
    ua, ms_a = ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=1, max_inner=400)
    ub, ms_b = ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=1, max_inner=400)
    assert bits_equal(ua, ub) and np.isfinite(ua).all()
    # a warm-started second warp continues from the first: 2 x 20 equals nothing else than itself twice, and differs from 1 x 40
    uc, _ = ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=2, max_inner=20)
    oc = oracle.nltvcsad_global(i0n, i1n, lab, u0, warps=2, max_inner=20)
    assert bits_equal(uc, oc), mismatch_report(uc, oc)
    print("configs[4] 1024x436: 1 x 400 iterations %.1f / %.1f ms on the device" % (ms_a, ms_b))


# ------------------------------------------------------------------ the other local functionals
@pytest.mark.parametrize("method", [2, 3, 4, 5, 6, 7])
def test_match_growing_other_functionals_bit_exact(ctx, oracle, method):
    """The whole local step with of_estimation's other branches (-m 2..7: NLTV-L1, TV-CSAD, NLTV-CSAD, plain and
    weighted), every candidate on the device in the reference's order, against the CPU checker -- whose window
    operators and tables are pinned bit for bit to the reference's compiled objects (tests/test_oracle_models_cpu.py):
    flow, energy map and the pop / window / fill counts."""
    crop = (400, 150, 56, 40) if method < 4 else (400, 150, 40, 30)       # the CPU checker sorts 97 floats per pixel-iteration
    i0, i1 = sintel_pair(crop)
    a, b = oracle.preprocess_local(i0, i1)
    go, ba = _seed_flows(oracle, a, b, 0.02, seed=method)
    want, want_ene, st = oracle.local_faldoi_raw(i0, i1, go, ba, method)
    got, got_ene, gst = ctx.match_growing(i0, i1, go, ba, method=method, budget=2000)
    assert list(gst.pops) == [st[0], st[3]] and list(gst.solves) == [st[1], st[4]] and list(gst.fills) == [st[2], st[5]]
    assert float(F.flo.epe(got, want).mean()) <= EPE_GATE
    assert bits_equal(got, want), mismatch_report(got, want)
    assert bits_equal(got_ene, want_ene), mismatch_report(got_ene, want_ene)


@pytest.mark.parametrize("method", [3, 4])
def test_local_faldoi_cli_other_functionals(oracle, tmp_path, method):
    """`local_faldoi -m 3` (NLTV-L1, weighted) and `-m 4` (TV-CSAD) through the executable: the argv and file surface of the
    reference (`src/local_faldoi.cpp:1756-2110`), raw RGB frames in, `.flo` + energy map out, byte-identical to the CPU checker."""
    from PIL import Image
    bins = F.build.build_host()
    crop = (400, 150, 48, 36)
    i0, i1 = sintel_pair(crop)
    h, w = i0.shape[1:]
    for k, im in enumerate((i0, i1)):
        Image.fromarray(im.transpose(1, 2, 0).astype(np.uint8), "RGB").save(tmp_path / ("f%d.png" % k))
    (tmp_path / "ims.txt").write_text("%s\n%s\n" % (tmp_path / "f0.png", tmp_path / "f1.png"))
    a, b = oracle.preprocess_local(i0, i1)
    go, ba = _seed_flows(oracle, a, b, 0.02, seed=20 + method)
    F.flo.write_flo(str(tmp_path / "s0.flo"), go)
    F.flo.write_flo(str(tmp_path / "s1.flo"), ba)
    r = subprocess.run([str(c) for c in (bins["local_faldoi"], tmp_path / "ims.txt", tmp_path / "s0.flo", tmp_path / "s1.flo",
                                         tmp_path / "local.flo", tmp_path / "sim.tiff", "-m", method, "-wr", 5)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    want, want_ene, _ = oracle.local_faldoi_raw(i0, i1, go, ba, method)
    got = F.flo.read_flo(str(tmp_path / "local.flo"))
    assert bits_equal(got, want), mismatch_report(got, want)
    sim = open(tmp_path / "sim.tiff").read().split()
    assert np.array_equal(np.array([float.fromhex(v) for v in sim[4:]], np.float32), want_ene.ravel())


def test_match_growing_other_functionals_argument_checks(ctx):
    i0, i1 = sintel_pair((400, 150, 40, 30))
    z = np.full((2, 30, 40), np.nan, np.float32)
    with pytest.raises(F.capi.FaldoiError):          # partitions are a TV-l2 feature
        ctx.match_growing(i0, i1, z, z, method=2, split_img=1)
    with pytest.raises(F.capi.FaldoiError):          # Lab weights need the raw RGB frames
        ctx.match_growing(i0[0], i1[0], z, z, method=6)
    with pytest.raises(F.capi.FaldoiError):
        ctx.match_growing(i0, i1, z, z, method=8)
