"""CPU suite, part 4: the host-side executables (the reference's command-line surface) -- image
decoders against PIL, sparse_flow against its definition, global_faldoi's argument handling."""
import os
import subprocess

import numpy as np
import pytest

import faldoi_ipol_amd as F
from util import GOLDEN


@pytest.fixture(scope="module")
def bins():
    return F.build.build_host()


def _fnv(b):
    a = np.frombuffer(b, np.uint8)
    h = 1469598103934665603
    for x in a.tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def _info(bins, path):
    out = subprocess.run([bins["faldoi_imginfo"], str(path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    w, h, pd, hx = out.stdout.split()
    return int(w), int(h), int(pd), hx


def test_png_decoder_matches_pil(bins, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    cases = []
    rgb = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    Image.fromarray(rgb, "RGB").save(tmp_path / "rgb.png"); cases.append(("rgb.png", rgb.transpose(2, 0, 1)))
    gray = rng.integers(0, 256, (9, 31), dtype=np.uint8)
    Image.fromarray(gray, "L").save(tmp_path / "gray.png"); cases.append(("gray.png", gray[None]))
    rgba = rng.integers(0, 256, (7, 5, 4), dtype=np.uint8)
    Image.fromarray(rgba, "RGBA").save(tmp_path / "rgba.png"); cases.append(("rgba.png", rgba.transpose(2, 0, 1)))
    g16 = rng.integers(0, 65536, (6, 8), dtype=np.uint16)
    Image.fromarray(g16, "I;16").save(tmp_path / "g16.png"); cases.append(("g16.png", g16[None]))
    pal = Image.fromarray(rgb, "RGB").quantize(16)
    pal.save(tmp_path / "pal.png"); cases.append(("pal.png", np.asarray(pal.convert("RGB")).transpose(2, 0, 1)))
    with open(tmp_path / "a.ppm", "wb") as f:
        f.write(b"P6\n# c\n17 13\n255\n" + rgb.tobytes())
    cases.append(("a.ppm", rgb.transpose(2, 0, 1)))
    for name, planes in cases:
        want = np.ascontiguousarray(planes, dtype=np.float32)
        w, h, pd, hx = _info(bins, tmp_path / name)
        assert (pd, h, w) == want.shape, name
        assert hx == _fnv(want.tobytes()), name
    # the committed Sintel frame (interesting filters: a real-world encoder)
    frame = os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png")
    want = np.asarray(Image.open(frame).convert("RGB"), np.float32).transpose(2, 0, 1)
    w, h, pd, hx = _info(bins, frame)
    assert (w, h, pd) == (1024, 436, 3)
    assert hx == _fnv(np.ascontiguousarray(want).tobytes())
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"\x89PNG\r\n\x1a\nxxxx")
    assert subprocess.run([bins["faldoi_imginfo"], str(bad)], capture_output=True).returncode == 1


def test_sparse_flow_cli(bins, tmp_path):
    m = tmp_path / "m.txt"
    m.write_text("1.5 2.25 4.0 1.0\n0 0 -1 -2 extra columns ignored\n50 50 1 1\nnot a match\n6.9 3.1 7.9 3.6\n")
    out = tmp_path / "s.flo"
    r = subprocess.run([bins["sparse_flow"], str(m), "8", "5", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    f = F.flo.read_flo(str(out))
    assert f.shape == (2, 5, 8)
    want = np.full((2, 5, 8), np.nan, np.float32)
    want[:, 2, 1] = (2.5, -1.25)
    want[:, 0, 0] = (-1.0, -2.0)
    want[:, 3, 6] = (1.0, 0.5)                      # (50,50) is outside 8x5 and skipped
    assert np.array_equal(np.isnan(f), np.isnan(want))
    assert np.array_equal(f[~np.isnan(f)], want[~np.isnan(want)])
    assert subprocess.run([bins["sparse_flow"], str(m)], capture_output=True).returncode == 1


def test_global_faldoi_cli_argument_handling(bins, tmp_path):
    exe = bins["global_faldoi"]
    r = subprocess.run([exe, "only_one_arg"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr
    ims = tmp_path / "ims.txt"
    ims.write_text(os.path.join(GOLDEN, "sintel_clean_easy_frame_0002.png") + "\n"
                   + os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png") + "\n")
    z = tmp_path / "zero.flo"
    F.flo.write_flo(str(z), np.zeros((2, 436, 1024), np.float32))
    # option pairs may sit anywhere; unsupported functionals are refused, not silently replaced
    r = subprocess.run([exe, "-m", "8", str(ims), str(z), str(tmp_path / "o.flo")], capture_output=True, text=True)
    assert r.returncode == 2 and "not available" in r.stderr
    small = tmp_path / "small.flo"
    F.flo.write_flo(str(small), np.zeros((2, 4, 4), np.float32))
    r = subprocess.run([exe, str(ims), str(small), str(tmp_path / "o.flo")], capture_output=True, text=True)
    assert r.returncode == 1 and "flow field size mismatch" in r.stderr
    import ctypes
    n = ctypes.c_int(0)
    if F.capi.lib().faldoi_device_count(ctypes.byref(n)) != 0:       # CI container: no GPU, no fallback
        r = subprocess.run([exe, str(ims), str(z), str(tmp_path / "o.flo")], capture_output=True, text=True)
        assert r.returncode == 3 and "no usable GPU" in r.stderr
        assert not (tmp_path / "o.flo").exists()


def test_local_faldoi_cli_argument_handling(bins, tmp_path):
    exe = bins["local_faldoi"]
    r = subprocess.run([exe, "a", "b"], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    f2, f3 = (os.path.join(GOLDEN, "sintel_clean_easy_frame_000%d.png" % k) for k in (2, 3))
    ims = tmp_path / "ims.txt"
    ims.write_text(f2 + "\n" + f3 + "\n")
    seeds = np.full((2, 436, 1024), np.nan, np.float32)
    seeds[:, 100, 200] = 1.0
    s0 = tmp_path / "s0.flo"
    F.flo.write_flo(str(s0), seeds)
    base = [exe, str(ims), str(s0), str(s0), str(tmp_path / "o.flo"), str(tmp_path / "sim.tiff")]
    r = subprocess.run(base + ["-m", "8"], capture_output=True, text=True)          # the occlusion model is not built
    assert r.returncode == 2 and "not available" in r.stderr
    r = subprocess.run(base + ["occ.png"], capture_output=True, text=True)          # occlusion outputs not built
    assert r.returncode == 2
    ims3 = tmp_path / "ims3.txt"
    ims3.write_text(f2 + "\n" + f3 + "\n" + f2 + "\n")                            # src/local_faldoi.cpp:1852
    r = subprocess.run([exe, str(ims3)] + base[2:], capture_output=True, text=True)
    assert r.returncode == 1 and "3 images given as input" in r.stderr
    small = tmp_path / "small.flo"
    F.flo.write_flo(str(small), np.zeros((2, 4, 4), np.float32))
    r = subprocess.run([exe, str(ims), str(small), str(small), base[4], base[5]], capture_output=True, text=True)
    assert r.returncode == 1 and "flow field size mismatch" in r.stderr
    import ctypes
    n = ctypes.c_int(0)
    if F.capi.lib().faldoi_device_count(ctypes.byref(n)) != 0:       # CI container: no GPU, no fallback
        r = subprocess.run(base, capture_output=True, text=True)
        assert r.returncode == 3 and "no usable GPU" in r.stderr
        assert not (tmp_path / "o.flo").exists()


def test_faldoi_flow_cli_argument_handling(bins, tmp_path):
    exe = bins["faldoi_flow"]
    r = subprocess.run([exe, "ims.txt"], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    f2, f3 = (os.path.join(GOLDEN, "sintel_clean_easy_frame_000%d.png" % k) for k in (2, 3))
    ims = tmp_path / "ims.txt"
    ims.write_text(f2 + "\n" + f3 + "\n")
    m = tmp_path / "m.txt"
    m.write_text("10 20 1 1 12 21 1 1 5\n")
    r = subprocess.run([exe, str(ims), str(m), str(m), str(tmp_path / "o.flo"), "-matcher", "surf"], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    r = subprocess.run([exe, str(ims), str(tmp_path / "missing.txt"), str(m), str(tmp_path / "o.flo")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr
    import ctypes
    n = ctypes.c_int(0)
    if F.capi.lib().faldoi_device_count(ctypes.byref(n)) != 0:       # CI container: no GPU, no fallback
        r = subprocess.run([exe, str(ims), str(m), str(m), str(tmp_path / "o.flo")], capture_output=True, text=True)
        assert r.returncode == 3 and "no usable GPU" in r.stderr


def test_global_faldoi_parameter_file_and_occlusion_argument_form(bins, tmp_path):
    """-p params.txt as init_params reads it (src/utils_preprocess.cpp:65-155): nine lines lambda, theta, tau, beta, alpha,
    tau_u, tau_eta, tau_chi, mu; a value <= 0 falls back to the default, tau > 0.25 too; a short or malformed file
    makes stof throw.  The 5-positional form (occl_input occl_out, src/global_faldoi.cpp:1865-1876) is accepted for
    methods 0-7, which never touch those two paths."""
    exe = bins["global_faldoi"]
    ims = tmp_path / "ims.txt"
    ims.write_text(os.path.join(GOLDEN, "sintel_clean_easy_frame_0002.png") + "\n"
                   + os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png") + "\n")
    z = tmp_path / "zero.flo"
    F.flo.write_flo(str(z), np.zeros((2, 436, 1024), np.float32))
    o = str(tmp_path / "o.flo")

    def resolved(text, extra=()):
        pf = tmp_path / "params.txt"
        pf.write_text(text)
        return subprocess.run([exe, str(ims), str(z), o, "-p", str(pf), "-print_params", "1", *extra],
                              capture_output=True, text=True)

    nine = "35\n0.25\n0.2\n1\n1\n0.1\n0.1\n0.1\n1\n"
    r = resolved(nine)
    assert r.returncode == 0 and r.stdout.split() == ["lambda", "35", "theta", "0.25", "tau", "0.200000003", "tol", "0.00999999978"]
    r = resolved("-1\n0\n0.3\n1\n1\n0.1\n0.1\n0.1\n1\n")            # <= 0 -> default; tau > 0.25 -> default
    assert r.returncode == 0 and r.stdout.split() == ["lambda", "40", "theta", "0.300000012", "tau", "0.125", "tol", "0.00999999978"]
    r = resolved("35\n0.25\n0.25\n1\n1\n0.1\n0.1\n0.1\n1\n")         # tau == 0.25 is kept
    assert r.returncode == 0 and r.stdout.split()[5] == "0.25"
    r = resolved("35\n0.25\n0.2\n")                                   # three lines: the reference's stof("") throws
    assert r.returncode == 1 and "stof" in r.stderr
    r = resolved("forty\n" + nine)
    assert r.returncode == 1
    # occlusion form: five positional arguments, options anywhere
    r = subprocess.run([exe, str(ims), "-m", "0", str(z), o, str(tmp_path / "occ_in.png"), str(tmp_path / "occ_out.png"),
                        "-print_params", "1"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("lambda 40 ")
    r = subprocess.run([exe, str(ims), str(z), o, "one_too_many"], capture_output=True, text=True)
    assert r.returncode == 1 and "With occlusions" in r.stderr


# ------------------------------------------------------------------ the C++ batch host: its rank logic on CPU
def _batch_inputs(tmp_path, njobs, pipeline, rng, w=19, h=11):
    from PIL import Image
    lines, want = [], []
    for j in range(njobs):
        names = []
        for k in range(2):
            p = tmp_path / ("f%d_%d.png" % (j, k))
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB").save(p)
            names.append(str(p))
        flows = []
        for k in range(2 if pipeline else 1):
            fl = rng.standard_normal((2, h, w)).astype(np.float32) + j
            p = tmp_path / ("in%d_%d.flo" % (j, k))
            F.flo.write_flo(str(p), fl)
            flows.append(fl)
            names.append(str(p))
        if not pipeline and j == 1:
            names[2] = "zero"                                   # the 2-line list-file form without an initial flow
            flows[0] = np.zeros((2, h, w), np.float32)
        out = tmp_path / ("out%d.flo" % j)
        lines.append(" ".join(names + [str(out)]))
        want.append((out, flows[0] + np.float32(1)))           # the stub solver: first flow of the job + 1
    jobs = tmp_path / "jobs.txt"
    jobs.write_text("# a comment line\n" + "\n".join(lines) + "\n")
    return jobs, want


@pytest.mark.parametrize("gpus", [1, 2, 3])
@pytest.mark.parametrize("pipeline", [False, True])
def test_batch_host_rank_logic_with_stub_communicator(bins, tmp_path, gpus, pipeline):
    """faldoi_batch (host/faldoi_batch.cpp) built with -DFALDOI_BATCH_STUB: the same source with pipes in place of RCCL and
    `flow + 1` in place of the solver.  Job table broadcast, round-robin shards with a ragged last round (5 jobs on 1 / 2 / 3
    ranks), padded scatter, gather, job -> output file mapping, both job forms."""
    rng = np.random.default_rng(10 * gpus + int(pipeline))
    jobs, want = _batch_inputs(tmp_path, 5, pipeline, rng)
    r = subprocess.run([bins["faldoi_batch_stub"], str(jobs), "-gpus", str(gpus)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "5 pairs on %d GPU(s) written" % gpus in r.stdout
    for path, flow in want:
        got = F.flo.read_flo(str(path))
        assert np.array_equal(got.view(np.uint32), flow.view(np.uint32)), path


def test_batch_host_failures_reach_every_rank(bins, tmp_path):
    """Rank 0 cannot read an input: every rank learns it from the header broadcast and leaves with exit code 3 (nobody is
    left waiting in a collective); bad job tables and usage errors are reported as such."""
    rng = np.random.default_rng(3)
    jobs, _ = _batch_inputs(tmp_path, 3, False, rng)
    text = jobs.read_text().replace("f1_0.png", "missing.png")
    jobs.write_text(text)
    r = subprocess.run([bins["faldoi_batch_stub"], str(jobs), "-gpus", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "cannot read the jobs" in r.stderr
    (tmp_path / "mixed.txt").write_text("a b c d\na b c d e\n")
    r = subprocess.run([bins["faldoi_batch_stub"], str(tmp_path / "mixed.txt"), "-gpus", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "cannot be mixed" in r.stderr
    r = subprocess.run([bins["faldoi_batch_stub"]], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "usage" in r.stderr


def test_batch_host_without_a_gpu_ends_every_rank(bins, tmp_path):
    """The RCCL build of faldoi_batch on a machine without a GPU (this container): rank 0 cannot get a communicator id, says so and
    exits 3; the ranks it had already started find their pipe closed and leave with 3 as well -- nobody is left waiting."""
    import ctypes as C
    n = C.c_int(0)
    if F.capi.lib().faldoi_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present: the GPU suite runs this program for real")
    rng = np.random.default_rng(5)
    jobs, _ = _batch_inputs(tmp_path, 2, False, rng)
    for gpus in ("1", "3"):
        r = subprocess.run([bins["faldoi_batch"], str(jobs), "-gpus", gpus], capture_output=True, text=True, timeout=120)
        assert r.returncode == 3, (gpus, r.returncode, r.stderr[-500:])
        assert "faldoi_batch:" in r.stderr


@pytest.mark.parametrize("njobs", [32, 29])
def test_batch_host_eight_ranks_32_jobs(bins, tmp_path, njobs):
    """BASELINE configs[3]'s shape on the rank logic alone: 8 ranks, 32 jobs (four per rank, exact) and 29 (ragged last round:
    ranks 5..7 hold three), stub communicator; every output lands in its own file with its own job's content."""
    rng = np.random.default_rng(800 + njobs)
    jobs, want = _batch_inputs(tmp_path, njobs, False, rng)
    r = subprocess.run([bins["faldoi_batch_stub"], str(jobs), "-gpus", "8"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "%d pairs on 8 GPU(s) written" % njobs in r.stdout
    for path, flow in want:
        got = F.flo.read_flo(str(path))
        assert np.array_equal(got.view(np.uint32), flow.view(np.uint32)), path


def _no_stub_left():
    out = subprocess.run(["ps", "-eo", "pid,comm"], capture_output=True, text=True).stdout
    return not any("faldoi_batch_st" in line for line in out.splitlines())


@pytest.mark.parametrize("hook,rank", [("FALDOI_BATCH_TEST_HANG_RANK", 1), ("FALDOI_BATCH_TEST_HANG_RANK", 2),
                                       ("FALDOI_BATCH_TEST_DIE_RANK", 2), ("FALDOI_BATCH_TEST_DIE_RANK", 0)])
def test_batch_host_never_waits_for_ever(bins, tmp_path, hook, rank):
    """A rank that hangs or dies between two collectives (test hooks): the ranks waiting for it give up after
    FALDOI_BATCH_TIMEOUT_S (watchdog, exit code 4) or see their pipe close (3); rank 0 ends whatever is left; the program
    returns non-zero within seconds and leaves no process behind."""
    import time
    rng = np.random.default_rng(77)
    jobs, _ = _batch_inputs(tmp_path, 6, False, rng)
    env = dict(os.environ, FALDOI_BATCH_TIMEOUT_S="2")
    env[hook] = str(rank)
    t0 = time.time()
    r = subprocess.run([bins["faldoi_batch_stub"], str(jobs), "-gpus", "3"], capture_output=True, text=True, timeout=60, env=env)
    took = time.time() - t0
    assert r.returncode != 0, (hook, rank, r.stdout, r.stderr)
    assert took < 20, took
    if hook.endswith("HANG_RANK"):
        assert "no progress in" in r.stderr and "FALDOI_BATCH_TIMEOUT_S" in r.stderr, r.stderr
    for _ in range(50):
        if _no_stub_left():
            break
        time.sleep(0.1)
    assert _no_stub_left()


def test_batch_host_rendezvous_file_ignores_a_stale_record(bins, tmp_path):
    """The launcher-mode rendezvous (the RCCL id travels through a file): a record left behind by a rank 0 that no longer
    exists is not accepted -- the reader keeps polling until a live rank 0 has published --, a record whose writer is not the
    reader's sibling is not accepted under the default (per-launcher) name, rank 0 replaces what it finds and removes its
    file when it leaves, and a reader without any rank 0 gives up after FALDOI_BATCH_TIMEOUT_S."""
    import struct
    import time
    exe = bins["faldoi_batch_stub"]
    path = tmp_path / "rv.id"
    dead = subprocess.Popen(["true"])
    dead.wait()
    stale = struct.pack("<Qq", 0x46414c444f494231, dead.pid) + b"stale".ljust(256, b"\0")
    path.write_bytes(stale)
    env = dict(os.environ, FALDOI_BATCH_TIMEOUT_S="20")
    reader = subprocess.Popen([exe, "--rendezvous-selftest", str(path), "fetch", "-"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    time.sleep(0.5)
    assert reader.poll() is None                                   # still polling: the stale record was not taken
    writer = subprocess.Popen([exe, "--rendezvous-selftest", str(path), "publish", "fresh-token"], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, text=True, env=env)
    assert writer.stdout.readline().strip() == "published"
    out, err = reader.communicate(timeout=30)
    assert reader.returncode == 0 and out.strip() == "fresh-token", (out, err)
    # a sibling check: this test process is the parent of both, so a sibling fetch succeeds ...
    r = subprocess.run([exe, "--rendezvous-selftest", str(path), "fetch-sibling", "-"], capture_output=True, text=True, timeout=30, env=env)
    assert r.returncode == 0 and r.stdout.strip() == "fresh-token"
    # ... and one started by another parent (a shell in between) does not accept the record
    r = subprocess.run(["sh", "-c", "%s --rendezvous-selftest %s fetch-sibling - ; exit $?" % (exe, path)], capture_output=True, text=True,
                       timeout=30, env=dict(env, FALDOI_BATCH_TIMEOUT_S="1"))
    assert r.returncode == 3 and "no rendezvous record of a live rank 0" in r.stderr
    writer.stdin.close()
    assert writer.wait(timeout=30) == 0
    assert not path.exists()                                       # rank 0 unlinks its record however it leaves
    r = subprocess.run([exe, "--rendezvous-selftest", str(path), "fetch", "-"], capture_output=True, text=True, timeout=30,
                       env=dict(env, FALDOI_BATCH_TIMEOUT_S="1"))
    assert r.returncode == 3
