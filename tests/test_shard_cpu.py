"""CPU suite, part 3: the N>1 path of bench.py (pair sharding + timing reduction) under
torch.distributed with the gloo backend, world_size 2."""
import os

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import faldoi_ipol_amd as F


def test_assign_pairs_round_robin_covers_every_pair_once():
    for world in (1, 2, 3, 8):
        got = sorted(p for r in range(world) for p in F.shard.assign_pairs(32, world, r))
        assert got == list(range(32))
    assert F.shard.assign_pairs(32, 8, 3) == [3, 11, 19, 27]
    with pytest.raises(ValueError):
        F.shard.assign_pairs(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import faldoi_ipol_amd as F2
    mine = F2.shard.assign_pairs(5, world, rank)
    secs = 1.0 + rank                  # rank 1 is the slow one
    t, u = F2.shard.reduce_timing(secs, 100.0 * len(mine), dist)
    dist.barrier()
    q.put((rank, t, u, mine))
    dist.destroy_process_group()


def test_reduce_timing_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t, u, mine in res:
        assert t == 2.0 and u == 500.0          # MAX of seconds, SUM of units on every rank
    assert res[0][3] == [0, 2, 4] and res[1][3] == [1, 3]


def _batch_worker(rank, world, port, q):
    import numpy as np
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import faldoi_ipol_amd as F2
    njobs, h, w = 5, 6, 8
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (njobs, 2, 3, h, w), dtype=np.uint8) if rank == 0 else None
    flows = rng.standard_normal((njobs, 2, h, w)).astype(np.float32) if rank == 0 else None
    dev = torch.device("cpu")
    # pipeline jobs carry four planes per pair (forward and backward seed flows)
    seeds = rng.standard_normal((njobs, 4, h, w)).astype(np.float32) if rank == 0 else None
    _, sd = F2.batch.scatter_inputs(dist, torch, dev, rank, world, frames, seeds, (njobs, h, w, 4))
    assert sd.shape == (len(F2.batch.shard_sizes(njobs, world)[rank]), 4, h, w)
    back = F2.batch.gather_flows(dist, torch, dev, rank, world, sd[:, 2:], (njobs, h, w))
    if rank == 0:
        assert np.array_equal(back, seeds[:, 2:])
    fr, fl = F2.batch.scatter_inputs(dist, torch, dev, rank, world, frames, flows, (njobs, h, w))
    # stand-in for the GPU solve (not available in CI): a deterministic function of the inputs
    res = fl + fr[:, 0, 0].astype(np.float32).mean(axis=(1, 2))[:, None, None, None]
    allres = F2.batch.gather_flows(dist, torch, dev, rank, world, res, (njobs, h, w))
    if rank == 0:
        want = flows + frames[:, 0, 0].astype(np.float32).mean(axis=(1, 2))[:, None, None, None]
        q.put(bool(np.array_equal(allres, want)))
    dist.barrier()
    dist.destroy_process_group()


def test_batch_scatter_gather_gloo_world2():
    """The RCCL data path of batch.py (scatter inputs, gather flows) with gloo, 2 ranks, 5 jobs:
    ragged shards (3 + 2) must come back in job order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_batch_job_file_parsing(tmp_path):
    f = tmp_path / "jobs.txt"
    f.write_text("# comment\na.png b.png zero o1.flo\n\nc.png d.png in.flo o2.flo\n")
    assert F.batch.parse_jobs(str(f)) == [("a.png", "b.png", "zero", "o1.flo"), ("c.png", "d.png", "in.flo", "o2.flo")]
    f.write_text("a.png b.png\n")
    with pytest.raises(ValueError):
        F.batch.parse_jobs(str(f))
    f.write_text("a.png b.png s0.flo s1.flo o.flo\n")
    assert F.batch.parse_jobs(str(f)) == [("a.png", "b.png", "s0.flo", "s1.flo", "o.flo")]
    f.write_text("a.png b.png s0.flo s1.flo o.flo\nc.png d.png zero o2.flo\n")
    with pytest.raises(ValueError):
        F.batch.parse_jobs(str(f))
    assert F.batch.shard_sizes(5, 2) == [[0, 2, 4], [1, 3]]


def test_bench_gpus_n_starts_n_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` as ONE plain process must start two ranks under torch.distributed.run (before
    anything touches a GPU), reduce over them and print one JSON line with n_gpus 2 (launcher self-test: gloo, stub
    step).  A WORLD_SIZE that disagrees with --gpus is refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0",
                        "--pairs-per-gpu", "4", "--launcher-selftest"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and len(out["per_rank"]) == 2
    assert out["units"] == 2 * 3 * 4                       # SUM over ranks of steps x pairs per GPU
    assert out["seconds_max"] >= 3 * 0.02 * 0.9            # MAX over ranks: rank 1 sleeps twice as long
    env["WORLD_SIZE"] = "3"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_bench_launcher_at_eight_ranks_configs3_shape():
    """BASELINE configs[3] is 8 ranks x 4 pairs: the same launcher path at N = 8 (no 8-GPU node has been available to the driver,
    so the rank plumbing of a future SCALE line is rehearsed here): one line, eight per-rank rates, SUM = 8 x steps x 4 units,
    MAX = the slowest rank's time (rank r sleeps (r + 1) x 10 ms per step)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "0",
                        "--pairs-per-gpu", "4", "--launcher-selftest"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks"] == 8 and len(out["per_rank"]) == 8
    assert out["units"] == 8 * 2 * 4
    assert out["seconds_max"] >= 2 * 0.08 * 0.9
    assert all(v > 0 for v in out["per_rank"])             # (barriers bracket the timed region: every rank's clock covers the slowest)


def test_batch_load_jobs_rejects_mismatched_inputs(tmp_path):
    """load_jobs (rank 0 of batch.py) raises on unreadable or mismatched inputs; main() turns that into an error
    marker broadcast to every rank instead of leaving them in the collective."""
    import numpy as np
    from PIL import Image
    a, b = tmp_path / "a.png", tmp_path / "b.png"
    Image.fromarray(np.zeros((6, 8, 3), np.uint8)).save(a)
    Image.fromarray(np.zeros((6, 9, 3), np.uint8)).save(b)
    jobs = tmp_path / "jobs.txt"
    jobs.write_text("%s %s zero o.flo\n" % (a, b))
    with pytest.raises(ValueError):
        F.batch.load_jobs(str(jobs), F)
    jobs.write_text("%s %s zero o.flo\n" % (a, tmp_path / "missing.png"))
    with pytest.raises(OSError):
        F.batch.load_jobs(str(jobs), F)
    F.flo.write_flo(str(tmp_path / "f.flo"), np.zeros((2, 5, 8), np.float32))
    jobs.write_text("%s %s %s o.flo\n" % (a, a, tmp_path / "f.flo"))
    with pytest.raises(ValueError):
        F.batch.load_jobs(str(jobs), F)
    jobs.write_text("%s %s zero o.flo\n" % (a, a))
    j, fr, fl = F.batch.load_jobs(str(jobs), F)
    assert fr.shape == (1, 2, 3, 6, 8) and fl.shape == (1, 2, 6, 8)
