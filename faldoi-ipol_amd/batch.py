"""Batch mode: many independent image pairs sharded over the GPUs of one node (BASELINE configs[3]).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        faldoi-ipol_amd/batch.py jobs.txt [-w 5] [-max_inner 400] [-tol_stop 1] [-split_img 0|1|2] [-h_parts 3] [-v_parts 2]

jobs.txt: one job per line, all of one form and one image size:
  `I0.png I1.png in_flow.flo|zero out.flo`   what one `global_faldoi` call takes (2-line list-file semantics);
  `I0.png I1.png s0.flo s1.flo out.flo`      the whole chain of the driver scripts from the sparse seed flows:
                                             `local_faldoi` (-m 0, all the growings of a shard in one batch, on image
                                             partitions with -split_img) then `global_faldoi` on its result.

Data path: rank 0 decodes the inputs and SCATTERS them over RCCL
(uint8 frames + fp32 initial flows, one message per rank), every rank prepares and solves its
shard in one batched launch sequence on its own GPU, flows are GATHERED to rank 0, which writes the
.flo files.  No collective runs during a solve; pairs never talk to each other.
The collectives go through torch.distributed ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def parse_jobs(path):
    jobs = []
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            parts = line.split()
            if not parts or parts[0].startswith("#"):
                continue
            if len(parts) not in (4, 5):
                raise ValueError("%s:%d: expected `I0 I1 in_flow|zero out.flo` or `I0 I1 s0.flo s1.flo out.flo`" % (path, ln))
            jobs.append(tuple(parts))
    if len({len(j) for j in jobs}) > 1:
        raise ValueError("%s: global-only and pipeline jobs cannot be mixed" % path)
    return jobs


def shard_sizes(njobs, world):
    """Round-robin assignment (shard.assign_pairs) expressed as per-rank job index lists."""
    return [list(range(r, njobs, world)) for r in range(world)]


def scatter_inputs(dist, torch, device, rank, world, frames, flows, shape):
    """frames: (njobs, 2, 3, h, w) uint8 on rank 0 (None elsewhere); flows: (njobs, 2, h, w) float32
    (initial flows) or (njobs, 4, h, w) (forward and backward seed flows).  shape: (njobs, h, w[, planes]).
    Returns this rank's (frames_shard, flows_shard) as numpy arrays."""
    planes = shape[3] if len(shape) > 3 else 2
    njobs, h, w = shape[:3]
    idx = shard_sizes(njobs, world)
    cap = max(len(i) for i in idx)                       # equal-size messages: pad the short shards
    my_fr = torch.empty((cap, 2, 3, h, w), dtype=torch.uint8, device=device)
    my_fl = torch.empty((cap, planes, h, w), dtype=torch.float32, device=device)
    if rank == 0:
        fr_list, fl_list = [], []
        for r in range(world):
            pad = cap - len(idx[r])
            sel = idx[r] + [idx[r][-1] if idx[r] else 0] * pad
            fr_list.append(torch.from_numpy(np.ascontiguousarray(frames[sel])).to(device))
            fl_list.append(torch.from_numpy(np.ascontiguousarray(flows[sel])).to(device))
    else:
        fr_list = fl_list = None
    dist.scatter(my_fr, fr_list, src=0)
    dist.scatter(my_fl, fl_list, src=0)
    n_mine = len(idx[rank])
    return my_fr[:n_mine].cpu().numpy(), my_fl[:n_mine].cpu().numpy()


def gather_flows(dist, torch, device, rank, world, flows_mine, shape):
    """Inverse of scatter_inputs for the results: returns (njobs, 2, h, w) on rank 0, None elsewhere."""
    njobs, h, w = shape[:3]
    idx = shard_sizes(njobs, world)
    cap = max(len(i) for i in idx)
    buf = torch.zeros((cap, 2, h, w), dtype=torch.float32, device=device)
    if len(flows_mine):
        buf[:len(flows_mine)] = torch.from_numpy(np.ascontiguousarray(flows_mine)).to(device)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    res = np.empty((njobs, 2, h, w), np.float32)
    for r in range(world):
        for k, j in enumerate(idx[r]):
            res[j] = out[r][k].cpu().numpy()
    return res


def load_jobs(path, F):
    """Rank 0: decode every job's frames and flows.  Returns (jobs, frames uint8 (n,2,3,h,w), flows (n,2|4,h,w))."""
    from PIL import Image
    jobs = parse_jobs(path)
    if not jobs:
        raise ValueError("%s: no jobs" % path)
    fr = [[np.asarray(Image.open(p).convert("RGB")).transpose(2, 0, 1) for p in job[:2]] for job in jobs]
    if len({f.shape for pair in fr for f in pair}) != 1:
        raise ValueError("%s: all frames of a batch must have one size" % path)
    frames = np.ascontiguousarray(np.array(fr, dtype=np.uint8))
    h, w = frames.shape[-2:]
    if len(jobs[0]) == 5:
        flows = [np.concatenate([F.flo.read_flo(job[2]), F.flo.read_flo(job[3])]) for job in jobs]
    else:
        flows = [np.zeros((2, h, w), np.float32) if job[2] == "zero" else F.flo.read_flo(job[2]) for job in jobs]
    for job, fl in zip(jobs, flows):
        if fl.shape[-2:] != (h, w):
            raise ValueError("%s: flow of job %s is %dx%d, frames are %dx%d" % (path, job[-1], fl.shape[-1], fl.shape[-2], w, h))
    return jobs, frames, np.stack(flows)


def solve_shard_gpu(frames, flows, local_rank, local_params=None, **params):
    """frames (n,2,3,h,w) uint8, flows (n,2,h,w) -> refined flows (n,2,h,w) + per-pair iteration counts."""
    import faldoi_ipol_amd as F
    n = len(frames)
    if n == 0:
        return np.empty((0,) + flows.shape[1:], np.float32), []
    h, w = frames.shape[-2:]
    ctx = F.Context(local_rank)
    if flows.shape[1] == 4:                 # pipeline jobs: the local step first, every growing of the shard at once
        i0s = [frames[i, 0].astype(np.float32) for i in range(n)]
        i1s = [frames[i, 1].astype(np.float32) for i in range(n)]
        local, _, _ = ctx.match_growing_batch(i0s, i1s, [flows[i, :2] for i in range(n)], [flows[i, 2:] for i in range(n)],
                                              **(local_params or {}))
        flows = np.stack(local)
    b = F.Tvl1Batch(ctx, n, w, h)
    for i in range(n):
        b.upload_images(i, frames[i, 0].astype(np.float32), frames[i, 1].astype(np.float32), flows[i])
    iters, errs, ms, nl = b.solve(**params)
    out = np.stack([b.download(i) for i in range(n)])
    b.close()
    ctx.close()
    return out, [list(map(int, it)) for it in iters]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    sys.path.insert(0, os.path.dirname(_HERE))
    import torch
    import torch.distributed as dist
    import faldoi_ipol_amd as F

    def opt(name, default):
        if "-" + name in argv:
            i = argv.index("-" + name)
            v = argv[i + 1]
            del argv[i:i + 2]
            return v
        return default

    params = dict(warps=int(opt("w", 5)), max_inner=int(opt("max_inner", 400)), tol_stop=int(opt("tol_stop", 1)))
    local_params = dict(split_img=int(opt("split_img", 0)), h_parts=int(opt("h_parts", 3)), v_parts=int(opt("v_parts", 2)),
                        first_h_parts=int(opt("first_h_parts", 0)), first_v_parts=int(opt("first_v_parts", 0)))
    if len(argv) != 1:
        raise SystemExit(__doc__)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", device_id=device)

    meta = [None]
    frames = flows = None
    if rank == 0:
        # a failure while reading the inputs must not leave the other ranks blocked in the broadcast:
        # rank 0 always broadcasts, either the job table or an error marker, and every rank leaves together
        try:
            jobs, frames, flows = load_jobs(argv[0], F)
            h, w = frames.shape[-2:]
            meta = [(len(jobs), h, w, [job[-1] for job in jobs], flows.shape[1])]
        except Exception as e:                                   # noqa: BLE001 -- forwarded to every rank
            meta = [("error", "%s: %s" % (type(e).__name__, e))]
    dist.broadcast_object_list(meta, src=0)                     # the job table
    if meta[0][0] == "error":
        if rank == 0:
            sys.stderr.write("batch: cannot read the jobs: %s\n" % meta[0][1])
        dist.destroy_process_group()
        raise SystemExit(3)
    njobs, h, w, outs, planes = meta[0]
    my_fr, my_fl = scatter_inputs(dist, torch, device, rank, world, frames, flows, (njobs, h, w, planes))
    res, iters = solve_shard_gpu(my_fr, my_fl, local_rank, local_params=local_params, **params)
    allres = gather_flows(dist, torch, device, rank, world, res, (njobs, h, w))
    if rank == 0:
        for j, path in enumerate(outs):
            F.flo.write_flo(path, allres[j])
        print("batch: %d pairs on %d GPU(s) written" % (njobs, world))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
