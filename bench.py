#!/usr/bin/env python3
"""bench.py -- headline benchmark of the TV-L1 global minimisation hot path on MI355X.

A "step" is ONE full solve of BASELINE.json configs[1]: a 1024x436 pair, 300 warps x 10 inner
primal-dual iterations, tolerance stop off, fp32 -- i.e. 3000 inner iterations over 446 464 px.
Inputs are resident in HBM before the timed region starts.  With --gpus N every rank solves its
own pair (weak scaling; pairs are independent so there is no data-path collective).

Prints ONE JSON line (rank 0).  See DESIGN.md section 6 for how `roofline` is computed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 1024, 436
WARPS, INNER = 300, 10
BYTES_PER_PX_ITER = 76        # 19 fp32 planes touched once per inner iteration
BYTES_PER_PX_WARP = 44        # per outer warp (bicubic gathers + constants)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s
# HBM bytes per launch of tvl1_chunk_kernel from rocprofv3 PMC passes (profiles/): filled in from the
# committed counter summaries; None until measured for the current kernel.
TRAFFIC_BYTES_PER_LAUNCH = 36303872   # (2*10739 + 13975) KB, profiles/r01_pmc_sq.md (1024x436, k=10)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs-per-gpu", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--growing-pairs", type=int, default=512)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import numpy as np
    import torch
    import torch.distributed as dist
    import faldoi_ipol_amd as F
    from util import GOLDEN, load_rgb_planar, synthetic_pair

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ      # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev)               # "nccl" is RCCL on ROCm

    npairs = args.pairs_per_gpu
    ctx = F.Context(local_rank)
    batch = F.Tvl1Batch(ctx, npairs, W, H)
    # Pair 0 of rank 0 is the reference's own example pair (Sintel clean/easy frames 2->3, committed
    # under tests/golden), prepared by the library's device-side preprocessing; every other pair
    # (more pairs per GPU, other ranks) is a seeded synthetic pair of the same size.
    pairs = []
    f2 = os.path.join(GOLDEN, "sintel_clean_easy_frame_0002.png")
    f3 = os.path.join(GOLDEN, "sintel_clean_easy_frame_0003.png")
    data_kind = "synthetic"
    for i in range(npairs):
        u0 = np.zeros((2, H, W), np.float32)
        if rank == 0 and i == 0 and os.path.exists(f2) and os.path.exists(f3):
            i0n, i1n = ctx.preprocess_global(load_rgb_planar(f2), load_rgb_planar(f3))
            data_kind = "sintel clean/easy frames 2-3 (reference example_data, committed fixture), zero initial flow"
        else:
            i0n, i1n, _ = synthetic_pair(W, H, seed=1234 + rank * npairs + i, amp=4.0)
        batch.upload(i, i0n, i1n, u0)
        pairs.append((i0n, i1n, u0))
    prm = dict(warps=WARPS, max_inner=INNER, tol_stop=0, chunk=args.chunk)

    def step():
        return batch.solve(**prm)

    for _ in range(args.warmup):
        step()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    dev_ms, launches = 0.0, 0
    for _ in range(args.steps):
        iters, errs, ms, nl = step()          # returns after the solve's stop event has completed
        dev_ms += ms
        launches += nl
    fence()
    dt = time.perf_counter() - t0

    px_iters_local = float(args.steps) * npairs * W * H * WARPS * INNER
    dt_max, px_iters = F.shard.reduce_timing(dt, px_iters_local, dist if use_dist else None, dev)

    # Duration of the dominant kernel itself (outside the timed region): one more solve with a HIP
    # event pair around every launch of the fused iteration kernel, on the library's own stream.
    kernel_us = None
    if rank == 0:
        os.environ["FALDOI_EVENT_TIMING"] = "1"
        step()
        os.environ["FALDOI_EVENT_TIMING"] = "0"
        _, nl1, _, chunk_ms = batch.last_timing()
        if chunk_ms > 0:
            kernel_us = chunk_ms * 1e3 / nl1

    out = None
    if rank == 0:
        value = px_iters / dt_max / 1e6
        iters_per_launch = WARPS * INNER * args.steps / launches
        # dominant kernel = tvl1_chunk_kernel: algorithmic bytes of the inner iterations one launch
        # advances (76 B/px/iteration) over that launch's own duration
        bytes_per_launch = npairs * W * H * BYTES_PER_PX_ITER * iters_per_launch
        group_us = dev_ms * 1e3 / launches                 # whole solve / launches (incl. the per-warp kernel)
        launch_us = kernel_us if kernel_us else group_us
        achieved = bytes_per_launch / (launch_us * 1e-6) / 1e9
        out = {
            "metric": "Mpixels*warp-iters/sec on TV-L1 global_faldoi, Sintel 1024x436; EPE vs CPU ref",
            "value": round(value, 1), "unit": "Mpx*iter/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": data_kind if world == 1 and npairs == 1 else "rank 0 pair 0: " + data_kind + "; others synthetic",
            "config": {"workload": "TV-L1 global minimisation, %d pair(s)/GPU of %dx%d, %d warps x %d inner "
                                   "iterations, tolerance stop off (BASELINE configs[1])"
                                   % (npairs, W, H, WARPS, INNER),
                       "pairs_per_gpu": npairs, "parallelism": "pairs-sharded x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "tvl1_chunk_kernel",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": TRAFFIC_BYTES_PER_LAUNCH if (npairs == 1 and args.chunk in (0, 10)) else None,
                         "launch_us": round(launch_us, 2), "launch_us_source":
                             "HIP event pair around every launch" if kernel_us else "solve time / launches",
                         "solve_us_per_launch_incl_warp_kernel": round(group_us, 2),
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "inner_iters_per_launch": round(iters_per_launch, 2)},
        }
        if world == 1 and data_kind != "synthetic" and not args.no_secondary:
            # BASELINE configs[4] (not the headline): NLTV-CSAD, same pair, 1 warp x 400 iterations
            lab = F.capi.image_to_lab(load_rgb_planar(f2))
            i0n, i1n, u0 = pairs[0]
            ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=1, max_inner=20)            # warm-up
            _, nl_ms = ctx.nltvcsad_global(i0n, i1n, lab, u0, warps=1, max_inner=400)
            out["secondary"] = {"nltvcsad_1024x436_1warp_x_400": {
                "device_ms": round(nl_ms, 2), "Mpx_iter_per_s": round(W * H * 400 / nl_ms / 1e3, 1),
                "algorithmic_GBps": round(720.0 * W * H * 400 / (nl_ms * 1e-3) / 1e9, 1),
                "note": "two streaming launches per iteration, ~720 B/px/iteration (SURVEY 8d, approximate)"}}
        if world == 1 and data_kind != "synthetic" and not args.no_secondary:
            # SURVEY 8f-1 (not the headline): the local step (match growing).  One growing is a sequential
            # priority-queue process (one workgroup each); the device runs the growings of many pairs at once.
            gp, gw, gh = args.growing_pairs, 256, 192
            full0, full1 = load_rgb_planar(f2), load_rgb_planar(f3)
            rng = np.random.default_rng(7)
            g_i0, g_i1, g_go, g_ba = [], [], [], []
            for k in range(gp):
                x0, y0 = (k * 37) % (W - gw), (k * 53) % (H - gh)
                c0 = np.ascontiguousarray(full0[:, y0:y0 + gh, x0:x0 + gw])
                c1 = np.ascontiguousarray(full1[:, y0:y0 + gh, x0:x0 + gw])
                a, b = ctx.preprocess_local(c0, c1)
                sd = []
                for src, dst in ((a, b), (b, a)):           # seeds: ~1 % of a dense TV-L1 flow, rounded (matcher-like)
                    dense = ctx.tvl1_global(src, dst, np.zeros((2, gh, gw), np.float32), warps=3)[0]
                    sp = np.full_like(dense, np.nan)
                    m = rng.random((gh, gw)) < 0.011
                    sp[:, m] = np.round(dense[:, m])
                    sd.append(sp)
                g_i0.append(c0); g_i1.append(c1); g_go.append(sd[0]); g_ba.append(sd[1])
            g_out, _, gst = ctx.match_growing_batch(g_i0, g_i1, g_go, g_ba)
            windows = int(sum(gst.solves))
            grow = {"pairs": gp, "size": "%dx%d" % (gw, gh), "windows": windows, "device_ms": round(gst.growing_ms, 1),
                    "windows_per_s": round(windows / (gst.growing_ms * 1e-3), 0),
                    "note": "seeds -> 3 x (growing fwd+bwd, FB pruning, re-seeding) -> last growing; every candidate on the "
                            "device, in the reference's order"}
            if not args.no_cpu_baseline:
                from oracle import oracle as O
                a, b = O.preprocess_local(g_i0[0], g_i1[0])
                t1 = time.perf_counter()
                want, _, ost = O.local_faldoi(a, b, g_go[0], g_ba[0])
                cpu_s = time.perf_counter() - t1
                grow["cpu_windows_per_s"] = round((ost[1] + ost[4]) / cpu_s, 0)
                grow["cpu_sample"] = "pair 0 once, forward and backward on 2 threads (%.1f s)" % cpu_s
                grow["bit_identical_pair0"] = bool(np.array_equal(g_out[0].view(np.uint32), want.view(np.uint32)))
            out.setdefault("secondary", {})["local_match_growing"] = grow
            # one full-size pair: a single growing is sequential, so one pair only gets fast on image partitions
            # (-split_img 2, additive: partitions from the first local iteration on)
            a, b = ctx.preprocess_local(full0, full1)
            sd = []
            for src, dst in ((a, b), (b, a)):
                dense = ctx.tvl1_global(src, dst, np.zeros((2, H, W), np.float32), warps=5)[0]
                sp = np.full_like(dense, np.nan)
                m = rng.random((H, W)) < 0.011
                sp[:, m] = np.round(dense[:, m])
                sd.append(sp)
            _, _, st1 = ctx.match_growing(full0, full1, sd[0], sd[1], split_img=2, h_parts=8, v_parts=6)
            out["secondary"]["local_match_growing_one_pair"] = {
                "size": "%dx%d" % (W, H), "mode": "-split_img 2 -h_parts 8 -v_parts 6 (96 growings in flight)",
                "windows": int(sum(st1.solves)), "device_ms": round(st1.growing_ms, 1),
                "windows_per_s": round(sum(st1.solves) / (st1.growing_ms * 1e-3), 0),
                "note": "whole-image growings (the reference's default): 34 s for the same pair; its own -split_img 1 "
                        "with 3x2 partitions: 16 s (tools/split_study.py)"}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O            # checker + timed CPU baseline only
            i0n, i1n, u0 = pairs[0]
            O.lib()
            t1 = time.perf_counter()
            ou, oit, _, _ = O.tvl1_global(i0n, i1n, u0, warps=WARPS, max_inner=INNER, tol_stop=False)
            cpu_s = time.perf_counter() - t1
            gu = batch.download(0)
            epe = float(F.flo.epe(gu, ou).mean())
            out["cpu_baseline"] = {"value": round(W * H * WARPS * INNER / cpu_s / 1e6, 1), "unit": "Mpx*iter/s",
                                   "cores": int(O.lib().orc_num_threads()), "kind": "port",
                                   "sample": "the full step once (1 pair, %d warps x %d iterations, %.1f s)"
                                             % (WARPS, INNER, cpu_s)}
            # SURVEY 8d also asks for the one-thread figure: a tenth of the schedule (30 warps x 10 iterations)
            nthreads = int(O.lib().orc_num_threads())
            O.lib().orc_set_num_threads(1)
            t1 = time.perf_counter()
            O.tvl1_global(i0n, i1n, u0, warps=WARPS // 10, max_inner=INNER, tol_stop=False)
            cpu1_s = time.perf_counter() - t1
            O.lib().orc_set_num_threads(nthreads)
            out["cpu_baseline_1thread"] = {"value": round(W * H * (WARPS // 10) * INNER / cpu1_s / 1e6, 1), "unit": "Mpx*iter/s",
                                           "cores": 1, "kind": "port",
                                           "sample": "%d warps x %d iterations of the same pair (%.1f s)" % (WARPS // 10, INNER, cpu1_s)}
            out["epe_vs_cpu"] = {"mean_epe_px": epe, "bit_identical": bool(
                np.array_equal(gu.view(np.uint32), ou.view(np.uint32))), "gate": 1e-4}
        print(json.dumps(out), flush=True)
    batch.close()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
