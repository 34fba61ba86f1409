#!/usr/bin/env python3
"""Benchmark of the ORB front-end hot path on MI355X.

One step = one pass of ORB extraction (pyramid, FAST cells, quadtree,
orientation, blur, rBRIEF) over one batch of 256 synthetic 752x480 frames that
are already resident in HBM (BASELINE.json configs[1]).  `value` is whole-job
frames/s.  With --gpus N (launched by torch.distributed.run, one rank per GPU)
every rank processes its own batch: frames are independent, so there is no
data-path collective; the only communication is the timing barrier and the
max over ranks.

Extra legs reported in the same JSON line: Hamming brute-force matching
(1000x1000 per stereo pair, and 8000 x 100000), the per-stage GPU times, the
roofline of the dominant kernel and the CPU baseline (the C restatement of the
reference path timed on this node's host cores, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT, NLEVELS = 752, 480, 1000, 8
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
# algorithmic bytes per frame of the whole path: W*H + 4P - w7*h7 + 60N (P = sum of level pixels)
PATH_BYTES_PER_FRAME = 4862288
PYRAMID_PIXELS = 1117367       # P: the bytes the FAST stage has to read per frame


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frames per batch (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--single-frame", action="store_true",
                    help="also time orbx_extract on one host frame (adds small launches of every kernel, which\n"
                         "would dilute per-kernel averages in a profile of the default command)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; "
                    "gloo only to rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    return ap.parse_args()


def cpu_baseline(frames):
    """The oracle (literal restatement of the reference's CPU path, built with
    the reference's -O3 -march=native) on a bounded sample of the same frames."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    ncores = max(1, min(16, os.cpu_count() or 1))
    # single thread
    o = Oracle(NFEAT, 1.2, NLEVELS, 20, 7, native=True)
    n1 = 256
    t0 = time.perf_counter()
    for i in range(n1):
        o.extract(frames[i % len(frames)], literal=True)
    t1 = time.perf_counter()
    single = n1 / (t1 - t0)
    # one frame per thread on `ncores` threads (ctypes releases the GIL)
    oracles = [Oracle(NFEAT, 1.2, NLEVELS, 20, 7, native=True) for _ in range(ncores)]
    per = 96          # about 10 s of CPU work in total
    n_mt = per * ncores

    def work(k):
        for j in range(per):
            oracles[k].extract(frames[(k * per + j) % len(frames)], literal=True)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(ncores)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    t1 = time.perf_counter()
    multi = n_mt / (t1 - t0)
    return {"value": round(multi, 2), "unit": "frames/s", "cores": ncores, "kind": "port",
            "sample": "%d of the batch's frames, one frame per thread on %d threads (%.1f s); "
                      "single thread: %.2f frames/s over %d frames" % (n_mt, ncores, t1 - t0, single, n1),
            "single_thread_value": round(single, 2)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from orb_slam_amd import ORBextractor, ORBmatcher, KP_DTYPE
    from orb_slam_amd.synth import synthetic_frame

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the timing reduction lives

    B = args.frames
    # synthetic frames (deterministic; every rank gets its own frames)
    host = np.stack([synthetic_frame(W, H, 1234 + rank * 100000 + i) for i in range(B)])
    d_imgs = torch.from_numpy(host).to(dev)

    ex = ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, device=dev_index)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    # a dedicated (non-default) torch stream carries every kernel of the library, so that
    # torch events / synchronisation and the library's own HIP events see the same queue
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ex.set_stream(stream.cuda_stream)

    def step():
        ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # timed region: HIP events only around the launch that dominates (FAST + blur); every stage event
    # costs a little, and the roofline needs just this one duration
    ex.set_stage_timing(2)
    ex.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dom_timed = ex.stage_times()["fast_cells"]
    # per-stage breakdown: a few more steps outside the timed region with every stage bracketed
    ex.set_stage_timing(1)
    ex.reset_stage_times()
    for _ in range(min(args.steps, 5)):
        step()
    barrier()
    stages = ex.stage_times()
    ex.set_stage_timing(0)
    nkp = d_n.cpu().numpy()
    fps = world * B * args.steps / elapsed

    # the blur's work items ride in the FAST launch (k_fast_blur): one stage, the bytes of both
    per = lambda k: stages[k][0] / max(1, stages[k][1])
    if per("blur") < 0.05 * per("fast_cells"):
        stages["fast_blur"] = stages.pop("fast_cells")
        stages.pop("blur")
    # dominant kernel and its roofline (HIP events on the launch stream, same timed region)
    dom = max(stages, key=lambda k: stages[k][0])
    dom_ms = stages[dom][0] / max(1, stages[dom][1])
    if dom in ("fast_blur", "fast_cells") and dom_timed[1] > 0:
        dom_ms = dom_timed[0] / dom_timed[1]        # the duration measured inside the timed region
    stage_bytes = {  # algorithmic bytes per frame of each stage (terms of the path formula)
        "pyr_level0": W * H + W * H,
        "pyr_resize": 2 * PYRAMID_PIXELS - W * H - 210 * 134,
        "blur": 2 * PYRAMID_PIXELS,
        "fast_cells": PYRAMID_PIXELS,
        "fast_blur": 3 * PYRAMID_PIXELS,
        "quadtree": 8 * 5000,
        "orient_desc": 60 * NFEAT + 2 * 749 * NFEAT,
    }
    achieved = stage_bytes[dom] * B / (dom_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel_ms": round(dom_ms, 4),
                "path_achieved": round(fps / world * PATH_BYTES_PER_FRAME / 1e9, 2),
                "path_frac": round(fps / world * PATH_BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS, 5)}

    out = {
        "metric": "ORB frames/sec (752x480, 1000 feat, 8 lvl)", "value": round(fps, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "batch of %d synthetic 752x480 frames per GPU, ORB extraction only "
                               "(1000 features, 8 levels, scale 1.2, FAST 20/7), inputs resident in HBM" % B,
                   "frames_per_batch": B, "mean_keypoints_per_frame": round(float(nkp.mean()), 1)},
        "stage_ms_per_step": {k: round(v[0] / max(1, v[1]), 4) for k, v in stages.items()},
        "stage_ms_note": "every stage bracketed, %d extra steps after the timed region; roofline.kernel_ms is from the "
                         "timed region itself" % min(args.steps, 5),
        "roofline": roofline,
    }

    # ---- Hamming matching legs (rank-local; reported per GPU and as whole job) ----
    if not args.no_match:
        m = ORBmatcher(ex)
        g = torch.Generator(device="cpu").manual_seed(99 + rank)
        # (i) 128 stereo pairs x (1000 x 1000)
        nb, Q = 128, 1000
        dq = torch.randint(0, 256, (nb, Q, 32), dtype=torch.uint8, generator=g).to(dev)
        dt = torch.randint(0, 256, (nb, Q, 32), dtype=torch.uint8, generator=g).to(dev)
        outs = [torch.zeros((nb, Q), dtype=torch.int32, device=dev) for _ in range(3)]

        def mstep1():
            m.match_bruteforce_device(dq, Q, Q, None, dt, Q, Q, None, nb, *outs)

        # (ii) 8000 queries x 100000 map descriptors
        Q2, T2 = 8000, 100000
        dq2 = torch.randint(0, 256, (Q2, 32), dtype=torch.uint8, generator=g).to(dev)
        dt2 = torch.randint(0, 256, (T2, 32), dtype=torch.uint8, generator=g).to(dev)
        outs2 = [torch.zeros((Q2,), dtype=torch.int32, device=dev) for _ in range(3)]

        def mstep2():
            m.match_bruteforce_device(dq2, Q2, Q2, None, dt2, T2, T2, None, 1, *outs2)

        res = {}
        for name, fn, pairs in (("stereo_128x1000x1000", mstep1, nb * Q * Q), ("map_8000x100000", mstep2, Q2 * T2)):
            for _ in range(3):
                fn()
            barrier()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            barrier()
            dt_ = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt_], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_ = float(t.item())
            res[name] = {"pair_distances_per_s": round(world * pairs * reps / dt_, 1),
                         "ms_per_call": round(dt_ / reps * 1e3, 4)}
        out["hamming"] = res

    # ---- latency of one frame, as an online tracker calls it (reported beside the throughput; not `value`) ----
    if rank == 0 and args.single_frame:
        ex.set_stream(None)
        one = host[0]
        for _ in range(10):
            ex(one)
        lat = []
        for i in range(100):
            t0 = time.perf_counter()
            ex(host[i % min(B, 8)])
            lat.append(time.perf_counter() - t0)
        out["single_frame"] = {"host_buffers_median_us": round(float(np.median(lat)) * 1e6, 1),
                               "note": "orbx_extract on one 752x480 host image: upload, kernels, download, synchronise"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(host[:min(B, 256)])
    elif rank == 0:
        out["cpu_baseline"] = None

    ex.set_stream(None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
