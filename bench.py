#!/usr/bin/env python3
"""Benchmark of the ORB front-end hot path on MI355X.

Default workload (BASELINE.json configs[1]): one step = one pass of ORB extraction (pyramid, FAST
cells, quadtree, orientation, blur, rBRIEF) over one batch of 256 synthetic 752x480 frames that are
already resident in HBM.  `value` is whole-job frames/s.

`--gpus N`: one process per GPU.  Under torch.distributed.run the ranks already exist (WORLD_SIZE
set); started plainly, this script starts the N ranks itself -- before anything in the parent
touches the GPU -- and relays rank 0's JSON line.  Frames are independent, so there is no data-path
collective: the timed region ends with a barrier and the max over ranks; one result gather
(orb_slam_amd.shard.gather_features) runs after it and its counts are checked.

Other BASELINE configs: --config stereo (128 pairs: extraction + row-band stereo + 1000 x 1000 brute
force), --config kitti (4 x 256 frames 1241x376 / 2000 features, sharded over the ranks by contiguous
frame ranges: strong scaling), --config 4k-map (8 frames 3840x2160 / 8000 features per GPU + brute
force of every frame's descriptors against a 100 000-point map).  The default run reports a short
leg of each of them next to the headline number (`legs`), plus: Hamming brute force alone, natural
images, host-fed batches (page-locked and pageable), the roofline of the dominant kernel and the CPU
baseline (the C restatement of the reference path on this node's host cores, rank 0 at N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
FP4_PEAK_TFLOPS = 10000.0      # dense block-scaled FP4 matrix peak (the brute-force matcher's inner products)
SIMDS, CLOCK_GHZ = 1024, 2.4   # 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles

CONFIGS = {
    # name: frame size, features, frames per batch (per GPU), first seed, scaling
    "euroc": dict(w=752, h=480, nfeat=1000, frames=256, seed0=1234, scaling="weak",
                  what="batch of %d synthetic 752x480 frames per GPU, ORB extraction only"),
    "stereo": dict(w=752, h=480, nfeat=1000, frames=256, seed0=4321, scaling="weak",
                   what="%d synthetic 752x480 frames = stereo pairs per GPU: extraction + row-band stereo search + "
                        "1000x1000 brute force per pair"),
    "kitti": dict(w=1241, h=376, nfeat=2000, frames=256, seed0=777, scaling="strong", total=1024,
                  what="4 sequences x 256 synthetic 1241x376 frames (2000 features) sharded over the GPUs by contiguous "
                       "frame ranges, batches of %d"),
    "4k-map": dict(w=3840, h=2160, nfeat=8000, frames=8, seed0=99, scaling="weak",
                   what="%d synthetic 3840x2160 frames (8000 features) per GPU: extraction + brute force of every frame's "
                        "descriptors against a replicated 100000-point map"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-ms", type=float, default=400.0,
                    help="run the workload untimed for this long before the warm-up steps: the first ~30 ms of full load\n"
                         "after idle run 10-25 %% slower (power management pulls the clock down at the load step and lets\n"
                         "it recover), and a 20-step timed region would otherwise sit inside that transient")
    ap.add_argument("--config", default="euroc", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=0, help="frames per batch (per GPU); 0 = the config's own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline workload (no legs, no host-fed rates)")
    ap.add_argument("--single-frame", action="store_true",
                    help="also time orbx_extract on one host frame (adds small launches of every kernel, which\n"
                         "would dilute per-kernel averages in a profile of the default command)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group at world size 1 as well, so that the RCCL initialisation, the barrier, the\n"
                         "all-reduce and the device-tensor result gather execute on a one-GPU box")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; "
                    "gloo only to rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# N ranks from a plain start
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """Start args.gpus copies of this script, one per GPU, and relay rank 0's JSON line.  The parent
    never initialises the GPU (no torch import, no HIP call)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    # exactly one line on stdout: rank 0's JSON (the gloo backend prints connection notes there)
    lines = [ln for ln in out.decode().splitlines() if ln.startswith('{"metric"')]
    sys.stderr.write("".join(ln + "\n" for ln in out.decode().splitlines() if not ln.startswith('{"metric"')))
    if lines:
        print(lines[-1], flush=True)
    return rc if lines else (rc or 1)


# ------------------------------------------------------------------------------------------------
# algorithmic bytes
# ------------------------------------------------------------------------------------------------
def path_bytes(ex, w, h, nfeat):
    """(A, P): algorithmic bytes per frame of the whole path, A = W*H + 4P - w_last*h_last + 60N with
    P = sum of level pixels (input read, pyramid write, resize source reads, FAST read, one read for
    blur / orientation / descriptors, 28-byte keypoint + 32-byte descriptor out)"""
    sizes = [ex.level_size(w, h, l) for l in range(ex.nlevels)]
    P = sum(a * b for a, b in sizes)
    return w * h + 4 * P - sizes[-1][0] * sizes[-1][1] + 60 * nfeat, P


# ------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------
def host_cores():
    """threads this job may use: the node's cores, capped by the job's CPU share when one is set"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    share = os.environ.get("OMP_NUM_THREADS")
    if share and share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


def cpu_frame_steps(sample):
    """The oracle's single-thread time for the per-frame steps that follow extraction (undistortion, grid
    candidate search, row-band stereo search, classic stereo search with SAD refinement) on one stereo pair
    of the GPU leg: milliseconds per frame / pair."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import KP_DTYPE, Oracle
    from orb_slam_amd.synth import synthetic_stereo_pair
    kp = lambda a: np.ascontiguousarray(a).view(KP_DTYPE).reshape(-1)      # noqa: E731
    kl, kr, dl, dr, q = kp(sample["kl"]), kp(sample["kr"]), sample["dl"], sample["dr"], sample["q"]
    K, D, R, P = sample["cam"]
    mbf, mb = sample["mbf"], sample["mb"]
    ol = Oracle(1000, 1.2, 8, 20, 7, native=True, trig="libm")
    orr = Oracle(1000, 1.2, 8, 20, 7, native=True, trig="libm")
    left, right = synthetic_stereo_pair(752, 480, sample["seed"])
    ol.extract(left)                 # the classic search reads both pyramids
    orr.extract(right)
    sf = ol.tables()["scale"]
    xy = np.stack([kl["x"], kl["y"]], axis=1)

    def grid():
        off, idx = ol.features_in_area(kl, *sample["grid"], q["qx"], q["qy"], q["qr"], q["qmin"], q["qmax"])
        ol.match_windowed(q["qd"], dl, off, idx, kl["octave"])
    steps = {"undistort": lambda: ol.undistort_points(xy, K, D, R, P),
             "grid_search": grid,
             "stereo_rowband": lambda: ol.stereo_frame(kl, dl, kr, dr, sf, 480, mbf, mb, 0.0).ComputeStereoMatches_Undistorted(False),
             "stereo_sad": lambda: ol.stereo_classic(orr, kl, dl, kr, dr, mbf, mb)}
    out = {}
    for name, fn in steps.items():
        fn()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            fn()
            n += 1
        out[name] = round((time.perf_counter() - t0) / n * 1e3, 4)
    return out


def cpu_baseline(frames, nfeat, frame_sample=None):
    """The oracle (literal restatement of the reference's CPU path, built with the reference's
    -O3 -march=native) on a bounded sample of the same frames: one frame per thread on every core
    this job may use, for a fixed wall-clock time, then the same for 1000 x 1000 brute-force matching."""
    import threading
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    ncores = host_cores()
    o = Oracle(nfeat, 1.2, 8, 20, 7, native=True, trig="libm")
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < 2.0:
        o.extract(frames[n1 % len(frames)], literal=True)
        n1 += 1
    single = n1 / (time.perf_counter() - t0)
    oracles = [Oracle(nfeat, 1.2, 8, 20, 7, native=True, trig="libm") for _ in range(ncores)]
    done = [0] * ncores
    budget = 5.0

    def work(k):
        j = 0
        t_end = time.perf_counter() + budget
        while time.perf_counter() < t_end:
            oracles[k].extract(frames[(k * 61 + j) % len(frames)], literal=True)
            j += 1
        done[k] = j

    threads = [threading.Thread(target=work, args=(k,)) for k in range(ncores)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    multi = sum(done) / wall
    # matching: 1000 x 1000 brute force (best / second best), one search per thread at a time
    rng = np.random.default_rng(7)
    q = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
    t_ = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
    mdone = [0] * ncores

    def mwork(k):
        j = 0
        t_end = time.perf_counter() + 2.0
        while time.perf_counter() < t_end:
            oracles[k].match_bruteforce(q, t_)
            j += 1
        mdone[k] = j

    threads = [threading.Thread(target=mwork, args=(k,)) for k in range(ncores)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    mwall = time.perf_counter() - t0
    frame_steps = None
    if frame_sample is not None:
        frame_steps = {"ms_per_frame_single_thread": cpu_frame_steps(frame_sample), "cores": 1,
                       "sample": "the first stereo pair of legs.frame_steps, 0.3 s per step"}
    return {"frame_steps": frame_steps, "value": round(multi, 2), "unit": "frames/s", "cores": ncores, "kind": "port",
            "sample": "%d of the batch's frames, one frame per thread on %d threads for %.1f s (the job's CPU share; the node "
                      "shows %d logical CPUs); single thread: %.2f frames/s over %d frames"
                      % (sum(done), ncores, wall, os.cpu_count() or 0, single, n1),
            "single_thread_value": round(single, 2), "frames_per_s_per_core": round(multi / ncores, 2),
            "node_logical_cpus": os.cpu_count(),
            "matching": {"pair_distances_per_s": round(sum(mdone) * 1e6 / mwall, 1), "cores": ncores,
                         "sample": "%d brute-force searches 1000 x 1000 (SWAR popcount, best / second best) in %.1f s"
                                   % (sum(mdone), mwall)}}


# ------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    # Exactly one line goes to stdout: the JSON.  Libraries write banners there (RCCL prints its version block at
    # communicator creation), so file descriptor 1 points at stderr for the whole run and the JSON line is written
    # to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from orb_slam_amd import ORBextractor, ORBmatcher, KP_DTYPE
    from orb_slam_amd.shard import frame_range, gather_features
    from orb_slam_amd.synth import synthetic_frame, synthetic_stereo_pair, random_descriptors, map_descriptors

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, ndev)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:            # a plain one-rank start
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            s_.close()
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the timing reduction lives
    # a dedicated (non-default) torch stream carries every kernel of the library, so that torch
    # synchronisation and the library's own HIP events see the same queue
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist_on:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    class Workload:
        """one config on this rank: resident frames, output buffers, step()"""

        def __init__(self, name, frames=0, total=None):
            c = CONFIGS[name]
            self.name, self.c = name, c
            self.w, self.h, self.nfeat = c["w"], c["h"], c["nfeat"]
            self.B = frames or c["frames"]
            self.ex = ORBextractor(self.nfeat, 1.2, 8, 20, 7, device=dev_index)
            self.ex.set_stream(stream.cuda_stream)
            self.m = ORBmatcher(self.ex)
            self.cap = self.ex.max_keypoints(self.w, self.h)
            seed0 = c["seed0"]
            if name == "kitti":
                # the whole job is `total` frames; this rank's shard runs in batches of at most B
                self.total = total or c["total"]
                b, e = frame_range(rank, world, self.total)
                self.shard = (b, e)
                host = np.stack([synthetic_frame(self.w, self.h, seed0 + i) for i in range(b, e)])
                self.frames_per_step = e - b
                self.global_frames_per_step = self.total
            elif name == "stereo":
                assert self.B % 2 == 0
                pairs = [synthetic_stereo_pair(self.w, self.h, seed0 + rank * 100000 + i) for i in range(self.B // 2)]
                host = np.stack([p[0] for p in pairs] + [p[1] for p in pairs])      # all left frames, then all right ones
                self.frames_per_step = self.B
                self.global_frames_per_step = self.B * world
            else:
                kw = dict(nrect=1500) if name == "4k-map" else {}
                host = np.stack([synthetic_frame(self.w, self.h, seed0 + rank * 100000 + i, **kw) for i in range(self.B)])
                self.frames_per_step = self.B
                self.global_frames_per_step = self.B * world
            self.host = host
            F = len(host)
            self.d_imgs = torch.from_numpy(host).to(dev)
            self.d_kps = torch.zeros((F, self.cap, 7), dtype=torch.float32, device=dev)
            self.d_desc = torch.zeros((F, self.cap, 32), dtype=torch.uint8, device=dev)
            self.d_n = torch.zeros((F,), dtype=torch.int32, device=dev)
            self.pairs_per_step = 0
            if name == "stereo":
                P_ = F // 2
                self.st = [torch.zeros((P_, self.cap), dtype=torch.int32, device=dev) for _ in range(2)] + \
                          [torch.zeros((P_, self.cap), dtype=torch.float32, device=dev) for _ in range(2)] + \
                          [torch.zeros((P_,), dtype=torch.int32, device=dev)]
                self.bf = [torch.zeros((P_, self.cap), dtype=torch.int32, device=dev) for _ in range(3)]
                self.pairs_per_step = P_
            if name == "4k-map":
                q = random_descriptors(8000, 5)
                self.d_map = torch.from_numpy(map_descriptors(100000, q, seed=99)).to(dev)
                self.bf = [torch.zeros((F, self.cap), dtype=torch.int32, device=dev) for _ in range(3)]

        def extract(self, f0, n):
            W_, H_ = self.w, self.h
            self.ex.extract_batch_device(self.d_imgs[f0:], n, W_, H_, W_, W_ * H_, self.d_kps[f0:], self.d_desc[f0:],
                                         self.cap, self.d_n[f0:])

        def step(self):
            F = len(self.host)
            if self.name == "kitti":
                for f0 in range(0, F, self.B):
                    self.extract(f0, min(self.B, F - f0))
                return
            self.extract(0, F)
            if self.name == "stereo":
                P_ = F // 2
                cap = self.cap
                self.m.stereo_match_device(self.d_kps, self.d_desc, self.d_n, self.d_kps[P_:], self.d_desc[P_:], self.d_n[P_:],
                                           P_, cap, self.h, 47.9, 0.11, 0.0, *self.st)
                self.m.match_bruteforce_device(self.d_desc, cap, cap, self.d_n, self.d_desc[P_:], cap, cap, self.d_n[P_:],
                                               P_, *self.bf)
            elif self.name == "4k-map":
                cap = self.cap
                self.m.match_bruteforce_device(self.d_desc, cap, cap, self.d_n, self.d_map, 100000, 0, None, F, *self.bf)

        def close(self):
            self.ex.set_stream(None)
            self.ex.close()

    def settle(fn, ms):
        """untimed calls of fn for `ms` of wall clock (queue kept short); returns how many"""
        n, t0 = 0, time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            fn()
            n += 1
            if n % 8 == 0:
                torch.cuda.synchronize()
        return n

    def time_steps(wl, steps, warmup):
        settle(wl.step, args.settle_ms / 2)
        for _ in range(warmup):
            wl.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.step()
        barrier()
        return time.perf_counter() - t0

    # ---------------------------------------------------------------------------------------------
    # headline workload
    # ---------------------------------------------------------------------------------------------
    wl = Workload(args.config, args.frames)
    ex, B, W, H = wl.ex, wl.B, wl.w, wl.h
    A_bytes, P_pixels = path_bytes(ex, W, H, wl.nfeat)

    # The same W + K steps straight from an idle GPU first (what a run without the settle phase measures: the first
    # tens of milliseconds of full load after idle run at a lower clock), reported as value_from_idle.
    wl.step()                                   # first call: buffers are sized here
    barrier()
    time.sleep(0.5)
    for _ in range(args.warmup):
        wl.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    barrier()
    idle_elapsed = max_over_ranks(time.perf_counter() - t0)

    settle_steps = settle(wl.step, args.settle_ms)
    for _ in range(args.warmup):
        wl.step()
    barrier()
    # timed region: HIP events only around the launch that dominates (FAST + blur); every stage event
    # costs a little, and the roofline needs just this one duration
    ex.set_stage_timing(2)
    ex.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    barrier()
    my_elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(my_elapsed)
    dom_timed = ex.stage_times()["fast_cells"]
    # per-stage breakdown: a few more steps outside the timed region with every stage bracketed
    ex.set_stage_timing(1)
    ex.reset_stage_times()
    for _ in range(min(args.steps, 5)):
        wl.step()
    barrier()
    stages = ex.stage_times()
    ex.set_stage_timing(0)
    nkp = wl.d_n.cpu().numpy()
    fps = wl.global_frames_per_step * args.steps / elapsed
    my_fps = wl.frames_per_step * args.steps / my_elapsed

    # the blur's work items ride in the FAST launch (k_fast_blur): one stage, the bytes of both
    per = lambda k: stages[k][0] / max(1, stages[k][1])
    if per("blur") < max(0.05 * per("fast_cells"), 0.008):      # an empty bracket still costs its two event records
        stages["fast_blur"] = stages.pop("fast_cells")
        stages.pop("blur")
    launches_per_step = max(1, dom_timed[1] // max(1, args.steps))          # kitti: one launch per batch
    frames_per_launch = wl.frames_per_step / launches_per_step
    # dominant kernel and its roofline (HIP events on the launch stream, inside the timed region)
    longest = max(stages, key=lambda k: stages[k][0])
    # the quadtree has no term in the path formula (it moves a few bytes per candidate and is bound by its barrier chain):
    # when it is the longest stage (a handful of very large frames), the roofline prices the longest stage that has bytes
    dom = max((k for k in stages if k != "quadtree"), key=lambda k: stages[k][0])
    dom_ms = stages[dom][0] / max(1, stages[dom][1])
    if dom in ("fast_blur", "fast_cells") and dom_timed[1] > 0:
        dom_ms = dom_timed[0] / dom_timed[1]
    last = ex.level_size(W, H, ex.nlevels - 1)
    stage_bytes = {  # algorithmic bytes per frame of each stage: its terms of the path formula
        "pyr_level0": 2 * W * H,
        "pyr_resize": 2 * P_pixels - W * H - last[0] * last[1],
        "blur": P_pixels,                      # the path's single read for blur / orientation / descriptors
        "fast_cells": P_pixels,                # FAST read
        "fast_blur": 2 * P_pixels,             # FAST read + blur read
        "quadtree": 0,
        "orient_desc": 60 * wl.nfeat,
    }
    achieved = stage_bytes[dom] * frames_per_launch / (dom_ms * 1e-3) / 1e9
    prof = {}
    for name in ("traffic", "issue"):
        try:
            prof[name] = json.load(open(os.path.join(ROOT, "profiles", name + ".json")))
        except Exception:
            prof[name] = {}
    comparable = args.config == "euroc" and B == 256      # what the committed counter passes measured
    traffic = prof["traffic"].get(dom) if comparable else None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same 256-frame "
                                  "launch, (2*FETCH_SIZE + WRITE_SIZE)*1024; not measured in this run)" if traffic else None,
                "kernel_ms": round(dom_ms, 4), "longest_stage": longest,
                "algorithmic_bytes_per_launch": int(stage_bytes[dom] * frames_per_launch),
                "bytes_note": ("FAST read P + blur read P per frame (P = %d level pixels); with the blurred-plane write, which the "
                               "path formula does not count: %.2f GB/s" % (P_pixels, 1.5 * achieved)) if dom == "fast_blur" else
                              "this stage's terms of the path formula (P = %d level pixels)" % P_pixels,
                "path_achieved": round(fps / world * A_bytes / 1e9, 2),
                "path_frac": round(fps / world * A_bytes / 1e9 / HBM_PEAK_GBS, 5),
                "path_bytes_per_frame": A_bytes}
    valu = prof["issue"].get(dom) if comparable else None
    if valu:
        # the limit that binds this kernel: vector instruction issue (every wave64 VALU instruction holds its SIMD for 4 cycles)
        roofline["limited_by"] = "valu_issue"
        roofline["valu_issue_frac"] = round(valu["SQ_INSTS_VALU"] * 4 / (SIMDS * dom_ms * 1e-3 * CLOCK_GHZ * 1e9), 4)
        roofline["valu_issue_source"] = "SQ_INSTS_VALU per launch from profiles/issue.json (rocprofv3 --pmc pass) x 4 cycles / " \
                                        "(1024 SIMDs x kernel time of this run x 2.4 GHz)"

    out = {
        "metric": "ORB frames/sec (752x480, 1000 feat, 8 lvl)" if args.config in ("euroc", "stereo") else
                  "ORB frames/sec (%dx%d, %d feat, 8 lvl)" % (W, H, wl.nfeat),
        "value": round(fps, 1), "unit": "frames/s",
        "value_from_idle": round(wl.global_frames_per_step * args.steps / idle_elapsed, 1),
        "value_from_idle_note": "the same %d warm-up + %d timed steps started 0.5 s after the GPU went idle, before the settle "
                                "phase: `value` is the rate under sustained load, this one includes the clock transient after "
                                "idle" % (args.warmup, args.steps),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "settle": {"ms": args.settle_ms, "steps": settle_steps,
                   "note": "untimed steps of the same workload before the warm-up, so that the timed steps run at the clock "
                           "the GPU holds under sustained load, not in the transient after idle"},
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": wl.c["scaling"],
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": (wl.c["what"] % B) + " (features %d, 8 levels, scale 1.2, FAST 20/7), inputs resident in HBM"
                               % wl.nfeat,
                   "name": args.config, "frames_per_batch": B, "frames_per_step_whole_job": wl.global_frames_per_step,
                   "mean_keypoints_per_frame": round(float(nkp.mean()), 1)},
        "stage_ms_per_launch": {k: round(v[0] / max(1, v[1]), 4) for k, v in stages.items()},
        "stage_ms_note": "every stage bracketed, %d extra steps after the timed region; roofline.kernel_ms is from the "
                         "timed region itself" % min(args.steps, 5),
        "roofline": roofline,
    }

    # ---- multi-rank facts: the world that was really initialised, per-rank rates, the result gather ----
    if dist_on:
        rates = [torch.zeros(1, dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(rates, torch.tensor([my_fps], dtype=torch.float64, device=red_dev))
        Fl = len(wl.host)
        total = wl.total if args.config == "kitti" else Fl * world
        g_in = (wl.d_n, wl.d_kps, wl.d_desc) if args.backend == "nccl" else (wl.d_n.cpu(), wl.d_kps.cpu(), wl.d_desc.cpu())
        barrier()
        t0 = time.perf_counter()
        n_all, k_all, d_all = gather_features(*g_in, total)
        barrier()
        g_ms = (time.perf_counter() - t0) * 1e3
        b0 = frame_range(rank, world, total)[0]
        ok = bool((n_all[b0:b0 + Fl].cpu() == wl.d_n.cpu()).all()) and n_all.shape[0] == total and \
            bool((d_all[b0].cpu() == wl.d_desc[0].cpu()).all())
        okt = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        out["distributed"] = {"backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""),
                              "world_size_initialised": dist.get_world_size(), "devices_visible": ndev,
                              "per_rank_frames_per_s": [round(float(r.item()), 1) for r in rates],
                              "result_gather": {"frames": int(total), "ms": round(g_ms, 3),
                                                "bytes_per_rank": int(Fl * wl.cap * 60 + Fl * 4),
                                                "counts_and_payload_check": "ok" if okt.item() == 1.0 else "MISMATCH"}}
        assert okt.item() == 1.0, "gathered results differ from the local ones"

    # ---- cheap invariant on the benchmarked shape: first and last frame equal a single-frame call ----
    if rank == 0:
        kb = wl.d_kps.cpu().numpy()
        db = wl.d_desc.cpu().numpy()
        ex.set_stream(None)
        for i in (0, len(wl.host) - 1):
            k1, d1 = ex(wl.host[i])
            n_i = int(nkp[i])
            assert n_i == len(k1), "frame %d: batch gave %d keypoints, a single-frame call %d" % (i, n_i, len(k1))
            assert np.array_equal(kb[i, :n_i].view(np.uint8), k1.view(np.uint8).reshape(n_i, 28)), "keypoints of frame %d" % i
            assert d1 is None or np.array_equal(db[i, :n_i], d1), "descriptors of frame %d" % i
        ex.set_stream(stream.cuda_stream)
        out["invariant"] = "frames 0 and %d of the timed batch are byte-identical to single-frame calls" % (len(wl.host) - 1)

    # ---- the north-star descriptor criterion on the benchmarked frames: this run's GPU output against the CPU
    #      restatement with the reference's trig formulation (this host's libm cosf / sinf), part of the CPU leg ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import Oracle
        o_ = Oracle(wl.nfeat, 1.2, 8, 20, 7, native=False, trig="libm")
        nfr = min(len(wl.host), 32 if args.config != "4k-map" else 2)
        tot = dif = 0
        worst = 0.0
        same_kp = True
        for i in range(nfr):
            ko, do = o_.extract(wl.host[i])
            n_i = int(nkp[i])
            kg = kb[i, :n_i].view(np.uint8).reshape(n_i, 28).copy().view(KP_DTYPE).reshape(n_i)
            same_kp = same_kp and n_i == len(ko) and all(np.array_equal(kg[f_], ko[f_]) for f_ in
                                                         ("x", "y", "size", "response", "octave", "class_id"))
            if n_i == len(ko):
                worst = max(worst, float(np.abs(kg["angle"] - ko["angle"]).max(initial=0.0)))
                dif += int((db[i, :n_i] != do).any(axis=1).sum())
                tot += n_i
        out["parity_vs_libm_trig"] = {"frames": nfr, "keypoints": tot, "keypoints_bit_identical": bool(same_kp),
                                      "max_angle_diff_deg": worst, "descriptors_identical_frac": round(1.0 - dif / max(1, tot), 6),
                                      "what": "GPU output of the timed batch's first frames against the CPU restatement in its "
                                              "libm mode (cosf / sinf of this host, as src/ORBextractor.cc:108-109 calls them); "
                                              "criterion: keypoints identical, angles within 1e-3 deg, >= 0.999 of descriptors identical"}

    extras = not args.no_extras and world == 1 and args.config == "euroc"
    frame_sample = None
    # ---- Hamming matching alone: every rank searches its own queries against its replica of the train set;
    #      the whole-job rate is the sum over the ranks ----
    if not args.no_extras and args.config == "euroc":
        m = wl.m
        g = torch.Generator(device="cpu").manual_seed(99 + rank)
        nb, Q = 128, 1000
        dq = torch.randint(0, 256, (nb, Q, 32), dtype=torch.uint8, generator=g).to(dev)
        dt = torch.randint(0, 256, (nb, Q, 32), dtype=torch.uint8, generator=g).to(dev)
        outs = [torch.zeros((nb, Q), dtype=torch.int32, device=dev) for _ in range(3)]
        Q2, T2 = 8000, 100000
        dq2 = torch.randint(0, 256, (Q2, 32), dtype=torch.uint8, generator=g).to(dev)
        dt2 = torch.randint(0, 256, (T2, 32), dtype=torch.uint8, generator=g).to(dev)
        outs2 = [torch.zeros((Q2,), dtype=torch.int32, device=dev) for _ in range(3)]
        nb8 = 8
        dq8 = torch.randint(0, 256, (nb8, Q2, 32), dtype=torch.uint8, generator=g).to(dev)
        outs8 = [torch.zeros((nb8, Q2), dtype=torch.int32, device=dev) for _ in range(3)]
        try:
            mprof = json.load(open(os.path.join(ROOT, "profiles", "match_issue.json")))
        except Exception:
            mprof = {}
        res = {}
        for name, fn, pairs, nq, nt in (
                ("stereo_128x1000x1000", lambda: m.match_bruteforce_device(dq, Q, Q, None, dt, Q, Q, None, nb, *outs), nb * Q * Q, nb * Q, nb * Q),
                ("map_8000x100000", lambda: m.match_bruteforce_device(dq2, Q2, Q2, None, dt2, T2, T2, None, 1, *outs2), Q2 * T2, Q2, T2),
                # one ordinary frame's features against the same map (the first 1000 queries of the set above)
                ("frame_1000x100000", lambda: m.match_bruteforce_device(dq2, Q, Q, None, dt2, T2, T2, None, 1, *outs2), Q * T2, Q, T2),
                # the matching half of config 4k-map: 8 frames of 8000 queries against one shared 100 000-point map
                ("map_8x8000x100000", lambda: m.match_bruteforce_device(dq8, Q2, Q2, None, dt2, T2, 0, None, nb8, *outs8),
                 nb8 * Q2 * T2, nb8 * Q2, T2)):
            settle(fn, args.settle_ms / 2)
            for _ in range(3):
                fn()
            barrier()
            reps = 20
            # HIP events on the stream the kernels run on (torch's current stream is the library's stream here)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            barrier()
            dt_ = time.perf_counter() - t0
            dt_ = max_over_ranks(dt_)
            # the kernel time on its own, after the timed calls (an event pair per call costs a few microseconds of its own)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for e0, e1 in evs:
                e0.record()
                fn()
                e1.record()
            barrier()
            call_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / reps
            tflops = pairs * 512 / (call_ms * 1e-3) / 1e12          # 256 multiply-adds per pair of 256-bit descriptors
            formal = 32 * (nq + nt) + 12 * nq
            roof = {"bound": "mfma", "kernel": "k_match_bruteforce_fp4 (+ k_fp4_expand_trains where many query tiles share a train set, k_match_merge)", "achieved": round(tflops, 1),
                    "peak": FP4_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / FP4_PEAK_TFLOPS, 4), "traffic": None,
                    "kernel_ms": round(call_ms, 4),
                    "flops_note": "pair distances x 512 (256-bit inner product as block-scaled FP4 matrix instructions) against the "
                                  "dense FP4 peak; kernel_ms = HIP events around the call's launches on their stream",
                    "hbm_formal": {"bytes_per_call": formal, "GB/s": round(formal / (call_ms * 1e-3) / 1e9, 2),
                                   "frac_of_hbm_peak": round(formal / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                   "note": "32 (Q + T) + 12 Q bytes: every descriptor read once, three results per query"}}
            pm = mprof.get(name)
            if pm:
                roof["valu_issue_frac"] = round(pm["SQ_INSTS_VALU"] * 4 / (SIMDS * call_ms * 1e-3 * CLOCK_GHZ * 1e9), 4)
                roof["mfma_busy_frac"] = round(pm["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * call_ms * 1e-3 * CLOCK_GHZ * 1e9), 4) \
                    if "SQ_VALU_MFMA_BUSY_CYCLES" in pm else None
                if pm.get("SQ_LDS_IDX_ACTIVE"):
                    roof["lds_conflict_frac"] = round(pm["SQ_LDS_BANK_CONFLICT"] / pm["SQ_LDS_IDX_ACTIVE"], 4)
                roof["counters_source"] = "profiles/match_issue.json (rocprofv3 --pmc passes of tools/match_one_shape.py, the same " \
                                          "shape; per-launch counters, not measured in this run) over this run's kernel time"
            res[name] = {"pair_distances_per_s": round(world * pairs * reps / dt_, 1),
                         "queries_per_s": round(world * nq * reps / dt_, 1), "ms_per_call": round(dt_ / reps * 1e3, 4),
                         "roofline": roof}
            if world > 1:
                res[name]["note"] = "whole job: %d ranks, each its own %d queries against a replicated train set" % (world, nq)
        out["hamming"] = res

    # ---- short legs of the other BASELINE configs, natural images, host-fed batches ----
    if extras:
        legs = {}
        for name in ("stereo", "kitti", "4k-map"):
            w2 = Workload(name, total=256 if name == "kitti" else None)
            steps2 = 5
            dt_ = time_steps(w2, steps2, 2)
            A2, _ = path_bytes(w2.ex, w2.w, w2.h, w2.nfeat)
            f2 = w2.frames_per_step * steps2 / dt_
            leg = {"frames_per_s": round(f2, 1), "ms_per_step": round(dt_ / steps2 * 1e3, 4),
                   "frames_per_step": w2.frames_per_step, "mean_keypoints_per_frame": round(float(w2.d_n.float().mean().item()), 1),
                   "path_frac_of_hbm": round(f2 * A2 / 1e9 / HBM_PEAK_GBS, 5), "path_bytes_per_frame": A2}
            if name == "stereo":
                leg["pairs_per_s"] = round(w2.pairs_per_step * steps2 / dt_, 1)
                leg["mean_stereo_matches_per_pair"] = round(float((w2.st[2] >= 0).sum().item()) / w2.pairs_per_step, 1)
            if name == "4k-map":
                leg["pair_distances_per_s_incl_extraction"] = round(float(w2.d_n.sum().item()) * 100000 * steps2 / dt_, 1)
            if name == "kitti":
                leg["note"] = "one sequence of 256 frames; --config kitti runs the 4 x 256 sharded job"
            legs[name] = leg
            w2.close()
            del w2
            torch.cuda.empty_cache()
        # the per-frame steps that follow extraction in ORB_SLAM2::Frame, on 128 resident stereo pairs
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from frame_bench import measure as measure_frame_steps
        fs_ms, fs_facts, frame_sample = measure_frame_steps(128, 20, dev, stream)
        legs["frame_steps"] = {
            "pairs": 128, "ms_per_call": {k: round(v, 4) for k, v in fs_ms.items()},
            "frames_per_s": {k: round(128 / v * 1e3, 1) for k, v in fs_ms.items()},
            "what": "orbx_undistort_keypoints_device (5 coefficients + rectification), orbx_grid_search_device (one query per "
                    "left keypoint, radius 15 x its octave's scale), orbx_stereo_match_device, orbx_stereo_match_sad_device on "
                    "the extraction output of 128 synthetic stereo pairs, resident",
            **{k: round(v, 1) for k, v in fs_facts.items()}}
        # natural images: the reference's own EuRoC pair, 128 copies of each
        try:
            l_ = np.load(os.path.join(ROOT, "tests", "fixtures", "EuRoC_l.npy"))
            r_ = np.load(os.path.join(ROOT, "tests", "fixtures", "EuRoC_r.npy"))
            nat = np.stack([l_, r_] * ((B + 1) // 2))[:B]
            d_nat = torch.from_numpy(nat).to(dev)
            keep = wl.d_imgs
            wl.d_imgs = d_nat
            dt_ = time_steps(wl, 10, 2)
            legs["natural_images"] = {"frames_per_s": round(B * 10 / dt_, 1), "ms_per_step": round(dt_ / 10 * 1e3, 4),
                                      "what": "%d frames = copies of the reference's EuRoC_l / EuRoC_r test images, resident" % B}
            wl.d_imgs = keep
            del d_nat
        except FileNotFoundError:
            pass
        # host-fed batches: upload + kernels + download + synchronise per call (never `value`)
        import ctypes as C
        ex.set_stream(None)
        for label, pinned in (("pinned", True), ("pageable", False)):
            imgs = np.ascontiguousarray(wl.host)
            kps_h = np.zeros((B, wl.cap), KP_DTYPE)
            desc_h = np.zeros((B, wl.cap, 32), np.uint8)
            n_h = np.zeros(B, np.int32)
            keepalive = []
            if pinned:
                def pin(a):
                    t = torch.empty(a.nbytes, dtype=torch.uint8).pin_memory()
                    keepalive.append(t)
                    v = t.numpy().view(a.dtype).reshape(a.shape)
                    v[...] = a
                    return v
                imgs, kps_h, desc_h = pin(imgs), pin(kps_h), pin(desc_h)
            ptrs = (C.c_void_p * B)(*[imgs[i].ctypes.data for i in range(B)])

            def call():
                rc = ex._lib.orbx_extract_batch(ex._ctx, ptrs, B, W, H, W, kps_h.ctypes.data, desc_h.ctypes.data, wl.cap,
                                                n_h.ctypes.data)
                assert rc == 0, rc
            for _ in range(2):
                call()
            t0 = time.perf_counter()
            for _ in range(5):
                call()
            dt_ = (time.perf_counter() - t0) / 5
            legs["with_transfer_" + label] = {"frames_per_s": round(B / dt_, 1), "ms_per_batch": round(dt_ * 1e3, 3),
                                              "what": "orbx_extract_batch on %s host buffers: upload, kernels, download, "
                                                      "synchronise" % ("page-locked" if pinned else "pageable")}
        # the searches a tracker runs once per frame, as whole-routine calls (host arrays in and out)
        from search_leg import measure as measure_search_calls
        legs["search_calls"] = measure_search_calls(ex, wl.host[0])
        ex.set_stream(stream.cuda_stream)
        out["legs"] = legs

    # ---- latency of one frame, as an online tracker calls it (reported beside the throughput; not `value`) ----
    if rank == 0 and args.single_frame:
        ex.set_stream(None)
        for _ in range(10):
            ex(wl.host[0])
        lat = []
        for i in range(100):
            t0 = time.perf_counter()
            ex(wl.host[i % min(B, 8)])
            lat.append(time.perf_counter() - t0)
        out["single_frame"] = {"host_buffers_median_us": round(float(np.median(lat)) * 1e6, 1),
                               "note": "orbx_extract on one host image: upload, kernels, download, synchronise"}
        ex.set_stream(stream.cuda_stream)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl.host[:min(len(wl.host), 256)], wl.nfeat, frame_sample)
    elif rank == 0:
        out["cpu_baseline"] = None

    wl.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
