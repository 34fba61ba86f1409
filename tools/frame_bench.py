#!/usr/bin/env python3
"""Times of the per-frame steps that follow extraction in ORB_SLAM2::Frame, batched over stereo pairs
resident on the device: keypoint undistortion (src/Frame.cc:745-852), grid candidate search
(:669-722 fused with the scan of src/ORBmatcher.cc:186-225), row-band stereo search with disparity
(:1262-1422) and the classic stereo search with SAD refinement (:984-1173).
Usage: python tools/frame_bench.py [pairs] [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orb_slam_amd import ORBextractor, ORBmatcher  # noqa: E402
from orb_slam_amd.frame import CameraModel, Frame  # noqa: E402
from orb_slam_amd.synth import synthetic_stereo_pair  # noqa: E402

W, H = 752, 480
MBF, MB = 47.9, 0.11
K = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]], np.float64)
D5 = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0], np.float64)
R = np.array([[0.999966347530033, -0.001422739138722922, 0.008079580483432283],
              [0.001365741834644127, 0.9999741760894847, 0.007055629199258132],
              [-0.008089410156878961, -0.007044357138835809, 0.9999424675829176]], np.float64)
P = np.array([[435.2046959714599, 0, 367.4517211914062, 0], [0, 435.2046959714599, 252.2008514404297, 0],
              [0, 0, 1, 0]], np.float64)


def frame_workload(pairs, dev, stream, seed0=4321):
    """extraction of `pairs` stereo pairs on two contexts and the query set of a tracking-style grid search"""
    lr = [synthetic_stereo_pair(W, H, seed0 + i) for i in range(pairs)]
    d_l = torch.from_numpy(np.stack([p[0] for p in lr])).to(dev)
    d_r = torch.from_numpy(np.stack([p[1] for p in lr])).to(dev)
    exl, exr = ORBextractor(1000, 1.2, 8, 20, 7), ORBextractor(1000, 1.2, 8, 20, 7)
    for e in (exl, exr):
        e.set_stream(stream.cuda_stream)
    cap = exl.max_keypoints(W, H)
    out = {}
    for side, ex, d in (("l", exl, d_l), ("r", exr, d_r)):
        k = torch.zeros((pairs, cap, 7), dtype=torch.float32, device=dev)
        de = torch.zeros((pairs, cap, 32), dtype=torch.uint8, device=dev)
        n = torch.zeros((pairs,), dtype=torch.int32, device=dev)
        ex.extract_batch_device(d, pairs, W, H, W, W * H, k, de, cap, n)
        out[side] = (k, de, n)
    torch.cuda.synchronize()
    # queries: the left keypoints seen again a few pixels away (a projected map point), search radius
    # 15 x the scale of its octave and octaves within one of its own, descriptor with a few bits flipped
    kl = out["l"][0].cpu().numpy()
    dl = out["l"][1].cpu().numpy()
    rng = np.random.default_rng(7)
    scale = exl.GetScaleFactors()
    octave = kl[:, :, 5].view(np.int32)
    qx = (kl[:, :, 0] + rng.normal(0, 4, kl.shape[:2])).astype(np.float32)
    qy = (kl[:, :, 1] + rng.normal(0, 4, kl.shape[:2])).astype(np.float32)
    qr = (15.0 * np.asarray(scale, np.float32)[np.clip(octave, 0, 7)]).astype(np.float32)
    qmin = (octave - 1).astype(np.int32)
    qmax = (octave + 1).astype(np.int32)
    qd = dl.copy()
    flip = rng.integers(0, 256, kl.shape[:2] + (6,))
    for j in range(6):
        np.bitwise_xor.at(qd, (np.arange(pairs)[:, None], np.arange(cap)[None, :], flip[:, :, j] // 8),
                          (1 << (flip[:, :, j] % 8)).astype(np.uint8))
    q = dict(qx=qx, qy=qy, qr=qr, qmin=qmin, qmax=qmax, qd=qd)
    d_q = {k_: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k_, v in q.items()}
    return exl, exr, cap, out, q, d_q


def measure(pairs, reps, dev, stream):
    """{step: ms per call over `pairs` stereo pairs} for the steps after extraction, the keypoint total, and a
    host copy of the first pair's data (what a CPU timing of the same steps needs)"""
    exl, exr, cap, out, q, d_q = frame_workload(pairs, dev, stream)
    (kl, dl, nl), (kr, dr, nr) = out["l"], out["r"]
    m = ORBmatcher(exl)
    fr = Frame(exl)
    cam = CameraModel(K, D5, R, P)
    i32 = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)      # noqa: E731
    f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)    # noqa: E731
    ku = torch.zeros_like(kl)
    g_out = [i32(pairs, cap) for _ in range(6)]
    s_out = [i32(pairs, cap), i32(pairs, cap), f32(pairs, cap), f32(pairs, cap), i32(pairs)]
    c_out = [i32(pairs, cap), i32(pairs, cap), f32(pairs, cap), f32(pairs, cap), i32(pairs)]
    inv_w, inv_h = float(np.float32(64) / np.float32(W)), float(np.float32(48) / np.float32(H))
    steps = {
        "undistort": lambda: fr.undistort_keypoints_device(kl, nl, 0, cap, pairs, cam, ku),
        "grid_search": lambda: m.grid_search_device(kl, dl, nl, 0, cap, pairs, 0.0, 0.0, inv_w, inv_h, d_q["qx"], d_q["qy"],
                                                    d_q["qr"], d_q["qmin"], d_q["qmax"], d_q["qd"], nl, cap, cap, *g_out),
        "stereo_rowband": lambda: m.stereo_match_device(kl, dl, nl, kr, dr, nr, pairs, cap, H, MBF, MB, 0.0, *s_out),
        "stereo_sad": lambda: m.stereo_match_sad_device(exr, kl, dl, nl, kr, dr, nr, pairs, cap, MBF, MB, *c_out),
    }
    ms = {}
    for name, fn in steps.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            fn()
        t1.record()
        torch.cuda.synchronize()
        ms[name] = t0.elapsed_time(t1) / reps
    nkp = int(nl.sum().item())
    n0, m0 = int(nl[0].item()), int(nr[0].item())
    facts = {"left_keypoints": nkp, "grid_candidates_per_query": float(g_out[5].sum().item()) / max(1, nkp),
             "stereo_rowband_matches_per_pair": float((s_out[2] >= 0).sum().item()) / pairs,
             "stereo_sad_matches_per_pair": float(c_out[4].float().mean().item())}
    as_kp = lambda t, n: t[0, :n].cpu().numpy().view(np.uint8).reshape(n, 28)       # noqa: E731
    sample = {"kl": as_kp(kl, n0), "dl": dl[0, :n0].cpu().numpy(), "kr": as_kp(kr, m0), "dr": dr[0, :m0].cpu().numpy(),
              "q": {k_: v[0, :n0] for k_, v in q.items()}, "seed": 4321, "cam": (K, D5, R, P), "mbf": MBF, "mb": MB,
              "grid": (0.0, 0.0, inv_w, inv_h)}
    for e in (exl, exr):
        e.set_stream(None)
        e.close()
    return ms, facts, sample


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ms, facts, _ = measure(pairs, reps, dev, stream)
    nkp = facts["left_keypoints"]
    print("%d pairs, %d left keypoints (%.0f per frame)" % (pairs, nkp, nkp / pairs))
    for name, v in ms.items():
        print("%-15s %8.4f ms per %d pairs  %10.0f frames/s  %6.2f ns per keypoint" % (name, v, pairs, pairs / v * 1e3, v * 1e6 / nkp))
    print("grid candidates per query: %.1f   stereo matches per pair: row-band %.0f, SAD %.0f" % (
        facts["grid_candidates_per_query"], facts["stereo_rowband_matches_per_pair"], facts["stereo_sad_matches_per_pair"]))


if __name__ == "__main__":
    main()
