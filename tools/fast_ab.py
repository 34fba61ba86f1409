#!/usr/bin/env python3
"""Stage times of the resident 256-frame batch on synthetic and on natural images, with a hash of
all results (keypoint counts, keypoints, descriptors), so that two builds of liborbx.so can be
compared for speed and for bit-identical output in one run.  Usage: python tools/fast_ab.py [steps]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orb_slam_amd import ORBextractor  # noqa: E402
from orb_slam_amd.synth import synthetic_frame  # noqa: E402


def run(name, host, steps):
    B, H, W = host.shape
    dev = torch.device("cuda:0")
    d_imgs = torch.from_numpy(host).to(dev)
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ex.set_stream(stream.cuda_stream)
    import time
    t_end = time.perf_counter() + 0.4        # past the clock transient that follows idle
    while time.perf_counter() < t_end:
        for _ in range(8):
            ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)
        torch.cuda.synchronize()
    ex.set_stage_timing(1)
    ex.reset_stage_times()
    for _ in range(steps):
        ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)
    torch.cuda.synchronize()
    st = ex.stage_times()
    ex.set_stage_timing(0)
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps):
        ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / steps
    n = d_n.cpu().numpy()
    k = d_kps.cpu().numpy()
    d = d_desc.cpu().numpy()
    h = hashlib.md5(n.tobytes())
    for i in range(B):
        h.update(k[i, :n[i]].tobytes())
        h.update(d[i, :n[i]].tobytes())
    print("%-9s %.4f ms/step %.0f frames/s | %s | md5 %s" % (
        name, ms, B / ms * 1e3, "  ".join("%s %.4f" % (a, v[0] / max(1, v[1])) for a, v in st.items() if v[1]), h.hexdigest()[:12]))
    ex.set_stream(None)
    ex.close()


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    syn = np.stack([synthetic_frame(752, 480, 1234 + i) for i in range(64)])
    run("synthetic", np.concatenate([syn] * 4), steps)
    l_ = np.load(os.path.join(ROOT, "tests", "fixtures", "EuRoC_l.npy"))
    r_ = np.load(os.path.join(ROOT, "tests", "fixtures", "EuRoC_r.npy"))
    run("natural", np.stack([l_, r_] * 128), steps)


if __name__ == "__main__":
    main()
