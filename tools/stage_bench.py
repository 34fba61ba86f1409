#!/usr/bin/env python3
"""Per-stage GPU times of the extraction path on a resident synthetic batch
(HIP events inside liborbx).  Usage: python tools/stage_bench.py [frames] [steps] [w h nfeat]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orb_slam_amd import ORBextractor  # noqa: E402
from orb_slam_amd.synth import synthetic_frame  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    W, H, NF = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (752, 480, 1000)
    uniq = min(B, 32)
    host = np.stack([synthetic_frame(W, H, 1234 + i) for i in range(uniq)])
    host = np.concatenate([host] * ((B + uniq - 1) // uniq))[:B]
    dev = torch.device("cuda:0")
    d_imgs = torch.from_numpy(host).to(dev)
    ex = ORBextractor(NF, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ex.set_stream(stream.cuda_stream)
    for _ in range(3):
        ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)
    torch.cuda.synchronize()
    ex.set_stage_timing(True)
    ex.reset_stage_times()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps):
        ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / steps
    st = ex.stage_times()
    print("batch %d x %dx%d: %.3f ms/step, %.0f frames/s, mean kps %.1f" % (B, W, H, ms, B / ms * 1e3, d_n.float().mean().item()))
    print("  " + "  ".join("%s %.3f" % (k, v[0] / max(1, v[1])) for k, v in st.items()))
    # the same loop without the stage events in the launch sequence
    ex.set_stage_timing(False)
    t0.record()
    for _ in range(steps):
        ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / steps
    print("without stage events: %.3f ms/step, %.0f frames/s" % (ms, B / ms * 1e3))


if __name__ == "__main__":
    main()
