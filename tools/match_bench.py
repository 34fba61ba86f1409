#!/usr/bin/env python3
"""Brute-force Hamming matcher alone (for profiler passes): 8000 queries x 100000 train descriptors.
Usage: match_bench.py [reps]"""
import sys
import time

import torch

from orb_slam_amd import ORBextractor, ORBmatcher


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex)
    g = torch.Generator(device="cpu").manual_seed(99)
    Q, T = 8000, 100000
    dq = torch.randint(0, 256, (Q, 32), dtype=torch.uint8, generator=g).to(dev)
    dt = torch.randint(0, 256, (T, 32), dtype=torch.uint8, generator=g).to(dev)
    outs = [torch.zeros((Q,), dtype=torch.int32, device=dev) for _ in range(3)]
    for _ in range(3):
        m.match_bruteforce_device(dq, Q, Q, None, dt, T, T, None, 1, *outs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.match_bruteforce_device(dq, Q, Q, None, dt, T, T, None, 1, *outs)
    ex.synchronize() if hasattr(ex, "synchronize") else None
    torch.cuda.synchronize()
    dt_ = time.perf_counter() - t0
    print("%d x %d: %.4f ms per call, %.3e pair-distances/s" % (Q, T, dt_ / reps * 1e3, Q * T * reps / dt_))


if __name__ == "__main__":
    main()
