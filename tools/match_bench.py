#!/usr/bin/env python3
"""Brute-force Hamming matcher alone (for profiler passes and A/B runs of two builds): one frame's 8000 queries
against 100000 train descriptors, eight frames against the same 100000, and 128 pairs of 1000 x 1000, each with a
hash of the results.  Usage: match_bench.py [reps]"""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam_amd import ORBextractor, ORBmatcher  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex)
    g = torch.Generator(device="cpu").manual_seed(99)
    Q, T = 8000, 100000
    dq = torch.randint(0, 256, (8, Q, 32), dtype=torch.uint8, generator=g).to(dev)
    dt = torch.randint(0, 256, (T, 32), dtype=torch.uint8, generator=g).to(dev)
    dq3 = torch.randint(0, 256, (128, 1000, 32), dtype=torch.uint8, generator=g).to(dev)
    dt3 = torch.randint(0, 256, (128, 1000, 32), dtype=torch.uint8, generator=g).to(dev)
    outs = [torch.zeros((8, Q), dtype=torch.int32, device=dev) for _ in range(3)]
    outs3 = [torch.zeros((128, 1000), dtype=torch.int32, device=dev) for _ in range(3)]
    cases = [("8000 x 100000", lambda: m.match_bruteforce_device(dq, Q, Q, None, dt, T, T, None, 1, *outs), Q * T, outs),
             ("8 x 8000 x 100000", lambda: m.match_bruteforce_device(dq, Q, Q, None, dt, T, 0, None, 8, *outs), 8 * Q * T, outs),
             ("128 x 1000 x 1000", lambda: m.match_bruteforce_device(dq3, 1000, 1000, None, dt3, 1000, 1000, None, 128, *outs3), 128e6, outs3)]
    for name, fn, pairs, o in cases:
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            fn()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / reps
        h = hashlib.md5()
        for x in o:
            h.update(x.cpu().numpy().tobytes())
        print("%-18s %.4f ms per call, %.3e pair-distances/s, md5 %s" % (name, dt_ * 1e3, pairs / dt_, h.hexdigest()[:10]))
    ex.close()


if __name__ == "__main__":
    main()
