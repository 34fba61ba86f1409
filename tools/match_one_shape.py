#!/usr/bin/env python3
"""One shape of the brute-force search alone, a few launches (for counter passes: one kernel, one grid):
8 frames of 8000 queries against one shared set of 100000 train descriptors.  Usage: match_one_shape.py [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam_amd import ORBextractor, ORBmatcher  # noqa: E402


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex)
    g = torch.Generator(device="cpu").manual_seed(1)
    nb, Q, T = 8, 8000, 100000
    q = torch.randint(0, 256, (nb, Q, 32), dtype=torch.uint8, generator=g).to(dev)
    t = torch.randint(0, 256, (1, T, 32), dtype=torch.uint8, generator=g).to(dev)
    outs = [torch.zeros((nb, Q), dtype=torch.int32, device=dev) for _ in range(3)]
    for _ in range(launches):
        m.match_bruteforce_device(q, Q, Q, None, t, T, 0, None, nb, *outs)
    torch.cuda.synchronize()
    print("matched", int((outs[0] >= 0).sum().item()), "of", nb * Q)


if __name__ == "__main__":
    main()
