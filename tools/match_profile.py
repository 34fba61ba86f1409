#!/usr/bin/env python3
"""Where a wavefront of the brute-force matcher spends its cycles (8 x 8000 x 100000): expansion of the next group's
fragments, wait at the barrier, the sweeps of a group's tiles.
Needs the diagnostic build: make -C orb_slam_amd/csrc OUT=$PWD/build/liborbx_mp.so EXTRA=-DORBX_MATCH_PROFILE
Usage: ORBX_LIBRARY=build/liborbx_mp.so python tools/match_profile.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam_amd import ORBextractor, ORBmatcher, load_library  # noqa: E402


def main():
    lib = load_library()
    dev = torch.device("cuda:0")
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex)
    g = torch.Generator(device="cpu").manual_seed(99)
    Q, T = 8000, 100000
    dq = torch.randint(0, 256, (8, Q, 32), dtype=torch.uint8, generator=g).to(dev)
    dt = torch.randint(0, 256, (T, 32), dtype=torch.uint8, generator=g).to(dev)
    outs = [torch.zeros((8, Q), dtype=torch.int32, device=dev) for _ in range(3)]
    out = (C.c_uint64 * 8)()
    for _ in range(20):
        m.match_bruteforce_device(dq, Q, Q, None, dt, T, 0, None, 8, *outs)
    torch.cuda.synchronize()
    lib.orbx_debug_match_profile(out, 1)
    m.match_bruteforce_device(dq, Q, Q, None, dt, T, 0, None, 8, *outs)
    torch.cuda.synchronize()
    lib.orbx_debug_match_profile(out, 1)
    v = [int(x) for x in out]
    waves = max(1, v[4])
    groups = (T + 15) // 16 // 128 + 1          # groups of four tiles per wavefront, roughly (16 splits)
    tot = sum(v[:4])
    print("%d wavefronts, %.0f cycles each in the loop (~%d groups of 4 tiles)" % (waves, tot / waves, groups))
    for k, name in enumerate(["expansion of the next group", "barrier", "sweeps of 4 tiles", "loop overhead"]):
        print("   %-30s %5.1f %%  %9.0f cycles per wavefront" % (name, 100.0 * v[k] / tot, v[k] / waves))
    ex.close()


if __name__ == "__main__":
    main()
