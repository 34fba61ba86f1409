#!/usr/bin/env python3
"""Randomised parity soak of the brute-force Hamming search: random query / train counts, batches with ragged
counts, a train set shared by the whole batch, descriptors with planted duplicates and near-duplicates, both forms
of the sweep (the library's own choice: FP4 matrix instructions from 64 train descriptors on; the scalar sweep for
every size) against the CPU oracle; best index, best distance and second distance must agree exactly.
Usage: python tools/soak_match.py [n_cases] [seed]"""
import os
os.environ.setdefault("ORBX_SWITCHES_PER_CALL", "1")   # the form switch below is read at every call only then
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from orb_slam_amd import ORBextractor, ORBmatcher  # noqa: E402


def descriptors(rng, n, kind):
    if kind == 0:
        return rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if kind == 1:                               # sparse and dense bit patterns: extreme popcounts
        d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        d[::3] &= rng.integers(0, 256, (len(d[::3]), 32), dtype=np.uint8) & rng.integers(0, 256, (len(d[::3]), 32), dtype=np.uint8)
        d[1::3] |= rng.integers(0, 256, (len(d[1::3]), 32), dtype=np.uint8) | rng.integers(0, 256, (len(d[1::3]), 32), dtype=np.uint8)
        return d
    base = rng.integers(0, 256, (max(1, n // 50), 32), dtype=np.uint8)     # clusters: many near ties
    d = base[rng.integers(0, len(base), n)].copy()
    flips = rng.integers(0, 256, (n, 3))
    for j in range(3):
        d[np.arange(n), flips[:, j] // 8] ^= (1 << (flips[:, j] % 8)).astype(np.uint8)
    return d


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device("cuda:0")
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex)
    orc = Oracle(1000, 1.2, 8, 20, 7)
    bad = 0
    pairs = 0
    for case in range(n_cases):
        nb = int(rng.choice([1, 1, 2, 5]))
        qs = int(rng.choice([1, 31, 33, 64, 255, 257, 700, 1000, 2100, 4097]))
        ts = int(rng.choice([1, 63, 65, 300, 1000, 4097, 9000, 20000, 50000 // nb]))
        shared = nb > 1 and rng.integers(0, 2) == 1
        kind = int(rng.integers(0, 3))
        q = descriptors(rng, nb * qs, kind).reshape(nb, qs, 32)
        t = descriptors(rng, (1 if shared else nb) * ts, kind).reshape(-1, ts, 32)
        if ts > 8 and qs > 4:                   # exact duplicates: the lower index wins, second distance 0
            t[0, ts // 2] = t[0, ts // 3] = q[0, 3]
        nq = rng.integers(1, qs + 1, nb).astype(np.int32) if rng.integers(0, 2) else np.full(nb, qs, np.int32)
        nt = rng.integers(1, ts + 1, nb).astype(np.int32) if (rng.integers(0, 2) and not shared) else np.full(nb, ts, np.int32)
        dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
        dnq, dnt = torch.from_numpy(nq).to(dev), torch.from_numpy(nt).to(dev)
        ref = [orc.match_bruteforce(q[b, :nq[b]], t[0 if shared else b, :nt[b]]) for b in range(nb)]
        for form in ("auto", "scalar"):
            os.environ.pop("ORBX_MATCH_FORM", None)
            if form == "scalar":
                os.environ["ORBX_MATCH_FORM"] = form
            outs = [torch.full((nb, qs), -7, dtype=torch.int32, device=dev) for _ in range(3)]
            m.match_bruteforce_device(dq, qs, qs, dnq, dt, ts, 0 if shared else ts, dnt, nb, *outs)
            torch.cuda.synchronize()
            got = [o.cpu().numpy() for o in outs]
            for b in range(nb):
                ok = all(np.array_equal(got[i][b, :nq[b]], ref[b][i]) for i in range(3))
                if not ok:
                    bad += 1
                    print("MISMATCH case %d form %s batch %d of %d: %d x %d (stride %d x %d, shared %s, kind %d)" % (
                        case, form, b, nb, nq[b], nt[b], qs, ts, shared, kind))
        pairs += int((nq.astype(np.int64) * nt).sum())
    os.environ.pop("ORBX_MATCH_FORM", None)
    print("soak_match: %d cases, %.3g pair distances per form, %d mismatches" % (n_cases, pairs, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
