"""Multi-process test (gloo, world_size 2, CPU) of the sharding plumbing used
for N > 1 GPUs: contiguous frame shards, no data-path collective, one result
gather.  The per-frame results here come from the CPU oracle (this test has no
GPU); the GPU path fills the same tensors."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_range_partitions():
    from orb_slam_amd.shard import frame_range
    for n in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 8):
            rs = [frame_range(r, world, n) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        frame_range(2, 2, 10)


def _worker(rank, world, port, nframes, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle_lib import Oracle
    from orb_slam_amd.shard import frame_range, gather_features
    from orb_slam_amd.synth import synthetic_frame
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    b, e = frame_range(rank, world, nframes)
    o = Oracle(300, 1.2, 8, 20, 7)
    cap = 400
    n_out = torch.zeros(e - b, dtype=torch.int32)
    kps = torch.zeros((e - b, cap, 7), dtype=torch.float32)
    desc = torch.zeros((e - b, cap, 32), dtype=torch.uint8)
    for i in range(b, e):
        k, d = o.extract(synthetic_frame(240, 200, 900 + i), literal=False)
        n_out[i - b] = len(k)
        kps[i - b, :len(k)] = torch.from_numpy(k.view(np.float32).reshape(-1, 7).copy())
        desc[i - b, :len(k)] = torch.from_numpy(d)
    n_all, k_all, d_all = gather_features(n_out, kps, desc, nframes)
    if rank == 0:
        q.put((n_all.numpy(), k_all.numpy(), d_all.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gather_world2():
    import torch.multiprocessing as mp
    from oracle_lib import Oracle
    from orb_slam_amd.synth import synthetic_frame
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    nframes, world = 5, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, q)) for r in range(world)]
    for p in procs:
        p.start()
    n_all, k_all, d_all = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    o = Oracle(300, 1.2, 8, 20, 7)
    assert n_all.shape == (nframes,)
    for i in range(nframes):
        k, d = o.extract(synthetic_frame(240, 200, 900 + i), literal=False)
        assert n_all[i] == len(k) > 50
        assert np.array_equal(k_all[i, :len(k)].view(np.uint8), k.view(np.uint8).reshape(len(k), 28))
        assert np.array_equal(d_all[i, :len(k)], d)
