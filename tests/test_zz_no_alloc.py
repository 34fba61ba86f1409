"""The entry point that allocates nothing (orbx_match_bruteforce_device_static) against
orbx_match_bruteforce_device: same results, two calls in a row, the second on a smaller input."""
import numpy as np
import pytest

from orb_slam_amd.synth import random_descriptors, map_descriptors

pytestmark = pytest.mark.gpu


def _both(m, ex, torch, q, t, nb=1):
    dev = torch.device("cuda", 0)
    Q, T = q.shape[-2], t.shape[-2]
    dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    shape = (nb, Q) if nb > 1 else (Q,)
    ref = [torch.full(shape, -7, dtype=torch.int32, device=dev) for _ in range(3)]
    got = [torch.full(shape, -9, dtype=torch.int32, device=dev) for _ in range(3)]
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    m.match_bruteforce_device(dq, Q, Q, None, dt, T, T, None, nb, *ref)
    m.match_bruteforce_device_static(dq, Q, Q, None, dt, T, T, None, nb, *got)
    torch.cuda.synchronize()
    ex.set_stream(None)
    return [g.cpu().numpy() for g in got], [r.cpu().numpy() for r in ref]


def test_static_scratch_entry_point_matches_allocating_one():
    torch = pytest.importorskip("torch")
    from orb_slam_amd import ORBextractor, ORBmatcher, OrbxError
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex)
    # first call: the benchmark's 8000 x 100 000 (64 splits, 3.9 MiB of partial results)
    q = random_descriptors(8000, 5)
    t = map_descriptors(100000, q, seed=99)
    t[70000] = t[123]
    got, ref = _both(m, ex, torch, q, t)
    for g, r, name in zip(got, ref, ("best_idx", "best_dist", "second_dist")):
        assert np.array_equal(g, r), name
    assert (ref[1] < 40).sum() > 1000          # the planted matches were found
    # second call, smaller: stale partial results of the first call must not leak in
    q2 = random_descriptors(3000, 7)
    t2 = map_descriptors(20000, q2, seed=8)
    got, ref = _both(m, ex, torch, q2, t2)
    for g, r, name in zip(got, ref, ("best_idx", "best_dist", "second_dist")):
        assert np.array_equal(g, r), name
    # third: batched and tiny (one split: no scratch at all)
    rng = np.random.default_rng(3)
    qb = rng.integers(0, 256, (6, 500, 32), dtype=np.uint8)
    tb = rng.integers(0, 256, (6, 700, 32), dtype=np.uint8)
    got, ref = _both(m, ex, torch, qb, tb, nb=6)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    # fourth: more partial results than the static scratch holds at the preferred split count (16 batches of 4000
    # queries, each train set in 16 splits: 7.8 MiB): fewer splits, the same results
    rng = np.random.default_rng(4)
    qc = rng.integers(0, 256, (16, 4000, 32), dtype=np.uint8)
    tc = rng.integers(0, 256, (16, 40000, 32), dtype=np.uint8)
    tc[3, 39999] = qc[3, 17]
    got, ref = _both(m, ex, torch, qc, tc, nb=16)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    assert got[0][3, 17] == 39999 and got[1][3, 17] == 0
    # a train set beyond the 22-bit index field: a clear refusal, nothing launched
    dev = torch.device("cuda", 0)
    big_q = torch.zeros((1, 64, 32), dtype=torch.uint8, device=dev)
    big_t = torch.zeros((1, 1 << 22, 32), dtype=torch.uint8, device=dev)
    outs = [torch.zeros((1, 64), dtype=torch.int32, device=dev) for _ in range(3)]
    with pytest.raises(OrbxError) as ei:
        m.match_bruteforce_device_static(big_q, 64, 64, None, big_t, 1 << 22, 1 << 22, None, 1, *outs)
    assert ei.value.code == -5 and "2^22" in str(ei.value)
    ex.close()


def test_static_scratch_entry_point_runs_under_stream_capture():
    """What the entry point is for: it must be recordable into a HIP graph (a capture forbids hipMalloc / hipFree and
    any synchronisation) at a shape never run before, and the replayed graph must give the allocating entry point's
    results on new data.  (torch's capture context frees its own cached blocks, so the device's free memory is not
    a usable witness here; the capture itself is: an allocation or a synchronisation inside it is an error.)"""
    torch = pytest.importorskip("torch")
    from orb_slam_amd import ORBextractor, ORBmatcher
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex)
    dev = torch.device("cuda", 0)
    Q, T = 5000, 60000                       # several splits of the train set: the static scratch is in use
    dq = torch.from_numpy(random_descriptors(Q, 21)).to(dev)
    dt = torch.from_numpy(map_descriptors(T, random_descriptors(Q, 21), seed=22)).to(dev)
    got = [torch.full((Q,), -9, dtype=torch.int32, device=dev) for _ in range(3)]
    ref = [torch.full((Q,), -7, dtype=torch.int32, device=dev) for _ in range(3)]
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    ex.set_stream(side.cuda_stream)
    with torch.cuda.graph(graph, stream=side):
        m.match_bruteforce_device_static(dq, Q, Q, None, dt, T, T, None, 1, *got)
    torch.cuda.synchronize()
    assert int(got[0][0].item()) == -9                       # the capture only recorded
    for rep in range(2):                                     # replay on changing data
        if rep == 1:
            dq.copy_(torch.from_numpy(random_descriptors(Q, 23)).to(dev))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        ex.set_stream(torch.cuda.current_stream().cuda_stream)
        m.match_bruteforce_device(dq, Q, Q, None, dt, T, T, None, 1, *ref)
        torch.cuda.synchronize()
        for g, r in zip(got, ref):
            assert torch.equal(g, r)
    ex.set_stream(None)
    ex.close()
