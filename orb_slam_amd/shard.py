"""Multi-GPU sharding of the extraction path.

Frames are independent, so a batch or a sequence shards by contiguous frame
ranges with no collective on the data path; the only communication is one
result gather at the end (per-frame counts, then the padded keypoint and
descriptor blocks).  Works with any torch.distributed backend: "nccl" (RCCL
over xGMI) with device tensors, "gloo" with host tensors.
"""


def frame_range(rank, world, nframes):
    """Contiguous shard [begin, end) of rank `rank`: [rank*n/world, (rank+1)*n/world)."""
    if not (0 <= rank < world) or nframes < 0:
        raise ValueError("bad rank/world/nframes")
    return (rank * nframes) // world, ((rank + 1) * nframes) // world


def gather_features(n_out, kps, desc, nframes_total, group=None):
    """All-gather the per-rank results of a sharded batch, in frame order.

    n_out: int32 [F_local]; kps: float32 [F_local, cap, 7]; desc: uint8 [F_local, cap, 32]
    (F_local = this rank's shard, cap identical on every rank).  Returns tensors
    for all `nframes_total` frames on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    cap = kps.shape[1]
    sizes = [frame_range(r, world, nframes_total)[1] - frame_range(r, world, nframes_total)[0] for r in range(world)]
    if n_out.shape[0] != sizes[rank]:
        raise ValueError("this rank holds %d frames, its shard has %d" % (n_out.shape[0], sizes[rank]))
    fmax = max(sizes)

    def padded(t):
        if t.shape[0] == fmax:
            return t.contiguous()
        pad = torch.zeros((fmax - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        return torch.cat([t, pad], 0).contiguous()

    outs = []
    for t in (n_out, kps.reshape(kps.shape[0], cap * 7), desc.reshape(desc.shape[0], cap * 32)):
        p = padded(t)
        buf = [torch.empty_like(p) for _ in range(world)]
        dist.all_gather(buf, p, group=group)
        outs.append(torch.cat([buf[r][:sizes[r]] for r in range(world)], 0))
    return outs[0], outs[1].reshape(nframes_total, cap, 7), outs[2].reshape(nframes_total, cap, 32)
