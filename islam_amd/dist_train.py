"""Trajectory-parallel bilevel training (BASELINE configs[4]): whole trajectories are
independent units, one per GPU, weights replicated; the only exchange is ONE bucketed all-reduce of the pose-head
gradients (15.1 M fp32 = 60.5 MB) per epoch, right before ``vo_optimizer.step()`` (train.py:172-179 accumulates the
gradient over the whole trajectory and steps once per epoch).  The reference has no counterpart (single GPU)."""
import torch
import torch.distributed as dist


def allreduce_gradients(params, group=None, average=True, bucket_bytes=64 << 20):
    """Sum (or average) ``p.grad`` over the ranks with as few collectives as possible: gradients are packed into flat
    buckets of ``bucket_bytes`` (one bucket for the 60.5 MB pose head: xGMI rings are per-link bound, so few large
    messages beat many small ones).  Parameters without a gradient contribute zeros so every rank packs the same layout."""
    params = [p for p in params if p.requires_grad]
    world = dist.get_world_size(group)
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nb = p.numel() * p.element_size()
        if cur and cur_bytes + nb > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nb
    if cur:
        buckets.append(cur)
    for b in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in b])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        o = 0
        for p in b:
            n = p.numel()
            g = flat[o:o + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            o += n
    return len(buckets)


class TrajectoryParallel:
    """One BilevelLoop per rank on its own trajectory; ``end_epoch`` all-reduces the accumulated gradients and steps.
    Replicas stay bit-identical because every rank applies the same reduced gradient to the same weights."""

    def __init__(self, loop, group=None, average=True):
        self.loop, self.group, self.average = loop, group, average

    def step(self, sample, target='vo'):
        return self.loop.step(sample, target)

    def end_epoch(self):
        allreduce_gradients(self.loop.vo.vonet.flowPoseNet.parameters(), self.group, self.average)
        self.loop.end_epoch()
