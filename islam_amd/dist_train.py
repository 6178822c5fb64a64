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


class ShardedBatchNorm2d(torch.nn.BatchNorm2d):
    """BatchNorm2d whose batch statistics span the frames of ALL ranks.

    The "frozen" stereo network of the reference still normalises with train-mode statistics over the (2B, C, H, W)
    left/right-stacked batch (TartanVO.py:90-91).  When the B frames of a window are sharded over ranks (frame b on
    rank b mod P) each rank only sees 2B/P of them, so every layer all-reduces ONE packed vector [sum | sum of squares |
    count] (2C+1 values, <= 1 KB for C <= 128) and normalises with the global moments -- the result equals the
    un-sharded forward.  Forward-only: the stereo net is never differentiated on this path (fix_parts, TartanVO.py:109);
    with requires-grad inputs it raises instead of returning a wrong gradient."""

    group = None

    def forward(self, x):
        if not self.training or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return super().forward(x)
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            raise RuntimeError('ShardedBatchNorm2d is forward-only (frozen stereo net); got a tensor that requires grad')
        C = x.shape[1]
        xf = x.float()
        stat = torch.cat([xf.sum((0, 2, 3)), (xf * xf).sum((0, 2, 3)), xf.new_tensor([x.numel() / C])]).double()
        dist.all_reduce(stat, op=dist.ReduceOp.SUM, group=self.group)
        n = stat[-1]
        mean = stat[:C] / n
        var = (stat[C:2 * C] / n - mean * mean).clamp_(min=0.0)
        if self.track_running_stats:
            m = self.momentum if self.momentum is not None else 0.1
            with torch.no_grad():
                self.running_mean.mul_(1 - m).add_(m * mean.to(self.running_mean.dtype))
                self.running_var.mul_(1 - m).add_(m * (var * n / (n - 1)).to(self.running_var.dtype))
                self.num_batches_tracked += 1
        scale = (self.weight.double() / torch.sqrt(var + self.eps)).to(x.dtype)
        shift = (self.bias.double() - mean * self.weight.double() / torch.sqrt(var + self.eps)).to(x.dtype)
        return x * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1)


def shard_batchnorm(module, group=None):
    """Swap every BatchNorm2d under ``module`` for a ShardedBatchNorm2d sharing its parameters and buffers."""
    for name, child in list(module.named_children()):
        if isinstance(child, torch.nn.BatchNorm2d) and not isinstance(child, ShardedBatchNorm2d):
            new = ShardedBatchNorm2d(child.num_features, child.eps, child.momentum, child.affine, child.track_running_stats)
            new.weight, new.bias = child.weight, child.bias
            new.running_mean, new.running_var, new.num_batches_tracked = child.running_mean, child.running_var, child.num_batches_tracked
            new.group = group
            new.train(child.training)
            setattr(module, name, new)
        else:
            shard_batchnorm(child, group)
    return module


class FrameParallelVO(torch.nn.Module):
    """Front-end sharded over the frames of one window: rank r runs frames r, r+P, ... of the sample through TartanVO
    (PWC, pose head and the scale kernel are per-frame independent; the stereo net's BatchNorm layers are synchronised,
    see ShardedBatchNorm2d), then the (B,7) motions are all-gathered (28 B per frame) because PVGO needs the window."""

    def __init__(self, tartanvo, group=None):
        super().__init__()
        self.vo, self.group = tartanvo, group
        shard_batchnorm(tartanvo.vonet.stereoNet, group)

    @staticmethod
    def shard_sample(sample, rank, world):
        out = {}
        for k, v in sample.items():
            if isinstance(v, torch.Tensor) and v.dim() >= 1:
                out[k] = v[rank::world].contiguous()
            elif isinstance(v, (list, tuple)):
                out[k] = list(v[rank::world])
            else:
                out[k] = v
        return out

    def forward(self, sample, **kw):
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        B = sample['img0'].shape[0]
        if B % world:
            raise ValueError('window of %d frames does not split over %d ranks' % (B, world))
        res = self.vo(self.shard_sample(sample, rank, world), **kw)
        from . import lietensor as pp
        local = res['motion'].tensor() if isinstance(res['motion'], pp.LieTensor) else res['motion']
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.detach().contiguous(), group=self.group)
        parts[rank] = local                                    # keep this rank's rows differentiable
        full = torch.stack(parts, 1).reshape(B, local.shape[-1])          # frame b = parts[b % P][b // P]
        res = dict(res)
        res['motion_local'] = res['motion']
        res['motion'] = pp.SE3(full)
        return res
