"""Interval analysis of a rocprofv3 kernel trace of scripts/vio_only.py: the pipelined phase = the last `steps` bilevel steps (the
measurement runs the sequential schedule first).  Time with >= 1 / >= 2 kernels running, per-queue busy time, kernel time by family."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']))
steps = int(sys.argv[2])
rows.sort()
# the pipelined phase: find the last `steps` launches of the pose head's backward marker kernel? simpler: one frozen replay per step --
# the fused first pyramid level kernel runs once per flow forward
marks = [a for a, b, q, n in rows if 'pyr_level_kernel<3' in n]
assert len(marks) >= steps + 2, len(marks)
t0, t1 = marks[-steps - 1], marks[-1]          # `steps` whole periods of the pipelined schedule
win = [(max(a, t0), min(b, t1), q, n) for a, b, q, n in rows if b > t0 and a < t1]
span = (t1 - t0) / 1e6
pts = sorted([(a, 1) for a, b, q, n in win] + [(b, -1) for a, b, q, n in win])
depth, last, b1, b2 = 0, t0, 0, 0
for t, d in pts:
    if depth >= 1: b1 += t - last
    if depth >= 2: b2 += t - last
    depth += d; last = t
print('pipelined phase: %.2f ms per step over %d steps; >= 1 kernel running %.1f %%, >= 2 running %.1f %%; kernel-time sum %.2f ms per step; %d launches per step'
      % (span / steps, steps, 100.0 * b1 / (t1 - t0), 100.0 * b2 / (t1 - t0), sum(b - a for a, b, q, n in win) / 1e6 / steps, len(win) // steps))
perq = collections.defaultdict(list)
for a, b, q, n in win:
    perq[q].append((a, b))
for q, iv in sorted(perq.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
    iv.sort(); tot, end = 0, None
    for a, b in iv:
        if end is None or a > end: tot += b - a; end = b
        elif b > end: tot += b - end; end = b
    print('  queue %s: %d launches per step, busy (union) %.2f ms per step, kernel-time sum %.2f ms per step' % (q, len(iv) // steps, tot / 1e6 / steps, sum(b - a for a, b in iv) / 1e6 / steps))
fam = collections.defaultdict(lambda: [0, 0])
def family(n):
    for key in ('conv_nhwc_kernel', 'hg_residual', 'conv3x3_mfma', 'pyr_level', 'corr81', 'warp_mask', 'resize_bilinear', 'bn_', 'partial_fold', 'maxpool',
                'block_mean', 'deconv4x4s2', 'nchw_to_nhwc', 'igemm', 'naive_conv', 'SubTensorOp', 'trial_elim', 'bt_', 'linbuild', 'trial_lin', 'imu', 'scan_kernel',
                'chain_rot', 'finish_kernel', 'frame_kernel', 'scale_', 'edge_mask', 'vo_loss', 'elementwise', 'Cat', 'gemm', 'Cijk', 'reduce', 'ck::', 'tensor_operation'):
        if key in n:
            return key
    return n[:40]
for a, b, q, n in win:
    f = fam[family(n)]; f[0] += b - a; f[1] += 1
print('kernel time by family (ms per step, launches per step):')
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:28]:
    print('  %-28s %7.3f  %5d' % (k, v[0] / 1e6 / steps, v[1] // steps))
