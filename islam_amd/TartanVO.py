"""TartanVO front-end with the reference's call surface (reference TartanVO.py:15-198).

Differences in HOW (not WHAT): the edge mask is computed on the device (islam_amd/edges.py) instead of a
host round trip through OpenCV; the per-sample Python loop over scale_from_disp_flow is ONE batched HIP
launch (islam_scale_ls) whose gradient w.r.t. the pose is rebuilt from the first-order sums the kernel
accumulates; the two frozen nets can run under bf16 autocast (BASELINE config 2: "bf16 nets / fp64 LM").
"""
import numpy as np
import torch
import torch.nn as nn

from . import lietensor as pp
from . import ops
from .edges import edge_mask
from .nets import VONet
from .transformation import cvtSE3_pypose, tartan2kitti_pypose

DISP_TH = {'kitti': 5, 'tartanair': 1, 'euroc': 1}          # TartanVO.py:161


def stereo_scale(disp, flow, pose_enu, intr4, baseline, edge, disp_th, depth_input=False):
    """Batched dense_ba.scale_from_disp_flow (dense_ba.py:88-176), differentiable w.r.t. ``pose_enu`` (SE3 LieTensor).

    Value: the HIP reduction.  Gradient: s = Mw/MM with M linear in a = K t^ and w linear in R, so
    ds = (dMw - s dMM)/MM is a linear functional of (a, R) whose coefficients are sums over the masked pixels
    (islam_amd/csrc/scale_ls.hip); it is re-attached through the same LieTensor ops the reference differentiates
    (T.Inv().rotation() acting on points, T.Inv().translation(), dense_ba.py:142-166)."""
    s, z, mask, dmask, sums = ops.scale_ls(disp, flow, pose_enu.tensor().detach().to(disp.device), intr4, baseline, edge, disp_th,
                                           depth_input=depth_input)
    if not pose_enu.requires_grad:
        return s, z, mask, dmask
    dev, dt = pose_enu.device, pose_enu.dtype          # the (B,6)-sized gradient glue runs where the pose lives (host or device)
    sums = sums.to(dev, dt)
    s = s.to(dev)
    intr4 = intr4.to(dev, dt)
    fx, fy, cx, cy = intr4.unbind(-1)
    MM = sums[:, 0]
    sv = s.to(dt)
    # d s / d a  and  d s / d R  (B,3), (B,3,3)
    dMw_da = torch.stack([-sums[:, 2], -sums[:, 3], sums[:, 4]], -1)
    dMM_da = torch.stack([-2 * sums[:, 5], -2 * sums[:, 6], 2 * sums[:, 7]], -1)
    ga = (dMw_da - sv[:, None] * dMM_da) / MM[:, None]
    GR = torch.stack([fx[:, None] * sums[:, 8:11], fy[:, None] * sums[:, 11:14], sums[:, 14:17]], 1) / MM[:, None, None]
    Tinv = pose_enu.Inv()
    t = Tinv.translation()
    tn = torch.nn.functional.normalize(t, dim=-1)
    a = torch.stack([fx * tn[:, 0] + cx * tn[:, 2], fy * tn[:, 1] + cy * tn[:, 2], tn[:, 2]], -1)
    R = Tinv.rotation()
    eye = torch.eye(3, dtype=dt, device=dev)
    cols = torch.stack([R.Act(eye[j].expand(R.shape[0], 3)) for j in range(3)], -1)      # (B,3,3), column j = R e_j
    sur = (ga.detach() * a).sum(-1) + (GR.detach() * cols).sum((-1, -2))
    return sv.detach() + (sur - sur.detach()), z, mask, dmask


class TartanVO(nn.Module):
    def __init__(self, vo_model_name=None, pose_model_name=None, flow_model_name=None, stereo_model_name=None,
                 device_id=0, correct_scale=True, fix_parts=(), use_kitti_coord=True, frozen_dtype=None, flow_dtype=None,
                 host_glue=False, miopen_find=False, pose_channels_last=False, graph_frozen=False, graph_pose=False, pose_dtype=None,
                 graph_instances=1):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError('islam_amd.TartanVO runs on the MI355X only; there is no CPU fallback')
        # miopen_find: let MIOpen time its candidate kernels for every convolution shape the first time it is seen
        # (torch.backends.cudnn.benchmark, process-wide) instead of trusting its heuristics: the first forward + backward take
        # ~1-2 minutes longer, the stereo net then runs ~25 % faster (the heuristics pick split-K kernels that need
        # zero-fill and cast passes around them)
        if miopen_find:
            from .miopen_pin import use_pinned_db, check_pinned_db
            use_pinned_db()          # the search results of one MI355X run, shipped with the package: same kernels every run, no search
            check_pinned_db(device_id)   # warns (raises under ISLAM_MIOPEN_PIN_STRICT=1) when the set belongs to another MIOpen build / device
            torch.backends.cudnn.benchmark = True
        self.device_id = device_id
        self.correct_scale = correct_scale
        self.use_kitti_coord = use_kitti_coord
        self.pose_std = torch.tensor([0.13, 0.13, 0.13, 0.013, 0.013, 0.013]).cuda(self.device_id)
        self.vonet = VONet(fix_parts=fix_parts)
        self.vonet.set_frozen_dtype(frozen_dtype, flow_dtype)
        # host_glue: the (B,6)->(B,7) pose algebra after the networks (unit rescale, frame changes, scale re-attachment:
        # ~150 tiny tensor ops and as many autograd nodes) runs in float64 on the host instead of one device launch per op;
        # res['motion'] is still returned on the device, res['motion_host'] is the same LieTensor on the host
        self.host_glue = host_glue
        self.fused_glue = True        # host_glue as ONE autograd node (islam_amd/glue.py); False: operator by operator (A/B runs, tests)
        for name, part in ((vo_model_name, self.vonet), (flow_model_name, self.vonet.flowNet),
                           (pose_model_name, self.vonet.flowPoseNet), (stereo_model_name, self.vonet.stereoNet)):
            if name is not None and name != '':
                self.load_model(part, name)
        self.vonet = self.vonet.cuda(self.device_id)
        if pose_channels_last:
            self.vonet.set_pose_channels_last(True)
        if graph_frozen:        # the frozen flow + disparity forward replays from a HIP graph (VONet.set_graph_frozen)
            self.vonet.set_graph_frozen(True)
        # captured copies of the frozen forward, used round-robin: N copies let prefetch() queue N batches ahead (VONet._frozen_graphed)
        self.vonet.graph_instances = max(1, int(graph_instances))
        # forward + backward of the trainable pose head as HIP graphs; 'accumulate': the backward node adds the parameter gradients to
        # .grad itself (nets._PoseGraph; torch.autograd.grad callers list vonet.pose_graph_leaf() among their inputs)
        self.vonet.graph_pose = graph_pose if graph_pose in ('accumulate', 'hip') else bool(graph_pose)
        self.vonet.pose_dtype = pose_dtype           # bf16 autocast for the trainable pose head (fp32 master weights)

    def load_model(self, model, modelname):
        """TartanVO.py:49-87: suffix matching of state-dict keys with a size check."""
        pretrain = torch.load(modelname, map_location='cuda:%d' % self.device_id)
        own = model.state_dict()
        picked = {}
        for k, v in pretrain.items():
            for kk, vv in own.items():
                if (k.endswith(kk) or kk.endswith(k)) and v.size() == vv.size():
                    picked[kk] = v
        if not picked:
            raise Exception('Could not load model from %s.' % modelname, 'load_model')
        for kk in own:
            if kk not in picked:
                print('! [load_model] Key {} in model but not in {}!'.format(kk, modelname))
        own.update(picked)
        model.load_state_dict(own)
        if hasattr(self, 'vonet'):
            self.vonet.reset_graphs()          # captured graphs hold the old frozen weights
        return model

    def prefetch(self, sample, is_train=True):
        """Software pipelining across batches: run the frozen part of forward(sample) -- flow + disparity, 95 % of the GPU
        time of a forward -- on a side stream NOW, while the caller is still busy with the previous batch (IMU, PVGO,
        backward: small kernels and host work that leave the GPU mostly idle).  forward(sample) on the same dict object
        picks the result up.  Several batches may be under way at once (one entry per sample dict; the side stream runs them in
        the order they were asked for).  Only when both nets are frozen (no autograd state; the reference detaches flow and disparity
        anyway, TartanVO.py:109-110); returns False otherwise and forward() computes everything inline."""
        if self.vonet.frozen_nets_trainable():
            return False
        if getattr(self, '_prefetched', None) is None:
            self._prefetched = {}
        if id(sample) in self._prefetched:         # already under way (prefetch depth > 1: BilevelLoop.step asks again)
            return True
        dev = self.device_id
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=dev)
        self._side.wait_stream(torch.cuda.current_stream(dev))
        self.vonet.set_mode(is_train)
        with torch.cuda.stream(self._side), torch.no_grad():
            imgs = [sample[k].cuda(dev, non_blocking=True) for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm')]
            flow, disp = self.vonet.frozen_forward(*imgs)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._prefetched[id(sample)] = (sample, flow, disp, ev)
        return True

    def forward(self, sample, is_train=True, given_scale=None, need_grad=None):
        """need_grad (default: is_train): whether the forward builds autograd state.  is_train alone selects the BatchNorm mode (F4);
        a caller that wants train-mode statistics but no gradient (the VO forward of an IMU epoch, train.py:207-212) passes
        need_grad=False -- no autograd graph, and a graphed pose head (graph_pose) does not replay its training graph."""
        self.vonet.set_mode(is_train)                                       # BN batch statistics when training (F4)
        with torch.set_grad_enabled(is_train if need_grad is None else bool(need_grad)):
            dev = self.device_id
            img0 = sample['img0'].cuda(dev, non_blocking=True)
            img1 = sample['img1'].cuda(dev, non_blocking=True)
            intrinsic = sample['intrinsic'].cuda(dev, non_blocking=True)
            img0_norm = sample['img0_norm'].cuda(dev, non_blocking=True)
            img0_r_norm = sample['img0_r_norm'].cuda(dev, non_blocking=True)
            frozen = None
            pre = (getattr(self, '_prefetched', None) or {}).pop(id(sample), None)
            if pre is not None and pre[0] is sample:
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(pre[3])
                for t in pre[1:3]:
                    t.record_stream(cur)
                frozen = (pre[1], pre[2])
            intrinsic_calib = sample['intrinsic_calib']
            baseline = torch.linalg.norm(sample['extrinsic'][:, :3], dim=1)
            precalc_flow = sample['flow'] if 'flow' in sample else None

            flow, disp, pose = self.vonet(img0, img1, img0_norm, img0_r_norm, intrinsic, frozen=frozen)
            if self.host_glue and self.fused_glue and given_scale is None and not self.correct_scale and pose.is_cuda:
                # the whole (B,6) -> (B,7) algebra below as ONE autograd node in vectorised numpy (islam_amd/glue.py): same arithmetic,
                # same gradient conventions, two device reads instead of three, ~100 fewer autograd nodes on the step's main chain
                from .glue import fused_pose_glue
                flow, disp = flow.detach(), disp.detach()
                flow = flow * 5 if precalc_flow is None else precalc_flow.cuda(dev)
                disp = disp * (50 / 4)
                th = torch.tensor([float(DISP_TH[d]) for d in sample['datatype']])
                motion, scale, depth, mask, depth_mask = fused_pose_glue(pose, self.pose_std.double().cpu().numpy(), disp, flow, intrinsic_calib.float() / 4,
                                                                         baseline.float(), edge_mask(img0), th, self.use_kitti_coord)
                return {'flow': flow, 'disp': disp, 'mask': mask, 'depth': depth, 'depth_mask': depth_mask, 'baseline': baseline[0],
                        'intrinsic': intrinsic_calib[0] / 4, 'motion_host': motion, 'motion': pp.SE3(motion.tensor().float().cuda(dev))}
            if self.host_glue:
                pose = pose.double().cpu() * self.pose_std.double().cpu()
            else:
                pose = pose * self.pose_std
            flow, disp = flow.detach(), disp.detach()
            res = {}
            if given_scale is not None:
                trans = torch.nn.functional.normalize(pose[:, :3], dim=1) * given_scale.to(pose.device, pose.dtype).view(-1, 1)
                pose = torch.cat([trans, pose[:, 3:]], dim=1)
            elif not self.correct_scale:
                flow = flow * 5 if precalc_flow is None else precalc_flow.cuda(dev)     # pixels at 1/4 res
                disp = disp * (50 / 4)
                pose_enu = tartan2kitti_pypose(pose)
                edge = edge_mask(img0)
                th = torch.tensor([float(DISP_TH[d]) for d in sample['datatype']])
                intr4 = intrinsic_calib.float() / 4
                scale, depth, mask, depth_mask = stereo_scale(disp, flow, pose_enu, intr4, baseline.float(), edge, th)
                res.update(flow=flow, disp=disp, mask=mask, depth=depth, depth_mask=depth_mask, baseline=baseline[0],
                           intrinsic=intrinsic_calib[0] / 4)
                trans = torch.nn.functional.normalize(pose[:, :3], dim=1) * scale.to(pose.device, pose.dtype).view(-1, 1)
                pose = torch.cat([trans, pose[:, 3:]], dim=1)
            else:
                scale = torch.norm(sample['motion'][:, :3], dim=1).to(pose.device, pose.dtype)
                trans = torch.nn.functional.normalize(pose[:, :3], dim=1) * scale.view(-1, 1)
                pose = torch.cat([trans, pose[:, 3:]], dim=1)
            motion = tartan2kitti_pypose(pose) if self.use_kitti_coord else cvtSE3_pypose(pose)
            if self.host_glue:
                res['motion_host'] = motion
                motion = pp.SE3(motion.tensor().float().cuda(dev))
            res['motion'] = motion
            return res
