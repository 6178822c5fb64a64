"""ctypes binding of libislam_hip.so (the C ABI declared in include/islam_hip.h).

The library is built in-tree by ``islam_amd/csrc/Makefile`` (``__graft_entry__.build()``).  There
is no CPU fallback: if the shared object is missing or a call fails, an exception is raised.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ISLAM_HIP_LIB') or os.path.join(_HERE, 'lib', 'libislam_hip.so')      # (ISLAM_HIP_LIB: A/B runs of two builds)
_lib = None

c_void_p, c_int, c_int64, c_double, c_float, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                         ctypes.c_double, ctypes.c_float, ctypes.c_size_t)

SCALE_NBLK, SCALE_NSUM = 16, 18
MAX_LEVELS = 6          # ISLAM_PVGO_MAX_LEVELS


class PvgoParams(ctypes.Structure):
    _fields_ = [('w', c_double * 4), ('radius', c_double), ('vmin', c_double), ('vmax', c_double),
                ('high', c_double), ('low', c_double), ('up', c_double), ('down', c_double), ('factor', c_double),
                ('rmin', c_double), ('rmax', c_double), ('reject', c_int), ('max_steps', c_int), ('patience', c_int),
                ('decreasing', c_double), ('seg_len', c_int * 2)]


class PvgoResult(ctypes.Structure):
    _fields_ = [('steps', c_int), ('trials', c_int), ('status', c_int), ('loss', c_double), ('damping', c_double)]


class PvgoReproj(ctypes.Structure):
    _fields_ = [('points', c_void_p), ('targets', c_void_p), ('K', c_int), ('fx', c_double), ('fy', c_double),
                ('cx', c_double), ('cy', c_double), ('rgb2imu', c_double * 7), ('weight', c_double),
                ('compat_first_motion', c_int)]


REPROJ_REC = 32         # ISLAM_REPROJ_REC

# name -> (restype, argtypes); every symbol include/islam_hip.h declares
SIGNATURES = {
    'islam_last_error': (ctypes.c_char_p, []),
    'islam_abi_version': (c_int, []),
    'islam_clock_probe': (c_int, [c_void_p, c_int, c_void_p]),
    'islam_wall_clock_khz': (c_int, [c_int]),
    'islam_launch_cost_probe': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p]),
    'islam_corr81_scratch_bytes': (c_size_t, [c_int] * 4),
    'islam_corr81_fwd': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_void_p]),
    'islam_corr81_bwd': (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    'islam_warp_mask': (c_int, [c_void_p, c_void_p, c_float, c_void_p] + [c_int] * 4 + [c_void_p]),
    'islam_warp_mask_bwd': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    'islam_deconv4x4s2_to2_f32': (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    'islam_corr81_fwd_act': (c_int, [c_void_p] * 3 + [c_int, c_int, c_float] + [c_int] * 4 + [c_void_p, c_void_p]),
    'islam_flow_head_up_f32': (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    'islam_pyramid_packed_elems': (c_size_t, [c_int, c_int]),
    'islam_flow_pyramid_level': (c_int, [c_void_p] * 8 + [c_int] * 5 + [c_float, c_void_p]),
    'islam_flow_pyramid_level_pair': (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_float, c_void_p]),
    'islam_conv3x3_packed_elems': (c_size_t, [c_int, c_int]),
    'islam_conv3x3_mfma': (c_int, [c_void_p] * 4 + [c_int] * 11 + [c_float, c_void_p]),
    'islam_resize_bilinear_nhwc_bf16': (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    'islam_conv_nhwc_flow': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    'islam_nchw_f32_to_nhwc_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'islam_resize_bilinear_add_nhwc_bf16': (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p]),
    'islam_resize_bilinear_add_nhwc_bf16_into': (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p]),
    'islam_maxpool2_nhwc_bf16': (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    'islam_half_image_into_nhwc_bf16': (c_int, [c_void_p] * 2 + [c_int] * 6 + [c_void_p]),
    'islam_avgpool_nhwc_bf16': (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    'islam_resize_bilinear_nhwc_bf16_into': (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    'islam_stack_pair_pad8_nhwc_bf16': (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    'islam_stereo_pair_prepare_f32': (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    'islam_upsample_cat_nhwc_bf16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p] + [c_int] * 6 + [c_void_p]),
    'islam_bias_act_add_nhwc_bf16': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p]),
    'islam_bias_act_f32_nhwc': (c_int, [c_void_p] * 4 + [ctypes.c_longlong, c_int, c_int, c_void_p]),
    'islam_bias_act_bwd_scratch_floats': (ctypes.c_longlong, [ctypes.c_longlong, c_int]),
    'islam_bias_act_bwd_f32_nhwc': (c_int, [c_void_p] * 6 + [ctypes.c_longlong, c_int, c_int, c_void_p]),
    'islam_pose_head_workspace_bytes': (c_size_t, [c_int] * 3),
    'islam_pose_head_forward': (c_int, [c_void_p] * 4 + [c_size_t] + [c_int] * 3 + [c_void_p]),
    'islam_pose_head_backward': (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 4 + [c_void_p]),
    'islam_bn_scratch_floats': (c_size_t, [c_int]),
    'islam_bn_train_nhwc_bf16': (c_int, [c_void_p] * 8 + [c_double, c_double, c_int, ctypes.c_longlong, c_int, c_void_p, c_void_p]),
    'islam_conv_nhwc_packed_elems': (c_size_t, [c_int] * 3),
    'islam_deconv_nhwc_packed_elems': (c_size_t, [c_int] * 2),
    'islam_deconv4x4s2_nhwc_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    'islam_deconv4x4s2_nhwc_bf16_cat': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    'islam_conv_nhwc_bf16_s2_cat': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    'islam_conv_nhwc_stat_blocks': (c_int, [c_int] * 4),
    'islam_conv_nhwc_stats_floats': (c_size_t, [c_int] * 4),
    'islam_conv_nhwc_bf16': (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_void_p]),
    'islam_conv_nhwc_s2_stats_floats': (c_size_t, [c_int] * 4),
    'islam_conv_ws_mode': (c_int, [c_int]),
    'islam_conv_ws_launch_counts': (c_int, [c_void_p]),
    'islam_conv_nhwc_bf16_s2': (c_int, [c_void_p] * 7 + [c_int] * 9 + [c_void_p]),
    'islam_conv_nhwc_bf16_into': (c_int, [c_void_p] * 5 + [c_int] * 9 + [c_void_p]),
    'islam_conv_nhwc_bf16_bn': (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p] * 5 + [c_double, c_double] + [c_void_p] * 3),
    'islam_hg_residual_packed_elems': (c_size_t, [c_int, c_int]),
    'islam_hg_residual_nhwc_bf16': (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    'islam_bn_finalize': (c_int, [c_void_p, c_double] + [c_void_p] * 5 + [c_double, c_double, c_int, c_void_p, c_void_p]),
    'islam_bn_apply_nhwc_bf16': (c_int, [c_void_p] * 4 + [c_int, ctypes.c_longlong, c_int, c_void_p]),
    'islam_edge_mask_max_pixels': (c_int, []),
    'islam_edge_mask': (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    'islam_scale_ls': (c_int, [c_void_p] * 13 + [c_int] * 3 + [c_void_p]),
    'islam_scale_ls_depth': (c_int, [c_void_p] * 12 + [c_int] * 3 + [c_void_p]),
    'islam_imu_scratch_bytes': (c_size_t, [c_int64, c_int, c_int]),
    'islam_imu_preint': (c_int, [c_void_p] * 4 + [c_int, c_int64, c_int] + [c_void_p] * 3 + [c_double, c_int] +
                         [c_void_p] * 4 + [c_int, c_void_p]),
    'islam_imu_preint_both': (c_int, [c_void_p] * 4 + [c_int, c_int64, c_int] + [c_void_p] * 3 + [c_double] + [c_void_p] * 7 +
                              [c_int, c_void_p]),
    'islam_imu_preint_bwd_scratch_bytes': (c_size_t, [c_int]),
    'islam_imu_preint_bwd': (c_int, [c_void_p] * 4 + [c_int, c_int64, c_double, c_int] + [c_void_p] * 7 + [c_int, c_void_p]),
    'islam_pvgo_default_params': (None, [ctypes.POINTER(PvgoParams)]),
    'islam_pvgo_workspace_bytes': (c_size_t, [c_int]),
    'islam_pvgo_run_chain': (c_int, [c_void_p] * 7 + [c_int, ctypes.POINTER(PvgoParams), c_void_p, c_size_t,
                                                      ctypes.POINTER(PvgoResult), c_void_p, c_int, c_void_p]),
    'islam_pvgo_run_chain_reproj': (c_int, [c_void_p] * 7 + [c_int, ctypes.POINTER(PvgoParams), ctypes.POINTER(PvgoReproj),
                                                             c_void_p, c_size_t, ctypes.POINTER(PvgoResult), c_void_p, c_int,
                                                             c_void_p]),
    'islam_pvgo_reproj_reduce': (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(PvgoReproj), c_void_p, c_void_p]),
    'islam_pvgo_linearize': (c_int, [c_void_p] * 7 + [c_int] + [c_void_p] * 3),
    'islam_pvgo_build_normal': (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(c_double), c_double, c_double] +
                                [c_void_p] * 4),
    'islam_pvgo_solve_chain': (c_int, [c_void_p] * 3 + [c_double, c_int, ctypes.POINTER(c_int), c_void_p, c_size_t,
                                                        c_void_p, c_void_p]),
    'islam_pvgo_solve_chain_enqueue': (c_int, [c_void_p] * 3 + [c_double, c_int, ctypes.POINTER(c_int), c_void_p, c_size_t,
                                                                c_void_p, c_void_p]),
    'islam_pvgo_solve_status': (c_int, [c_int, c_void_p, c_size_t, c_void_p]),
    'islam_pvgo_solve_chain_timed': (c_int, [c_void_p] * 3 + [c_double, c_int, ctypes.POINTER(c_int), c_void_p, c_size_t,
                                                              c_void_p, ctypes.POINTER(c_float), ctypes.POINTER(c_int),
                                                              ctypes.POINTER(c_int), c_void_p]),
    'islam_pvgo_eliminate_level0': (c_int, [c_void_p] * 3 + [c_double, c_int, ctypes.POINTER(c_int), c_void_p, c_size_t,
                                                             c_void_p]),
    'islam_pvgo_trial_elim_burst': (c_int, [c_void_p] * 7 + [c_int, ctypes.POINTER(PvgoParams), c_void_p, c_size_t, c_int,
                                                             ctypes.POINTER(c_float), ctypes.POINTER(c_int), c_void_p]),
    'islam_pvgo_plan': (c_int, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'islam_dist_unique_id': (c_int, [c_void_p]),
    'islam_dist_comm_init': (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    'islam_dist_comm_destroy': (c_int, [c_void_p]),
    'islam_dist_comm_info': (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'islam_pvgo_sharded_scratch_bytes': (c_size_t, [c_int, c_int]),
    'islam_pvgo_run_chain_sharded': (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 7 + [c_int, ctypes.POINTER(PvgoParams),
                                             ctypes.POINTER(PvgoReproj), c_void_p, c_size_t, c_void_p, c_size_t, ctypes.POINTER(PvgoResult),
                                             ctypes.POINTER(ctypes.c_longlong), c_void_p]),
    'islam_pvgo_run_chain_sharded_cb': (c_int, [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 7 + [c_int, ctypes.POINTER(PvgoParams),
                                                ctypes.POINTER(PvgoReproj), c_void_p, c_size_t, c_void_p, c_size_t,
                                                ctypes.POINTER(PvgoResult), ctypes.POINTER(ctypes.c_longlong), c_void_p]),
    'islam_pvgo_shard_ranges': (c_int, [c_int, ctypes.POINTER(c_int), c_int, c_int, ctypes.POINTER(c_int)]),
    'islam_pvgo_shard_upsweep': (c_int, [c_void_p] * 3 + [c_double, c_int, ctypes.POINTER(c_int), c_int, c_int, c_int, c_void_p, c_size_t,
                                         c_void_p, c_void_p, c_void_p]),
    'islam_pvgo_shard_downsweep': (c_int, [c_void_p, c_int, ctypes.POINTER(c_int), c_int, c_int, c_int, c_void_p, c_size_t] + [c_void_p] * 3),
    'islam_pvgo_trial': (c_int, [c_void_p] * 9 + [c_int] + [c_void_p] * 4),
    'islam_pvgo_retract': (c_int, [c_void_p] * 3 + [c_double, c_int] + [c_void_p] * 3),
    'islam_pvgo_linearize_edges': (c_int, [c_void_p] * 3 + [c_int, c_void_p, c_void_p]),
    'islam_pvgo_assemble_dense': (c_int, [c_void_p] * 7 + [c_double, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'islam_pvgo_vo_loss_fwd': (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 4),
    'islam_pvgo_vo_loss_bwd': (c_int, [c_void_p] * 4 + [c_int] + [c_void_p] * 2),
    'islam_pvgo_align': (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 3),
}


class IslamHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libislam_hip error %d: %s' % (code, msg))
        self.code = code


def build(verbose=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ['make', '-C', os.path.join(_HERE, 'csrc'), '-j4']
    if not verbose:
        cmd.insert(1, '-s')
    subprocess.check_call(cmd)


class _StreamArg(c_void_p):
    """hipStream_t argument that remembers which device its stream lives on (see stream_ptr)."""
    device_index = None


def _scoped(fn):
    """Library calls launch on the HIP *current device*.  When the stream handed over (always the last argument) belongs to
    another device, the call runs under ``torch.cuda.device(...)`` and the caller's current device is restored afterwards."""
    def call(*args):
        s = args[-1] if args else None
        idx = getattr(s, 'device_index', None)
        if idx is not None and idx != torch.cuda.current_device():
            with torch.cuda.device(idx):
                return fn(*args)
        return fn(*args)
    call.__name__ = getattr(fn, '__name__', 'islam_fn')
    call.raw = fn
    return call


class _Lib:
    """The loaded shared object: one attribute per symbol of include/islam_hip.h (device-scoped wrappers of the ctypes functions)."""
    def __init__(self, cdll):
        self._cdll = cdll
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, _scoped(fn))

    def __getattr__(self, name):                  # symbols outside SIGNATURES (probe builds)
        return getattr(self._cdll, name)


def lib():
    """Load libislam_hip.so; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('%s not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                              '(islam_amd has no CPU fallback)' % LIB_PATH)
        _lib = _Lib(ctypes.CDLL(LIB_PATH))
    return _lib


def check(rc):
    if rc != 0:
        raise IslamHipError(rc, lib().islam_last_error().decode('utf-8', 'replace'))


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream on ``device`` (0 = default stream).  Kernel launches, hipMemsetAsync and
    hipFuncSetAttribute inside the library act on the HIP *current device*, and a stream of another device would be an invalid
    handle there (run_pvgo(device='cuda:1') without torch.cuda.set_device(1)): the returned handle carries its device index
    and the library call it is passed to runs under a scoped ``torch.cuda.device`` (``_scoped``) -- the caller's current device
    is never changed."""
    s = _StreamArg(torch.cuda.current_stream(device).cuda_stream)
    if device is not None:
        s.device_index = torch.device(device).index
    return s


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('islam_amd kernels need device tensors (got a %s tensor); there is no CPU path' % t.device)
