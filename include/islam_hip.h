/* islam_hip.h -- C ABI of libislam_hip.so: the MI355X (gfx950) implementation of iSLAM's bilevel
 * hot path.
 *
 * The reference (sair-lab/iSLAM @ 2024_10_08) has no FFI of its own: its hot path is Python on
 * PyTorch/PyPose plus four CuPy RawKernels.  Each entry point below therefore names the reference
 * Python/CUDA code it replaces (file:line under /root/reference); INTEGRATION.md shows the ctypes
 * stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (PyTorch tensors' data_ptr()),
 *     contiguous, row-major / NCHW; the library never frees or retains them;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it;
 *     islam_pvgo_run_chain[_reproj]() polls one 128-byte verdict per LM trial in pinned host memory (no stream
 *     synchronisation), islam_pvgo_solve_chain_timed() synchronises the stream; nothing else waits on the device;
 *   - return 0 on success, <0 on error: -1 bad argument, -2 HIP runtime error, -3 non-positive
 *     pivot in the block Cholesky (PyPose: "Linear solver failed. Breaking optimization step"),
 *     -4 unsupported graph topology;  islam_last_error() returns a thread-local message;
 *   - dtype codes: 0 = float32, 1 = float64.
 */
#ifndef ISLAM_HIP_H
#define ISLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISLAM_OK 0
#define ISLAM_EARG (-1)
#define ISLAM_EHIP (-2)
#define ISLAM_ENOTPD (-3)
#define ISLAM_ETOPO (-4)
#define ISLAM_F32 0
#define ISLAM_F64 1

const char* islam_last_error(void);
int islam_abi_version(void);

/* Measurement aid (no reference counterpart; bench.py's stereo_vio diagnostics): one wavefront runs `iters` dependent fp64 FMAs on
 * `stream` and writes out3[0] = elapsed ticks of the constant-rate wall clock (islam_wall_clock_khz), out3[1] = elapsed shader
 * cycles; shader MHz = out3[1] / out3[0] * wall kHz / 1e3.  Launched on a stream of its own it samples the clock the chip sustains
 * while other streams keep it busy.  out3: 3 x int64 in device memory. */
int islam_clock_probe(long long* out3, int iters, void* stream);
int islam_wall_clock_khz(int device);
/* Measurement aid (no reference counterpart; bench.py's roofline.latency_model, SURVEY 8(d) "t >= n_launch * t_launch + depth * t_block"):
 * `n` back-to-back launches of a kernel that does nothing (grid x block threads) on `stream` between one pair of events;
 * *us_per_launch = the period of a kernel in a chain of dependent launches on this platform.  Synchronises the stream. */
int islam_launch_cost_probe(int grid, int block, int n, float* us_per_launch, void* stream);

/* ---------------------------------------------------------------- PWC-Net front-end kernels */

/* 81-channel local correlation, forward.
 * Replaces Network/PWC/correlation.py:281-331 (_FunctionCorrelation.forward) and its two CUDA
 * kernels kernel_Correlation_rearrange (:8-33) + kernel_Correlation_updateOutput (:35-103):
 *   out[b,(dy+4)*9+(dx+4),y,x] = (1/C) sum_c f1[b,c,y,x] * f2[b,c,y+dy,x+dx],  dy,dx in [-4,4], zero pad.
 * f1,f2: (B,C,H,W) float32; out: (B,81,H,W) float32.
 * scratch: islam_corr81_scratch_bytes(B,C,H,W) bytes or NULL.  Small pyramid levels have too few tiles to fill the chip;
 * with scratch the channels are split over workgroups and summed in a fixed order (deterministic). */
size_t islam_corr81_scratch_bytes(int B, int C, int H, int W);
int islam_corr81_fwd(const float* f1, const float* f2, float* out, int B, int C, int H, int W, void* scratch, void* stream);
/* The same correlation followed by LeakyReLU(slope), written into channels [ooff, ooff + 81) of out = (B,otot,H,W): PWC-Net's
 * `corr = self.leakyRELU(corr)` and the torch.cat that follows (Network/PWC/PWCNet.py:225-227, 255-258) without the two extra passes. */
int islam_corr81_fwd_act(const float* f1, const float* f2, float* out, int otot, int ooff, float slope, int B, int C, int H, int W, void* scratch,
                         void* stream);

/* Correlation backward.  Replaces correlation.py:334-383 with kernels updateGradFirst (:105-167) and
 * updateGradSecond (:169-233).  g1 and/or g2 may be NULL (needs_input_grad false). */
int islam_corr81_bwd(const float* f1, const float* f2, const float* gout, float* g1, float* g2,
                     int B, int C, int H, int W, void* stream);

/* Backward warp + validity mask.  Replaces Network/PWC/PWCNet.py:170-206 (PWCDCNet.warp):
 * out = grid_sample(x, grid + flow*scale, bilinear, zeros, align_corners=True) * (grid_sample(1,..) >= 0.9999).
 * x,out: (B,C,H,W); flow: (B,2,H,W) float32.  `scale` is the per-level factor of PWCNet.py:259-268. */
int islam_warp_mask(const float* x, const float* flow, float scale, float* out, int B, int C, int H, int W,
                    void* stream);

/* Backward of islam_warp_mask (autograd of PWCNet.py:195-206; the mask is piecewise constant).
 * gx (B,C,H,W) and gflow (B,2,H,W) must be ZERO-INITIALISED by the caller (atomic scatter). */
int islam_warp_mask_bwd(const float* x, const float* flow, float scale, const float* gout, float* gx, float* gflow,
                        int B, int C, int H, int W, void* stream);
/* ConvTranspose2d(C, 2, kernel 4, stride 2, padding 1) + bias, fp32 NCHW: PWC-Net's `deconv%d` / `upfeat%d`
 * (Network/PWC/PWCNet.py:55-56 `deconv()`, layers :113-114 ff., uses :259-268).  x: (B,C,H,W); w: (C,2,4,4) as nn.ConvTranspose2d stores it;
 * the two output channels go to channels [coff, coff + 2) of y = (B,ytot,2H,2W).  Exact fp32 FMAs; channel sums in a fixed order. */
int islam_deconv4x4s2_to2_f32(const float* x, const float* w, const float* bias, float* y, int ytot, int coff, int B, int C, int H, int W,
                              void* stream);
/* PWC-Net's flow head + up-sampled features of one level in ONE pass over the level's DenseNet concatenation:
 *   flow = Conv2d(C, 2, kernel 3, padding 1)(x) + bf          (`predict_flow%d`, Network/PWC/PWCNet.py:112,122,132,142,152; no activation)
 *   up   = ConvTranspose2d(C, 2, 4, 2, 1)(x) + bu              (`upfeat%d`, as islam_deconv4x4s2_to2_f32), into channels [upoff, upoff+2) of
 *                                                               up = (B,uptot,2H,2W); wu == NULL: the head alone (level 2)
 * x: (B,C,H,W) fp32; wf: [C][2][3][3] fp32 (the Conv2d weight with its first two axes swapped); flow: (B,2,H,W).  Exact fp32 FMAs
 * (the reference's arithmetic; the matrix-core path rounded the head's operands to bf16), fixed summation order. */
int islam_flow_head_up_f32(const float* x, const float* wf, const float* bf, float* flow, const float* wu, const float* bu, float* up, int uptot,
                           int upoff, int B, int C, int H, int W, void* stream);
/* One level of PWC-Net's feature pyramid -- conv(k3, stride 2) + conv(k3) + conv(k3), each + bias + LeakyReLU(slope) -- in one
 * launch (Network/PWC/PWCNet.py:78-83 conv1a/conv1aa/conv1b, conv2a/conv2aa/conv2b; :16-20 `conv()`; used at :240-243).  bf16 operands
 * (round to nearest even), fp32 accumulation, fp32 bias / activation; intermediates stay in LDS.  x: (B,Cin,H,W) fp32 NCHW;
 * y: (B,C,(H-1)/2+1,(W-1)/2+1) fp32 NCHW; wA/wB/wC: bf16 [C][ceil(9*SC/32)*32], K index = (ky*3+kx)*SC + c, SC = 4 for Cin <= 4 else
 * Cin (wB, wC: SC = C), zero padded (islam_pyramid_packed_elems elements each; islam_amd/ops.py pack_pyramid_weight).
 * Built for (Cin, C) = (3, 16) and (16, 32); anything else is ISLAM_EARG. */
size_t islam_pyramid_packed_elems(int Cin, int Cout);
int islam_flow_pyramid_level(const float* x, const uint16_t* wA, const float* bA, const uint16_t* wB, const float* bB, const uint16_t* wC,
                             const float* bC, float* y, int B, int Cin, int H, int W, int C, float slope, void* stream);

/* Level 1 of the pyramid on the two frames of a pair tensor x (B, 6, H, W) fp32 = [frame 1 | frame 2] along the channels -- the input
 * PWCDCNet.forward splits (Network/PWC/PWCNet.py:224-226): y (2 B, 16, H/2, W/2), first frames first, as islam_flow_pyramid_level
 * gives for torch.cat((x[:, :3], x[:, 3:]), 0), without that copy. */
int islam_flow_pyramid_level_pair(const float* x, const uint16_t* wA, const float* bA, const uint16_t* wB, const float* bB, const uint16_t* wC,
                                  const float* bC, float* y, int B, int H, int W, float slope, void* stream);

/* 3x3 convolution (+ bias + LeakyReLU) of the frozen flow network on the matrix cores (implicit GEMM, bf16 operands,
 * fp32 accumulate).  Replaces cuDNN under Network/PWC/PWCNet.py:16-20 `conv()` (Conv2d k=3, padding = dilation, then
 * LeakyReLU(0.1)) for inference:  y[b, coff+n, ho, wo] = act(bias[n] + sum w[n,c,r,s] x[b, c, ho*stride + (r-1)*dil,
 * wo*stride + (s-1)*dil]),  act(v) = v >= 0 ? v : slope*v  (slope = 1: no activation).
 * x: channels [xoff, xoff+Cin) of a (B,xtot,H,W) fp32 NCHW buffer; y: channels [coff, coff+Cout) of a (B,ytot,Ho,Wo) fp32
 * NCHW buffer, Ho = (H-1)/stride+1 -- the DenseNet-style torch.cat of PWCNet.py:237-292 without a copy; bias (Cout) or NULL.
 * wpacked: bf16 bits, [9][CoutP][CinP] (tap = 3r+s; CoutP = Cout rounded up to 64, CinP = Cin rounded up to 16, zero padded),
 * islam_conv3x3_packed_elems(Cin, Cout) elements.  When Cin > 16 is not a multiple of 16 the LAST group of 16 channel slots holds
 * channels Cin-16 .. Cin-1 (the kernel's last chunk reads the last 16 channels of x), with zeros in the slots of channels the
 * previous group already holds; Cin <= 16 or a multiple of 16: slot c = channel c.  stride 1 or 2, dilation 1..16. */
size_t islam_conv3x3_packed_elems(int Cin, int Cout);
int islam_conv3x3_mfma(const float* x, const uint16_t* wpacked, const float* bias, float* y, int B, int Cin, int H, int W,
                       int Cout, int stride, int dilation, int xoff, int xtot, int coff, int ytot, float slope, void* stream);

/* Bilinear resize of a channels-last (NHWC) bf16 tensor: the F.upsample / F.interpolate calls of the frozen stereo network
 * (Network/PSM/submodule.py:124-155 SPP branches, Network/StereoNet7.py:100-146), ATen's upsample_bilinear2d arithmetic.
 * x (B,Hi,Wi,C), y (B,Ho,Wo,C) bf16 bits, C a multiple of 8; align_corners as in torch. */
int islam_resize_bilinear_nhwc_bf16(const uint16_t* x, uint16_t* y, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                    int align_corners, void* stream);
/* The same, written into channels [yoff, yoff+C) of a (B,Ho,Wo,ytot) tensor (ytot, yoff multiples of 8): bilinear resizing is
 * per channel, so the up-sampled concatenation of submodule.py:140-152 is assembled in place, piece by piece, without the
 * intermediate torch.cat and without copying the up-sampled tensor into the next concatenation. */
int islam_resize_bilinear_nhwc_bf16_into(const uint16_t* x, uint16_t* y, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                         int align_corners, int ytot, int yoff, void* stream);
/* torch.cat((F.interpolate(torch.cat(pieces, 1), (Ho, Wo), mode='bilinear'), tail), 1) in one launch: the `bigger` feature extractor of
 * StereoNet7 (Network/StereoNet7.py:36-46 -> Network/PSM/submodule.py:139-152 with the extra up-sampling and layer1's output joined).
 * srcs / chans: HOST arrays of n <= 8 device pointers to channels-last bf16 pieces (B, Hi, Wi, chans[k]) / their channel counts
 * (multiples of 8); tail (may be NULL with tailC = 0): (B, Ho, Wo, tailC); y: (B, Ho, Wo, sum(chans) + tailC).  Every sample is
 * islam_resize_bilinear_nhwc_bf16's arithmetic; whole pixels are written contiguously. */
int islam_upsample_cat_nhwc_bf16(const uint16_t* const* srcs, const int* chans, int n, const uint16_t* tail, int tailC, uint16_t* y,
                                 int B, int Hi, int Wi, int Ho, int Wo, int align_corners, void* stream);
/* The stereo pair as the feature extractor's batch (Network/StereoNet7.py:95-97 feeds left and right images through one
 * feature_extraction): x (B, H, W, C2) channels-last bf16 with the left image in channels [0, C2/2) and the right one behind it;
 * y (2B, H, W, 8): images [0, B) left, [B, 2B) right, channels [C2/2, 8) zero -- the input of the 3 -> 32 stride-2 first layer
 * (Network/PSM/submodule.py:63) on islam_conv_nhwc_bf16_s2 with its weights zero-padded to 8 input channels. */
int islam_stack_pair_pad8_nhwc_bf16(const uint16_t* x, uint16_t* y, int B, int C2, int H, int W, void* stream);
/* The same stacked batch AND the concatenated pair from the two fp32 NCHW images (B,c,H,W), c <= 4, in one pass: x6 (B,H,W,2c) bf16 =
 * torch.cat((left, right), 1).to(bfloat16) channels-last (Network/VONet.py:31-34), xs (2B,H,W,8) as above.  Round-to-nearest-even. */
int islam_stereo_pair_prepare_f32(const float* left, const float* right, uint16_t* x6, uint16_t* xs, int B, int c, int H, int W, void* stream);
/* y = add + resize(x) in one pass (hourglass.py:60-69 `up1 + up2(low3)`): the up-sampled value is rounded to bf16 before the add,
 * like the two separate ops.  add, y: (B,Ho,Wo,C). */
int islam_resize_bilinear_add_nhwc_bf16(const uint16_t* x, const uint16_t* add, uint16_t* y, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                        int align_corners, void* stream);
/* The same into channels [yoff, yoff+C) of y = (B,Ho,Wo,ytot): the hourglass's half of the decoder's concatenations
 * (Network/StereoNet7.py:129-138 `x = self.conv_c8(x); x = torch.cat((x, cat2), dim=1)` ...) is written in place. */
int islam_resize_bilinear_add_nhwc_bf16_into(const uint16_t* x, const uint16_t* add, uint16_t* y, int B, int C, int Hi, int Wi, int Ho,
                                             int Wo, int align_corners, int ytot, int yoff, void* stream);
/* MaxPool2d(2, 2) / F.max_pool2d(kernel_size=2) of a channels-last bf16 tensor (hourglass.py:52, StereoNet7.py:117-125), relu != 0:
 * of relu(x) (the two commute); (B,H,W,C) -> (B,H/2,W/2,C), C a multiple of 8. */
int islam_maxpool2_nhwc_bf16(const uint16_t* x, uint16_t* y, int B, int C, int H, int W, int relu, void* stream);

/* F.interpolate(x, scale_factor=0.5, mode='bilinear') of a channels-last bf16 image with C <= 8 (even) channels -- the half-resolution
 * stereo pair of Network/StereoNet7.py:101 -- written with 8 - C zero channels behind it into channels [yoff, yoff + 8) of a
 * (B,H/2,W/2,ytot) bf16 tensor (the padded tail of conv_c0's concatenated input): one launch instead of ATen's up-sampling kernel, a
 * slice copy and a zero fill; the same values as ATen bit for bit.  H, W even; ytot, yoff multiples of 8. */
int islam_half_image_into_nhwc_bf16(const uint16_t* x, uint16_t* y, int ytot, int yoff, int B, int C, int H, int W, void* stream);
/* AvgPool2d((k,k), stride=(k,k)) of a channels-last bf16 tensor (the SPP branches, submodule.py:103-122): fp32 accumulation, one
 * rounding; (B,H,W,C) -> (B,H/k,W/k,C). */
int islam_avgpool_nhwc_bf16(const uint16_t* x, uint16_t* y, int B, int C, int H, int W, int k, void* stream);

/* In-place epilogue of a bias-free convolution on a channels-last bf16 tensor: y <- act(bf16(y + bias[c]) [+ res]),
 * act = ReLU (relu != 0) or identity; res NULL or a tensor of y's shape.  One pass for the bias add, activation and residual
 * add around the convolutions of Network/PSM/hourglass.py:6-40 (Residual) and Network/StereoNet7.py:100-146.
 * y, res: (pixels, C) bf16 bits, C a multiple of 8; bias (C) fp32. */
int islam_bias_act_add_nhwc_bf16(uint16_t* y, const float* bias, const uint16_t* res, long long pixels, int C, int relu,
                                 void* stream);
/* The elementwise tail of the trainable pose head's convolutions (Network/VOFlowNet.py:42-157: conv + bias + ReLU, residual
 * blocks conv + bias + shortcut + ReLU), fp32 channels-last (pixels x C, C a multiple of 4), one launch each way:
 *   forward   y = act(x + bias[c] (+ res));  x may alias y
 *   backward  gx = gy * (y > 0) (relu) or gy;  gbias[c] = sum over pixels of gx (deterministic two-stage sum); gx may alias gy.
 * scratch: islam_bias_act_bwd_scratch_floats(pixels, C) floats; ticket: one zero-initialised word, left at zero. */
int islam_bias_act_f32_nhwc(const float* x, const float* bias, const float* res, float* y, long long pixels, int C, int relu,
                            void* stream);
long long islam_bias_act_bwd_scratch_floats(long long pixels, int C);
int islam_bias_act_bwd_f32_nhwc(const float* gy, const float* y, float* gx, float* gbias, float* scratch, unsigned* ticket,
                                long long pixels, int C, int relu, void* stream);

/* The trainable pose head, whole: VOFlowRes.forward_ (Network/VOFlowNet.py:185-194) over the feature embedding of config 1 (:110-157: three
 * plain convolutions, five stages of BasicBlocks :20-39 of 3/4/6/7/3 blocks at 64/128/128/256/256 channels, first block of a stage with
 * stride 2 and a 1x1 stride-2 convolution on its shortcut) and the two Linear heads (:84-92), and its backward (what
 * train.py:280-283's loss.backward() runs through this module) -- each ONE call that enqueues hand-written fp32 kernels (exact-fp32
 * matrix-core implicit GEMMs, csrc/pose_head.hip) on `stream`: no MIOpen / CK / ATen launch, deterministic summation orders.
 *   x        (B,H,W,4) fp32 channels-last: [flow(2), intrinsic layer(2)] (Network/VONet.py:36); B <= 16; the input's 1/64-size feature map must
 *            have 6 pixels (448x640 images: H = 112, W = 160)
 *   params   the module's 120 parameters in state_dict order (feat_net.0.0.weight, feat_net.0.0.bias, ..., voflow_rot.2.bias), device pointers;
 *            convolution weights in channels_last memory ([Cout][ky][kx][Cin]), everything else contiguous
 *   out6     (B,6) = cat(trans, rot)
 *   workspace  islam_pose_head_workspace_bytes(B,H,W) bytes (0: unsupported shape); holds every activation of the forward -- the backward
 *            reads what the LAST forward on this workspace left there -- plus gradient ping-pong buffers and split-K scratch.  Must be
 *            zero-filled once before its first use (ticket words; every launch leaves them zero).
 * backward: grads[i] receives (accumulate = 0) or is incremented by (accumulate = 1) the gradient of parameter i, same layouts as params;
 * grad_out6 (B,6).  No gradient w.r.t. x is produced (the reference detaches the flow in front of the head's consumers only when the flow
 * net is frozen, TartanVO.py:109; a trainable flow net takes the eager autograd path in islam_amd/nets.py). */
size_t islam_pose_head_workspace_bytes(int B, int H, int W);
int islam_pose_head_forward(const float* x, const float* const* params, float* out6, void* workspace, size_t workspace_bytes, int B, int H,
                            int W, void* stream);
int islam_pose_head_backward(const float* x, const float* const* params, float* const* grads, const float* grad_out6, void* workspace,
                             size_t workspace_bytes, int B, int H, int W, int accumulate, void* stream);

/* Train-mode BatchNorm2d (+ ReLU, + residual add) on a channels-last bf16 tensor -- the BatchNorm layers of the "frozen"
 * stereo feature extractor, which the reference still runs with batch statistics (TartanVO.py:90-91; Network/PSM/
 * submodule.py:10-43):  y = act(bf16(x*scale[c] + shift[c]) [+ res]), scale = weight*rsqrt(var_biased + eps),
 * shift = bias - mean*scale; running_mean / running_var (unbiased) / num_batches_tracked updated as nn.BatchNorm2d does
 * (pass NULL to skip).  x, y, res: (pixels, C) bf16 bits (y may alias x), C = 8 * (a divisor of 256), at most 256; weight, bias, running_*:
 * fp32 (C); scratch: islam_bn_scratch_floats(C) floats.  Deterministic (fixed-order reduction). */
size_t islam_bn_scratch_floats(int C);
int islam_bn_train_nhwc_bf16(const uint16_t* x, uint16_t* y, const uint16_t* res, const float* weight, const float* bias,
                             float* running_mean, float* running_var, long long* num_batches_tracked, double momentum,
                             double eps, int relu, long long pixels, int C, float* scratch, void* stream);

/* Channels-last bf16 convolution of the frozen stereo feature extractor on the matrix cores (implicit GEMM, fp32 accumulate,
 * round-to-nearest-even output), with the surrounding train-mode BatchNorm folded into its two ends.  Replaces cuDNN under
 * Network/PSM/submodule.py:10-13 `convbn` (Conv2d k=3 p=1 / k=1 p=0, stride 1, bias=False, then BatchNorm2d), :66-155.
 *   y[b,ho,wo,n] = epi( sum w[n,c,r,s] * pre(x[b, ho+r-P, wo+s-P, c]) ),  P = ksize/2, zero padding, output size = input size
 *   pre(v) = v  (in_affine NULL)  or  relu(bf16(v*in_affine[c] + in_affine[Cin+c]))  -- the producer's BatchNorm + ReLU on load
 *   epi(a) = bf16(a), and with `stats` the per-channel sums of bf16(a) and bf16(a)^2 over all pixels (stats NULL: skipped);
 *            or  act( bf16( bf16(a + bias[n]) [+ res] ) ),  act = ReLU if `relu` & 1;  `relu` & 2: pre(v) = relu(v) (in_affine NULL).
 * x (B,H,W,Cin), res / y (B,H,W,Cout) bf16 bits, Cin and Cout multiples of 8; wpacked: bf16 [ksize*ksize][CoutP][CinP], CoutP =
 * Cout rounded up to 64, CinP = Cin rounded up to 32, zero padded (islam_conv_nhwc_packed_elems elements).
 * stats: islam_conv_nhwc_stats_floats(B,H,W,Cout) floats; on return its LAST 256*2*Cout floats hold the folded partial sums
 * ([256][2][Cout], fixed-order = deterministic) that islam_bn_finalize reads.  ksize 1 or 3. */
size_t islam_conv_nhwc_packed_elems(int Cin, int Cout, int ksize);
int islam_conv_nhwc_stat_blocks(int B, int H, int W, int Cout);
size_t islam_conv_nhwc_stats_floats(int B, int H, int W, int Cout);
int islam_conv_nhwc_bf16(const uint16_t* x, const uint16_t* wpacked, const float* in_affine, const float* bias, const uint16_t* res,
                         uint16_t* y, float* stats, int B, int Cin, int H, int W, int Cout, int ksize, int relu, void* stream);
/* islam_conv_nhwc_bf16 (no residual, no statistics) with the result written into channels [yoff, yoff + Cout) of a (B,H,W,ytot) bf16
 * tensor: straight into a concatenation under construction (Network/StereoNet7.py:103-105) instead of a dense tensor torch.cat copies.
 * ytot, yoff: multiples of 8.  `relu` as in islam_conv_nhwc_bf16. */
int islam_conv_nhwc_bf16_into(const uint16_t* x, const uint16_t* wpacked, const float* in_affine, const float* bias, uint16_t* y, int ytot,
                              int yoff, int B, int Cin, int H, int W, int Cout, int ksize, int relu, void* stream);

/* Conv2d(Cin, Cout, k, stride = 2, padding = k / 2) on the same kernel: layer2's stride-2 3x3 convbn and its stride-2 1x1 downsample
 * (Network/PSM/submodule.py:24-26, 76-85) and the 2x2 stride-2 form of StereoNet7's last transposed convolution at the pixels VONet keeps
 * (Network/StereoNet7.py:88-90, Network/VONet.py:33-34).  x (B,Hi,Wi,Cin) -> y (B,Ho,Wo,Cout), Ho <= (Hi + 2 (k/2) - k) / 2 + 1 (fewer rows /
 * columns may be asked for); k = 1, 2, 3; the other arguments as islam_conv_nhwc_bf16, `stats` sized by islam_conv_nhwc_s2_stats_floats
 * (its last 256*2*Cout floats are the folded sums). */
size_t islam_conv_nhwc_s2_stats_floats(int B, int Ho, int Wo, int Cout);
int islam_conv_nhwc_bf16_s2(const uint16_t* x, const uint16_t* wpacked, const float* in_affine, const float* bias, const uint16_t* res,
                            uint16_t* y, float* stats, int B, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int ksize, int relu,
                            void* stream);

/* `convbn` in training mode up to the BatchNorm's [scale | shift] (Network/PSM/submodule.py:10-13): islam_conv_nhwc_bf16 with `stats`
 * followed by islam_bn_finalize as one call, bit for bit the same results (y raw convolution output, scale_shift 2*Cout floats, running
 * statistics updated like nn.BatchNorm2d; count = B*H*W).  Two launches behind the persistent kernels (their few rows of partial sums go
 * straight into the finalize), three behind the tile kernel.  counter: one int of device memory that is zero before the call and is left
 * zero (used by the ticketed variant only, ISLAM_BN_FINALIZE=1); calls that may run concurrently on different streams must not share it. */
int islam_conv_nhwc_bf16_bn(const uint16_t* x, const uint16_t* wpacked, const float* in_affine, uint16_t* y, float* stats, int B, int Cin,
                            int H, int W, int Cout, int ksize, int in_relu, const float* weight, const float* bias, float* running_mean,
                            float* running_var, long long* num_batches_tracked, double momentum, double eps, float* scale_shift,
                            int* counter, void* stream);
/* Which kernel serves the 128 -> 128, 64 -> 128 and 32 -> 32 3x3 layers (Network/PSM/submodule.py:66-155 feature_extraction: layer3 /
 * layer4, twelve per forward; firstconv / layer1, eight per forward) behind islam_conv_nhwc_bf16 / _into / _bn: 0 = the tile kernel, 1 (default;
 * ISLAM_CONV_WS presets it) = the persistent kernels of csrc/conv_ws.hip (weights in the register file, 32 x 4-pixel tiles) and
 * csrc/conv_ws32.hip (32 x 16-pixel tiles) when the image is whole tiles and has at least 1024 of them, 2 = the persistent kernels on
 * every whole-tile layer (tests, A/B runs).  Bit-identical outputs either way.  Returns the previous mode; any other argument only queries. */
int islam_conv_ws_mode(int mode);
/* Launches of the two persistent kernels since the library was loaded (host-side counters; a launch recorded into a HIP graph counts once,
 * at capture): out2[0] = conv3x3_ws_kernel (conv_ws.hip), out2[1] = conv3x3_ws32_kernel (conv_ws32.hip).  The parity tests of the stereo
 * feature extractor (Network/PSM/submodule.py:66-155) use it to prove which kernel produced the output they compare. */
int islam_conv_ws_launch_counts(long long* out2);
/* 3x3 stride-1 convolution of the flow net's DenseNet blocks (Network/PWC/PWCNet.py:16-20 `conv()` = Conv2d + LeakyReLU(0.1),
 * :237-292 the blocks) on the channels-last kernel.  x: bf16 channels [xoff, xoff + Cin) of a (B,H,W,xtot) MIRROR of the block's
 * concatenation buffer; the result goes as fp32 NCHW into channels [coff, coff + Cout) of y32 (B,ytot,H,W) -- what correlation /
 * warp / transposed convolutions / flow heads read -- and, when ymir != NULL, as bf16 into channels [moff, moff + Cout) of the
 * (B,H,W,mtot) mirror for the next convolution.  Same arithmetic as islam_conv3x3_mfma (operands rounded to bf16 nearest-even,
 * fp32 accumulation, bias, LeakyReLU(slope); slope 1: none).  wpacked: islam_conv_nhwc_packed_elems(Cin, Cout, 3) elements with
 * zero rows for padded input channels.  Cin, Cout, xtot, xoff, mtot, moff: multiples of 8.  y32 or ymir may be NULL (not both).
 * dilation d > 1 (the context layers dc_conv2-5): evaluated as d*d dense convolutions on the sub-grids (a::d, c::d) of the image;
 * H and W must be multiples of d. */
int islam_conv_nhwc_flow(const uint16_t* x, int xtot, int xoff, int Cin, const uint16_t* wpacked, const float* bias, float* y32, int ytot,
                         int coff, uint16_t* ymir, int mtot, int moff, int B, int H, int W, int Cout, int dilation, float slope, void* stream);
/* ConvTranspose2d(Cin, Cout, kernel 4, stride 2, padding 1) + bias (+ ReLU, relu = 1) of the frozen stereo net's decoder --
 * Network/StereoNet7.py:78-90 (deconv_c7_2, deconv_c7 ... deconv_c10) with the activation and concatenation of :121-136 -- on the
 * channels-last matrix-core kernel: four dense 2x2 convolutions, one per output parity class (a, c) of pixel (2y + a, 2x + c), taps
 * K[r][s] = W[:, :, 3 - 2r - a, 3 - 2s - c].  x: (B,H,W,Cin) bf16; wpacked: islam_deconv_nhwc_packed_elems(Cin, Cout) bf16 elements,
 * [class a*2+c][tap r*2+s][CoutP][CinP]; the result goes to channels [yoff, yoff + Cout) of y = (B,2H,2W,ytot) bf16 (the channel
 * slice of the concatenation the decoder builds next).  bf16 operands, fp32 accumulation, output rounded to nearest even.
 * Cin, Cout, ytot, yoff: multiples of 8. */
size_t islam_deconv_nhwc_packed_elems(int Cin, int Cout);
int islam_deconv4x4s2_nhwc_bf16(const uint16_t* x, const uint16_t* wpacked, const float* bias, uint16_t* y, int ytot, int yoff, int B, int Cin,
                                int H, int W, int Cout, int relu, void* stream);
/* The transposed convolution above / islam_conv_nhwc_bf16_s2 (kernel size 2) on torch.cat((x1, x2), 1) of two dense channels-last tensors
 * read where they lie (Network/StereoNet7.py:121-138: the decoder concatenates the previous stage's result with a skip tensor in front of
 * every transposed convolution): C1 a multiple of 32 (16 for _s2_cat), C2 of 8, wpacked as for Cin = C1 + C2; bit-identical. */
int islam_deconv4x4s2_nhwc_bf16_cat(const uint16_t* x1, int C1, const uint16_t* x2, int C2, const uint16_t* wpacked, const float* bias, uint16_t* y,
                                    int ytot, int yoff, int B, int H, int W, int Cout, int relu, void* stream);
int islam_conv_nhwc_bf16_s2_cat(const uint16_t* x1, int C1, const uint16_t* x2, int C2, const uint16_t* wpacked, const float* bias, uint16_t* y, int B,
                                int Hi, int Wi, int Cout, int Ho, int Wo, int ksize, int relu, void* stream);
/* One Residual module of the stereo net's hourglass stacks in ONE launch -- Network/PSM/hourglass.py:28-52 (`Residual.forward`),
 * instantiated by Network/PSM/hourglass.py:53-77 / Network/StereoNet7.py:56-90:
 *   y = conv3(relu(conv2(relu(conv1(relu(x)))))) + res ;  conv1 1x1 Cin -> h, conv2 3x3 h -> h (padding 1), conv3 1x1 h -> Cout,
 *   h = Cout / 2, every convolution with a bias; res = x when Cin == Cout, else skip_layer(x) computed by the caller
 *   (islam_conv_nhwc_bf16, 1x1 of the raw x).  The two h-channel intermediates never leave LDS; rounding points as the layer-by-layer
 *   path: t1 = relu(bf16(. + b1)), t2 = relu(bf16(. + b2)), y = bf16(bf16(. + b3) + res); bf16 operands, fp32 accumulation.
 * x (B,H,W,Cin), res / y (B,H,W,Cout) bf16 bits; Cin, Cout multiples of 64 up to 256 (Cout = 64 with Cin = 64 only: res must be x).
 * wpacked: islam_hg_residual_packed_elems(Cin, Cout) bf16 elements = the weights as MFMA A fragments (64 lanes x 8 bf16 each) in
 * the order the waves consume them (layout: csrc/hourglass.hip; built by islam_amd/ops.py pack_hg_residual); bias: fp32
 * [b1 (h) | b2 (h) | b3 (Cout)]. */
size_t islam_hg_residual_packed_elems(int Cin, int Cout);
int islam_hg_residual_nhwc_bf16(const uint16_t* x, const uint16_t* res, uint16_t* y, const uint16_t* wpacked, const float* bias, int B, int Cin,
                                int H, int W, int Cout, void* stream);
/* fp32 NCHW channels [soff, soff + C) of src (B,stot,H,W) -> bf16 (nearest-even) channels [doff, doff + C) of dst (B,H,W,dtot);
 * channels up to the next multiple of 8 are zeroed.  Fills the mirror with what the non-convolution producers wrote. */
int islam_nchw_f32_to_nhwc_bf16(const float* src, int stot, int soff, uint16_t* dst, int dtot, int doff, int B, int C, int H, int W,
                                void* stream);
/* The two halves of islam_bn_train_nhwc_bf16 for a producer that delivers the statistics itself: folded = [256][2][C] partial
 * sums, count = pixels; scale_shift (2*C floats) = [weight*rsqrt(var+eps) | bias - mean*scale]; running statistics updated like
 * nn.BatchNorm2d (NULL: skipped).  C <= 256.  Apply: y = act( bf16(x*scale[c] + shift[c]) [+ res] ), C a multiple of 8. */
int islam_bn_finalize(const float* folded, double count, const float* weight, const float* bias, float* running_mean,
                      float* running_var, long long* num_batches_tracked, double momentum, double eps, int C, float* scale_shift,
                      void* stream);
int islam_bn_apply_nhwc_bf16(const uint16_t* x, uint16_t* y, const uint16_t* res, const float* scale_shift, int relu,
                             long long pixels, int C, void* stream);

/* ---------------------------------------------------------------- edge mask */

/* Edge mask of the front-end, whole batch in one launch, no host round trip.
 * Replaces TartanVO.py:145-155: img0.cpu() (27.5 MB D2H at B=8) -> (img*255).astype(uint8) -> per image
 * cv2.resize(fx=fy=1/4) -> cv2.Canny(im, 50, 100) -> cv2.dilate(5x5 ones) -> `> 0` -> .cuda().
 * img (B,3,H,W) float32 in [0,1]; mask (B,h,w) uint8 in {0,1} with h,w = H/4,W/4 (downscale != 0; H,W multiples of 4)
 * or H,W.  low/high: Canny thresholds (50, 100).  h*w <= islam_edge_mask_max_pixels(): the quarter-resolution image lives
 * in the LDS of one CU (112x160 for the 448x640 input); larger -> ISLAM_EARG. */
int islam_edge_mask_max_pixels(void);
int islam_edge_mask(const float* img, uint8_t* mask, int B, int H, int W, int downscale, int low, int high, void* stream);

/* ---------------------------------------------------------------- stereo scale recovery */

/* Replaces the per-sample Python loop TartanVO.py:159-167 around dense_ba.py:88-176
 * (scale_from_disp_flow, disparity branch) for a whole batch in one launch.
 * disp (B,1,H,W), flow (B,2,H,W) in pixels at 1/4 res; pose7 (B,7) = the ENU motion [t,q];
 * intr4 (B,4) = fx,fy,cx,cy at 1/4 res; baseline (B); edge (B,H,W) uint8 or NULL; disp_th (B).
 * Outputs: scale (B) = (M^T w)/(M^T M); z (B,H,W); mask, dmask (B,H,W) uint8;
 * sums (B,18) float64: [0] M^T M, [1] M^T w, [2..16] the first-order sums the backward pass of
 * `scale` w.r.t. the pose needs (layout in scale_ls.hip), [17] number of masked pixels;
 * partial: scratch of B*ISLAM_SCALE_NBLK*18 doubles. */
#define ISLAM_SCALE_NBLK 16
#define ISLAM_SCALE_NSUM 18
int islam_scale_ls(const float* disp, const float* flow, const float* pose7, const float* intr4,
                   const float* baseline, const uint8_t* edge, const float* disp_th,
                   float* scale, float* z, uint8_t* mask, uint8_t* dmask, double* sums, double* partial,
                   int B, int H, int W, void* stream);
/* Same with a depth map in place of the disparity (dense_ba.py:125-131, `depth=` argument of scale_from_disp_flow):
 * depth_mask = 0 < depth <= fx*baseline, z = depth there. */
int islam_scale_ls_depth(const float* depth, const float* flow, const float* pose7, const float* intr4,
                         const float* baseline, const uint8_t* edge, float* scale, float* z, uint8_t* mask,
                         uint8_t* dmask, double* sums, double* partial, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------- IMU pre-integration */

/* Replaces the frame loop of imu_integrator.py:116-158 and pp.module.IMUPreintegrator.forward
 * (integrate + predict; covariance propagation is discarded by the reference and not computed).
 * dt (S), gyro (S,3), acc (S,3): the batch slice (already bias-corrected / denoised);
 * seg (nframes+1) int64: sample offset of each frame boundary inside the slice;
 * init_pos(3), init_rot(4), init_vel(3); motion_mode as imu_integrator.py:69-80.
 * max_frame_samples = max_i (seg[i+1]-seg[i]) (known on the host from rgb2imu_sync).
 * Outputs: world mode nframes+1 rows (row 0 = init), motion mode nframes rows.
 * scratch: at least islam_imu_scratch_bytes(S, nframes, dtype) bytes.  Calls that may run CONCURRENTLY (different streams) must not share
 * a scratch buffer: it holds the hand-off counter between the two workgroups of the world-row kernel.  If that hand-off ever times out
 * (bounded wait), out_pos / out_vel are filled with NaN rather than left half-written. */
size_t islam_imu_scratch_bytes(int64_t S, int nframes, int dtype);
int islam_imu_preint(const void* dt, const void* gyro, const void* acc, const int64_t* seg, int nframes,
                     int64_t S, int max_frame_samples, const void* init_pos, const void* init_rot,
                     const void* init_vel, double gravity, int motion_mode, void* out_pos, void* out_rot,
                     void* out_vel, void* scratch, int dtype, void* stream);

/* Both call forms of IMUModule.integrate on one frame range from ONE pass (the reference's loop calls integrate twice per batch with
 * the same range and init['rot']: train.py:200-215 -> imu_integrator.py:69-164): world_* get nframes + 1 rows (row 0 = the initial
 * state), motion_* nframes rows (every frame from p = v = 0).  Bit-identical to islam_imu_preint(motion_mode = 0) followed by
 * islam_imu_preint(motion_mode = 1); same scratch size. */
int islam_imu_preint_both(const void* dt, const void* gyro, const void* acc, const int64_t* seg, int nframes, int64_t S,
                          int max_frame_samples, const void* init_pos, const void* init_rot, const void* init_vel, double gravity,
                          void* world_pos, void* world_rot, void* world_vel, void* motion_pos, void* motion_rot, void* motion_vel,
                          void* scratch, int dtype, void* stream);

/* Backward of islam_imu_preint w.r.t. the gyro and accelerometer samples: what PyPose's autograd returns through
 * pp.module.IMUPreintegrator when the denoiser runs with grad enabled (imu_integrator.py:107-113,146-153 with eval=False;
 * the IMU-target epoch of train.py:177-179,207-212 -- SURVEY F6).  fwd_scratch: the scratch buffer of the forward call on the
 * same inputs (holds incre_r, the frame-start rotations and the per-frame sums).  g_pos (rows,3), g_rot (rows,4), g_vel (rows,3):
 * gradients of the forward's outputs (rows as in the forward: nframes in motion mode, nframes+1 in world mode), any may be NULL
 * (= zero); g_rot is in PyPose's convention: a left-perturbation tangent vector in slots 0..2, slot 3 ignored.
 * g_gyro, g_acc: (S,3), every row of a non-empty frame is written (rows of samples outside all frames are left untouched:
 * zero-initialise).  scratch: islam_imu_preint_bwd_scratch_bytes(nframes) bytes.  Arithmetic in float64 for either dtype. */
size_t islam_imu_preint_bwd_scratch_bytes(int nframes);
int islam_imu_preint_bwd(const void* dt, const void* gyro, const void* acc, const int64_t* seg, int nframes, int64_t S,
                         double gravity, int motion_mode, const void* fwd_scratch, const void* g_pos, const void* g_rot,
                         const void* g_vel, void* g_gyro, void* g_acc, void* scratch, int dtype, void* stream);

/* ---------------------------------------------------------------- PVGO (pose-velocity graph optimisation) */

typedef struct {
    double w[4];          /* information scalars: lw0^2 (VO), lw1^2 (dvel), lw2^2 (IMU rot), lw3^2 (transvel); pvgo.py:125-129 */
    double radius;        /* TrustRegion(radius=..), pvgo.py:170 */
    double vmin, vmax;    /* LM(min=1e-4), max=1e32 default: diagonal clamp */
    double high, low, up, down, factor, rmin, rmax;   /* ppost.TrustRegion defaults 0.5,1e-3,2,0.5,0.5,1e-6,1e16 */
    int reject;           /* LM reject=16 default */
    int max_steps;        /* StopOnPlateau(steps=10) */
    int patience;         /* StopOnPlateau(patience=3) */
    double decreasing;    /* StopOnPlateau(decreasing=1e-3) */
    int seg_len[2];       /* partitioned block-Cholesky: interior nodes per segment at level 0 / 1 (0 = auto) */
} islam_pvgo_params;

typedef struct {
    int steps;            /* optimizer.step() calls made */
    int trials;           /* damped solves (inner while-loop iterations) over all steps */
    int status;           /* ISLAM_OK or ISLAM_ENOTPD (step broken like PyPose) */
    double loss;          /* final unweighted loss */
    double damping;       /* final damping */
} islam_pvgo_result;

#define ISLAM_PVGO_MAX_LEVELS 6   /* levels of the partitioned block Cholesky; plan arrays hold 3*ISLAM_PVGO_MAX_LEVELS ints */

void islam_pvgo_default_params(islam_pvgo_params* p);
size_t islam_pvgo_workspace_bytes(int N);

/* Whole LM loop on a canonical chain graph (links[k] = [k, k+1], E = N-1), float64.
 * Replaces pvgo.py:168-180: PoseVelGraph + pp.optim.LM(Cholesky, TrustRegion) + StopOnPlateau.
 * nodes (N,7), vels (N,3): in = initial values, out = optimised (NOT yet aligned, see islam_pvgo_align);
 * poses (N-1,7) VO motions; drots (N-1,4), dtrans (N-1,3), dvels (N-1,3), dts (N-1).
 * trace: optional HOST buffer of 3*trace_cap doubles receiving (loss, damping, accepted) per trial. */
int islam_pvgo_run_chain(double* nodes, double* vels, const double* poses, const double* drots,
                         const double* dtrans, const double* dvels, const double* dts, int N,
                         const islam_pvgo_params* prm, void* workspace, size_t workspace_bytes,
                         islam_pvgo_result* result, double* trace, int trace_cap, void* stream);

/* Sparse reprojection factor, the optional 5th residual of the graph (pvgo.py:53-61,130-143,163-165 with
 * dense_ba.py:276-305 SparseReprojectionLoss and pypose reprojerr/point2pixel):
 *   err[k][j] = pixel(K, T_k^-1 * points[k][j]) - targets[k][j],  T_k = rgb2imu^-1 (X_k^-1 X_{k+1}) rgb2imu,
 * weight (loss_weight[4]/K)^2 per row.  Link k couples only nodes k, k+1, so the system stays block-tridiagonal. */
typedef struct {
    const double* points;     /* (N-1, K, 3) device: SparseReprojectionLoss.point3d (camera frame of node k) */
    const double* targets;    /* (N-1, K, 2) device: SparseReprojectionLoss.target (pixels in frame k+1) */
    int K;                    /* SparseReprojectionLoss.N keypoints per link */
    double fx, fy, cx, cy;    /* SparseReprojectionLoss.K */
    double rgb2imu[7];        /* camera->IMU pose [t, q xyzw] */
    double weight;            /* (loss_weight[4]/K)^2, pvgo.py:131 */
    int compat_first_motion;  /* 1: replicate pvgo.py:57 `motion[0] = 0.1` (link 0 becomes a constant residual) */
} islam_pvgo_reproj;

#define ISLAM_REPROJ_REC 32   /* doubles per link written by islam_pvgo_reproj_reduce */

/* islam_pvgo_run_chain with the reprojection factor (reproj == NULL: identical to islam_pvgo_run_chain). */
int islam_pvgo_run_chain_reproj(double* nodes, double* vels, const double* poses, const double* drots,
                                const double* dtrans, const double* dvels, const double* dts, int N,
                                const islam_pvgo_params* prm, const islam_pvgo_reproj* reproj, void* workspace,
                                size_t workspace_bytes, islam_pvgo_result* result, double* trace, int trace_cap,
                                void* stream);
/* Stage-level: per-link reduction over the K keypoints at nodes (dx == NULL) or at Exp(dx)*nodes (dx (N,9)):
 * red (N-1, ISLAM_REPROJ_REC) = [ J^T J upper triangle (21, row-major) | J^T r (6) | r^T r (1) | pad ], J = d err / d eta
 * for the left perturbation T_k <- Exp(eta) T_k (unweighted). */
int islam_pvgo_reproj_reduce(const double* nodes, const double* dx, int N, const islam_pvgo_reproj* reproj,
                             double* red, void* stream);

/* Stage-level entry points (same kernels the loop above launches; exported for parity tests/profiling). */
/* residuals + Jacobian blocks per link -> lin (42 x M, component-major), loss_part (nblocks) */
int islam_pvgo_linearize(const double* nodes, const double* vels, const double* poses, const double* drots,
                         const double* dtrans, const double* dvels, const double* dts, int N,
                         double* lin, double* loss_part, void* stream);
/* block-tridiagonal normal equations: Hd (N,9,9), Ho (N-1,9,9) [rows k, cols k+1], rhs (N,9) = -J^T W r, diag clamped */
int islam_pvgo_build_normal(const double* lin, const double* dts, int N, const double w[4], double vmin, double vmax,
                            double* Hd, double* Ho, double* rhs, void* stream);
/* Hd.diag += Hd.diag*damping (in place, cumulative), then solve -> dx (N,9).  status (device int[4]). */
int islam_pvgo_solve_chain(double* Hd, const double* Ho, const double* rhs, double damping, int N,
                           const int seg_len[2], void* workspace, size_t workspace_bytes, double* dx, void* stream);
/* Stream-ordered variant for iterative methods (islam_amd/pvgo_dense.py: preconditioner of the loop-closure solver): enqueue
 * only -- no read-back, no synchronisation.  islam_pvgo_solve_status synchronises the stream, returns ISLAM_ENOTPD if any solve
 * enqueued on this workspace since the previous status call met a non-positive pivot, and re-initialises the workspace's
 * status / hand-off words; call it once before the first enqueue on a fresh workspace. */
int islam_pvgo_solve_chain_enqueue(double* Hd, const double* Ho, const double* rhs, double damping, int N,
                                   const int seg_len[2], void* workspace, size_t workspace_bytes, double* dx, void* stream);
int islam_pvgo_solve_status(int N, void* workspace, size_t workspace_bytes, void* stream);
/* Profiling variant: HIP events around every launch of one solve (on `stream`).  ms[i] = duration of launch i
 * (eliminate level 0..L-1, then back-substitution L-2..0; at most 2*ISLAM_PVGO_MAX_LEVELS-1 entries),
 * plan[3*l..] = (nodes, segment length, segments) for l < ISLAM_PVGO_MAX_LEVELS, plan[3*ISLAM_PVGO_MAX_LEVELS] = first
 * level solved inside the single-workgroup top kernel (3*ISLAM_PVGO_MAX_LEVELS+1 ints). */
int islam_pvgo_solve_chain_timed(double* Hd, const double* Ho, const double* rhs, double damping, int N,
                                 const int seg_len[2], void* workspace, size_t workspace_bytes, double* dx,
                                 float* ms, int* plan, int* nlaunch, void* stream);
/* Measurement hook: exactly the level-0 up-sweep launch of islam_pvgo_solve_chain (same kernel, grid, arguments) and
 * nothing else -- bench.py times a burst of these for the roofline figure.  Hd's diagonal is damped in place. */
int islam_pvgo_eliminate_level0(double* Hd, const double* Ho, const double* rhs, double damping, int N,
                                const int seg_len[2], void* workspace, size_t workspace_bytes, void* stream);
/* Measurement hook: the LM loop's dominant launch in its steady state -- trial step + loss / trust-region sums of a trial
 * (pvgo.py:26-64, ppost.TrustRegion), the linearisation at the trial point and the level-0 elimination of the next damped solve
 * (pp.optim.LM's J^T W J + Cholesky) in ONE kernel (trial_elim_kernel) -- after one linearisation and one solve of the given
 * problem: `launches` back-to-back launches between one pair of HIP events on `stream`; *us_per_launch = average period.
 * info[0..2] = (level-0 segment length, segments, workgroups).  ISLAM_EARG when the fused loop does not cover this N. */
int islam_pvgo_trial_elim_burst(const double* nodes, const double* vels, const double* poses, const double* drots,
                                const double* dtrans, const double* dvels, const double* dts, int N,
                                const islam_pvgo_params* prm, void* workspace, size_t workspace_bytes, int launches,
                                float* us_per_launch, int* info, void* stream);
/* ---- multi-GPU building blocks (islam_amd/dist_pvgo.py; no reference counterpart: the reference is single-GPU).
 * plan9 (3*ISLAM_PVGO_MAX_LEVELS+1 ints) receives (nodes, segment length, segments) per level (unused = 0) and, last,
 * the first level that runs inside the single-workgroup top kernel; returns the level count. */
int islam_pvgo_plan(int N, const int seg_len[2], int* plan9);
/* Sharding with an interface-only exchange: a rank owns a contiguous range of the segments of the
 * exchange level xl -- the highest level below the root with at least `world` segments -- and everything below it between the
 * two outer separators of that range.  islam_pvgo_shard_ranges: out[0] = xl, out[1] = P_xl, out[2+2l], out[3+2l] = first
 * segment / number of segments owned at level l (2 + 2*ISLAM_PVGO_MAX_LEVELS ints).  upsweep: levels 0..xl over the rank's own
 * segments; Hd/Ho/rhs are LOCAL level-0 arrays (row 0 = global node `node0`), `exchange` (351*P_xl doubles, array-major like the
 * level-0 products) is zeroed and the rank's rows written: summing it over the ranks is the ONLY data-path collective of a
 * solve (P_xl = 23 at N=5001: 64.6 KB, against 2.34 MB for the level-0 products).  downsweep: the levels above xl from the
 * summed buffer (redundantly), then the local back-substitution; dx is LOCAL like Hd.  Factors live in `workspace`
 * (islam_pvgo_workspace_bytes(N), the same buffer for both calls, ZERO-INITIALISED once by the caller: the product rows of
 * other ranks' segments are read as zero contributions). */
int islam_pvgo_shard_ranges(int N, const int seg_len[2], int world, int rank, int* out);
int islam_pvgo_shard_upsweep(double* Hd, const double* Ho, const double* rhs, double damping, int N, const int seg_len[2], int world,
                             int rank, int node0, void* workspace, size_t workspace_bytes, double* exchange, int* flags,
                             void* stream);
int islam_pvgo_shard_downsweep(const double* exchange, int N, const int seg_len[2], int world, int rank, int node0, void* workspace,
                               size_t workspace_bytes, double* dx, int* flags, void* stream);
/* The whole sharded LM loop in the library, on RCCL (islam_amd/csrc/pvgo_dist.hip): every rank passes the SAME full-size
 * inputs (device, float64: nodes (N,7), vels (N,3) in/out -- the full solution on every rank --, poses (N-1,7), drots (N-1,4),
 * dtrans, dvels (N-1,3), dts (N-1)) and works on its stretch of the chain.  Graphs the fused trial + elimination kernel covers
 * (those islam_pvgo_run_chain fuses, without the reprojection factor): ONE all-reduce per LM trial -- the interface blocks of the
 * next solve (351 doubles per segment of the exchange level) + [sum r^2 | sum JD.(2R+JD) | failed | 18 doubles per cut: the two
 * ranks' parts of the cut node's diagonal] -- with the trial step, the linearisation and the level-0 elimination in one launch
 * under a speculated damping; a rejected or re-damped trial costs one more solve with its own all-reduce (ISLAM_SHARD_FUSED=0:
 * off).  Otherwise per LM trial two all-reduces (the interface blocks;
 * [sum r^2 | sum JD.(2R+JD) | failed | 10-double halo per rank]); accept /
 * reject, TrustRegion and StopOnPlateau replicated on the DEVICE (every rank decides on the same summed scalars), the host
 * runs one trial ahead and reads 128-byte verdicts from pinned memory -- no stream synchronisation inside the loop; the
 * collectives of a cancelled run-ahead trial still execute, identically on every rank.  reproj (may be NULL): the sparse
 * reprojection factor of islam_pvgo_run_chain_reproj over the WHOLE graph; a rank reduces the keypoints of its own links.  comm: an ncclComm_t made
 * with islam_dist_comm_init (rank 0 creates the 128-byte id with islam_dist_unique_id and broadcasts it -- e.g. with
 * torch.distributed) or NULL for world == 1.  workspace: islam_pvgo_workspace_bytes(N); scratch:
 * islam_pvgo_sharded_scratch_bytes(N, world).  exchanged_bytes (may be NULL): bytes handed to the collectives of the trials.
 * The _cb variant takes the all-reduce as a callback (in-place sum of `count` doubles at `buf`, stream-ordered on `stream`;
 * return 0): the tests drive several ranks as threads on one GPU through it. */
typedef int (*islam_allreduce_fn)(void* user, double* buf, size_t count, void* stream);
int islam_dist_unique_id(void* out128);
int islam_dist_comm_init(const void* id128, int world, int rank, void** comm);
int islam_dist_comm_destroy(void* comm);
/* What RCCL itself says about a communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): bench.py prints `ranks` as
 * rccl_ranks_seen so that a multi-GPU record proves the collectives ran over that many ranks.  comm == NULL (world 1): 1, 0, the
 * current device.  Any of the three outputs may be NULL. */
int islam_dist_comm_info(void* comm, int* ranks, int* rank, int* device);
size_t islam_pvgo_sharded_scratch_bytes(int N, int world);
int islam_pvgo_run_chain_sharded(void* comm, int world, int rank, double* nodes, double* vels, const double* poses, const double* drots,
                                 const double* dtrans, const double* dvels, const double* dts, int N, const islam_pvgo_params* prm,
                                 const islam_pvgo_reproj* reproj, void* workspace, size_t workspace_bytes, void* scratch,
                                 size_t scratch_bytes, islam_pvgo_result* res, long long* exchanged_bytes, void* stream);
int islam_pvgo_run_chain_sharded_cb(islam_allreduce_fn fn, void* user, int world, int rank, double* nodes, double* vels,
                                    const double* poses, const double* drots, const double* dtrans, const double* dvels, const double* dts,
                                    int N, const islam_pvgo_params* prm, const islam_pvgo_reproj* reproj, void* workspace,
                                    size_t workspace_bytes, void* scratch, size_t scratch_bytes, islam_pvgo_result* res,
                                    long long* exchanged_bytes, void* stream);
/* Trial step on M links (M+1 node rows): retract on a copy, residuals, partial sums part[2*nblk] =
 * (sum r^2, sum JD.(2R+JD)) per 64-link block (ppost.TrustRegion.update's denominator is -sum). */
int islam_pvgo_trial(const double* nodes, const double* vels, const double* dx, const double* poses, const double* drots,
                     const double* dtrans, const double* dvels, const double* dts, const double* lin, int M,
                     double* nodes_t, double* vels_t, double* part, void* stream);
/* X <- Exp(sign*dx[:, :6]) * X ; v += sign*dx[:, 6:]  (LieTensor.add_) */
int islam_pvgo_retract(const double* nodes, const double* vels, const double* dx, double sign, int N,
                       double* nodes_out, double* vels_out, void* stream);

/* VO factors on ARBITRARY edges (loop closures; pvgo.py:36-39): out (24,E) component-major =
 * e(6) | G(9) | C(9) with d e/d delta_j = [[G,C],[0,G]], d e/d delta_i = -that.  Used by the dense general-topology
 * path (islam_amd/pvgo_dense.py). */
int islam_pvgo_linearize_edges(const double* nodes, const int64_t* edges, const double* poses, int E, double* out,
                               void* stream);
/* General-topology normal equations without ever forming J: A (9N x 9N row-major, fully written) and rhs (9N) from
 *   - the block-tridiagonal IMU part Hd, Ho (N,9,9), rhs_chain (N,9)  (islam_pvgo_build_normal with w[0] = 0, no clamp),
 *   - the VO factors of arbitrary edges `vo` (24,E) from islam_pvgo_linearize_edges, weight w0.
 * node_ptr (N+1), node_adj (2E) = CSR of the edge ends per node, entry = 2*edge + end (end 0 = i, 1 = j), so every
 * diagonal block and right-hand side is summed in a fixed order (bit-reproducible); off-diagonal VO blocks are added
 * with one atomic per entry.  Replaces PyPose's dense J^T W J (pp.optim.LM) on the loop-closure path. */
int islam_pvgo_assemble_dense(const double* Hd, const double* Ho, const double* rhs_chain, const double* vo,
                              const int64_t* edges, const int64_t* node_ptr, const int64_t* node_adj, double w0,
                              int N, int E, double* A, double* rhs, void* stream);
/* vo_loss forward/backward (pvgo.py:67-78 with PyPose's left-tangent gradient convention).
 * fwd: e (E,6) = Log(P^-1 Xi^-1 Xj); trans_loss, rot_loss (E).  bwd: grad_poses (E,7), last column 0. */
int islam_pvgo_vo_loss_fwd(const double* nodes, const int64_t* edges, const double* poses, int E,
                           double* err6, double* trans_loss, double* rot_loss, void* stream);
int islam_pvgo_vo_loss_bwd(const double* poses, const double* err6, const double* g_trans, const double* g_rot,
                           int E, double* grad_poses, void* stream);
/* align_to (pvgo.py:114-119): nodes <- target * nodes[0]^-1 * nodes ; vels <- R(target) R(nodes[0])^-1 vels */
int islam_pvgo_align(const double* nodes, const double* vels, const double* target7, int N,
                     double* nodes_out, double* vels_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISLAM_HIP_H */
