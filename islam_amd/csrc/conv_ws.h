// Internal interface between conv_nhwc.hip (the C-ABI entry points of the channels-last convolutions) and conv_ws.hip (the
// weight-stationary 128 -> 128 kernel) and conv_ws32.hip (the persistent 32 -> 32 kernel) they dispatch to.
#pragma once
#include <hip/hip_runtime.h>

namespace islam {

int conv_ws_set_mode(int mode);                           // returns the previous mode; out-of-range: query only
int conv_ws_spare_cus();
void conv_ws_count_launch(int which);                    // 0: conv3x3_ws_kernel, 1: conv3x3_ws32_kernel (host-side launch counters: islam_conv_ws_launch_counts)
void conv_ws_read_counts(long long out[2]);
int conv_ws_balanced(long long ntiles, int slots);      // smallest launch whose longest tile range is as short as with `slots` workgroups
bool conv_ws_applies(int Cin, int Cout, int ksize, int B, int H, int W);
int conv_ws_blocks(int B, int H, int W);                 // workgroups of the launch = rows of per-workgroup BatchNorm partial sums it writes
// raw bf16 output (no bias / residual / ReLU), optional BatchNorm + ReLU of the input on load, optional per-workgroup partial sums;
// x: channels [xoff, xoff + Cin) of a (B,H,W,xs) tensor (Cin = 64 or 128), y: channels [yoff, yoff + 128) of a (B,H,W,ys) tensor
int conv_ws_launch(const unsigned short* x, const unsigned short* wp, const float* in_affine, unsigned short* y, float* partial, int B, int Cin, int H,
                   int W, int Cout, int CoutP, int xs, int xoff, int ys, int yoff, hipStream_t s);

// conv_ws32.hip: the persistent 32 -> 32 3x3 kernel (same contract; whole tiles of 32 x 16 pixels)
bool conv_ws32_applies(int Cin, int Cout, int ksize, int B, int H, int W);
int conv_ws32_blocks(int B, int H, int W);
int conv_ws32_launch(const unsigned short* x, const unsigned short* wp, const float* in_affine, unsigned short* y, float* partial, int B, int H, int W,
                     int CoutP, int xs, int xoff, int ys, int yoff, hipStream_t s);

}  // namespace islam
