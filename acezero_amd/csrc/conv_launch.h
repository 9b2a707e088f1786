// conv_launch.h -- the implicit-GEMM convolution launcher (encoder_api.hip), shared with the head's large-batch inference path
// (a 1x1 convolution over [rows][512] is the head's layer GEMM).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acez {

struct ConvGemmArgs {
  const uint16_t* In;     // NHWC 16-bit [F][Hi][Wi][Ci]
  const uint16_t* W;      // 16-bit [Co][Kp]
  const float* bias;      // [Co]
  const uint16_t* add;    // [M][Co] or null: added (fp32) after the activation, before the single 16-bit store
  uint16_t* out;          // [M][Co]
  const uint16_t* zeros;  // >= 128 bytes of zeros: DMA source of every padded chunk
  int Hi, Wi, Ci, ci_shift, Ho, Wo, Co, ksize, stride, pad, K, Kp, M;
  // fused pointwise skip (conv3x3r_kernel only; null = none): out = act(conv3x3(In) + bias) + (In2 . W2 + bias2), the second product as
  // extra K stages on the SAME accumulators after the activation was applied to them in registers (ace_network.py:57-58: res2_skip)
  // W2 / bias2 / Kp2 WITHOUT In2: a pointwise Co -> Co layer with ReLU that FOLLOWS this one, run back to back on conv3x3r's finished tile
  // (Co = 256: res1_conv1 + res1_conv2); skip_scratch = the intermediate map of the two-launch fall-back
  const uint16_t* In2;    // [M][Ci2] (the skip layer's input at the output's own pixels)
  const uint16_t* W2;     // 16-bit [Co][Kp2]
  const float* bias2;     // [Co]
  int Ci2, Kp2;
  uint16_t* skip_scratch; // [M][Co]: where the launcher puts the skip product when the layer does not run on conv3x3r (small inputs: two launches)
  int f16;                // 16-bit operand format of In / W / add / out: 0 = bf16, 1 = fp16 (fp32 accumulation in both)
  int round_before_add;   // 1: the activation is rounded to 16 bits BEFORE the residual is added (the head stores it: ace_network.py:126,133)
  unsigned long long* trace;   // diagnostics build (tools/conv_trace.py): [tiles][8] s_memtime stamps of convgemm512's waves 0 and 8; else null
  int dbg;   // ablation (ACEZ_CONV_DBG; 0 in production). convgemm256/512: 2 = no MFMA, 4 = no loads. conv3x3p (loader side only):
             // 32 = weight DMA from 16 hot rows (same bytes into LDS), 64 = every other weight stage not fetched
};

// tile_mode: 0 = choose by size, 80 / 256 / 512 = force that kernel where the layer shape allows it
void launch_convgemm(const ConvGemmArgs& g, bool relu, hipStream_t s, int tile_mode);

}  // namespace acez
