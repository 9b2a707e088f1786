// head_infer.hip -- DIAGNOSTICS BUILD ONLY (a measured alternative, round 5; ACEZ_HEAD_INFER=1): the head's forward pass for INFERENCE on
// many rows (whole frames: Regressor.forward, ace_network.py:120-149,265-270; register_mapping.py:201-213 runs it on every frame) as ONE
// launch: a workgroup keeps its 64 rows' activations in LDS and walks all wide layers, the weights stream past it.
//
// Why it was built: per 64 frames of 480 x 640 the registration pipeline spends 4.2 ms in the encoder and 2.3 ms in the head -- eight
// dependent 307200 x 512 x 512 layers on the large-tile convolution kernels, each reading and writing 315 MB of activations. Rows are
// independent through the whole MLP, so nothing but the first input and the last output has to touch HBM.
// What was measured (MI355X, 307200 rows, tools/_gpu_s3.sh-style sweep, results BITWISE equal to the large-tile launches in every
// configuration): 2.69 ms for all eight layers in one launch against 2.26-2.40 ms for the eight large-tile launches; in groups of 4 / 3 /
// 2 / 1 layers per launch 2.77 / 2.83 / 2.93 / 3.26 ms (so it is not the eight layers' 4 MiB of weights cycling through a 4 MiB L2), with
// non-temporal hints on all activation traffic 2.84 ms (nor the activations evicting the weights).
// Why: every workgroup pulls all 512 KiB of a layer's weights through its CU's L2 -> LDS path for only 64 rows, and that path delivers
// bytes-in-flight / latency: 64 KiB of activations + the 80 KiB ring (five 16 KiB stages, four in flight) IS the LDS, 64 KiB in flight at
// ~2 us of loaded latency = 28-30 GB/s per CU = 18 us per layer and tile (the row-persistent training chain of round 2 reached 70 GB/s
// with 112 KiB in flight, the 80-row chains 83 GB/s with 84 KiB at 1 us). rows / us = R x (144 KiB - R KiB) / (512 KiB x latency) peaks
// at R ~ 70: the design point IS the optimum of this structure, and it is 15-20 % behind kernels that re-use a weight byte over 256 rows
// and pay for it in HBM traffic. Kept as the measurement (and as a correct one-launch path: tests/test_head_infer_gpu.py).
//
// Layout: weights are re-packed once per weight version (headinfer_pack_kernel) into the order the stream is consumed in:
//   Wp[layer][k-step of 16][k half][out channel 512][8 elements]  -- a 16 KiB stage is contiguous (1 KiB per DMA instruction) and lands
//   in LDS as it is; an MFMA A fragment (32 channels x 16 k) is two runs of 32 consecutive 16-byte chunks: conflict free.
// Activations: [64][512] with the 16-byte chunk index XOR row & 15 (act_off): conflict-free B fragments, in-place epilogue.
// MFMA v_mfma_f32_32x32x16 as D[channel][row], K ascending: the accumulation order of the large-tile convolution kernels.
// Rounding points are the training forward's (bias + ReLU rounded to 16 bits, residual added to the ROUNDED value, rounded again).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "head_kernels.h"

namespace acez {

struct HeadInferLayer {
  const uint16_t* res_in;   // residual layer (ace_network.py:126,133): [n][512] the block's input, added after the activation; else null
  uint16_t* res_out;        // where this layer's output goes if a LATER layer adds it as its residual (null: not needed again)
};
struct HeadInferArgs {
  const uint16_t* feat;     // [n][512] input rows
  uint16_t* out;            // [n][512] output of the last wide layer (fc2): the input of the fc3 / de-homogenisation kernel
  const uint16_t* Wp;       // packed weights (headinfer_pack_kernel)
  const float* params;      // flat fp32 parameters (bias of layer l at l * 262656 + 262144)
  int n, n_layers;
  HeadInferLayer layer[MAX_LAYERS];
};

constexpr int HI_ROWS = 64;
constexpr int HI_STAGE = 512 * 16;     // elements per stage: one K step of 16 for all 512 output channels (16 KiB)
constexpr int HI_NS = 5;               // ring slots; HI_NS - 1 stages (64 KiB) in flight
constexpr int HI_KS = 512 / 16;        // stages per layer
// The activations cross the L2 once (input rows in, output / residual rows out and, a few layers later, back in) beside a weight stream
// that every workgroup re-reads: they are moved with non-temporal hints so that they do not push the weights out of the L2.
#ifndef HI_NT
#define HI_NT 0     // (measured: 2.84 ms with the hints against 2.69 without)
#endif

// Wp[layer][ks][h][row][e] = Wb[layer][row][ks * 16 + h * 8 + e]; one thread per 16-byte chunk
__global__ __launch_bounds__(256) void headinfer_pack_kernel(const uint16_t* __restrict__ Wb, uint16_t* __restrict__ Wp, int n_layers) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= (int64_t)n_layers * 512 * 64) return;
  const int layer = (int)(c / (512 * 64)), d = (int)(c - (int64_t)layer * 512 * 64);
  const int row = d & 511, sh = d >> 9, ks = sh >> 1, h = sh & 1;
  *reinterpret_cast<uint4*>(Wp + (size_t)layer * 262144 + (size_t)d * 8) =
      *reinterpret_cast<const uint4*>(Wb + (size_t)layer * 262144 + (size_t)row * 512 + ks * 16 + h * 8);
}

template <class E = EltBf16>
__global__ __launch_bounds__(512) void headinfer_kernel(HeadInferArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[HI_ROWS * 512 + HI_NS * HI_STAGE];
  uint16_t* const act = smem;
  uint16_t* const ring = smem + HI_ROWS * 512;
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = l & 31, fh = l >> 5;
  const int m0 = blockIdx.x * HI_ROWS, n = a.n;
  const int G = a.n_layers * HI_KS;

  // each wave issues 2 of the 16 DMA instructions of a stage
  auto issue = [&](int g) {
    const uint16_t* src = a.Wp + (size_t)g * HI_STAGE + (size_t)(2 * w) * 512 + l * 8;
    uint16_t* dst = ring + (g % HI_NS) * HI_STAGE + (2 * w) * 512;
    __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gvoid_t*)(src + 512), (lvoid_t*)(dst + 512), 16, 0, 0);
  };
  // input rows 8 w .. 8 w + 7 -> the activation tile: physical chunk l of a row holds the logical chunk l ^ (row & 15)
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = 8 * w + r;
    const uint16_t* src = a.feat + (size_t)min(m0 + row, n - 1) * 512 + ((l ^ (row & 15)) << 3);
    __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(act + row * 512), 16, 0, HI_NT ? 2 : 0);   // (nt: see HI_NT)
  }
#pragma unroll
  for (int g = 0; g < HI_NS - 1; ++g)
    if (g < G) issue(g);

  // Vector-memory operations complete in order, so `s_waitcnt vmcnt(N)` certifies stage g once N is NOT LARGER than the number of younger
  // operations in flight: the stages g + 1 .. g + 3 (two instructions each) and, for the four iterations after it was issued, a burst of
  // residual loads (16), bias loads (8) or tile stores (this wave's valid rows), issued in iteration *_at behind that iteration's stage request.
  int ld_at = -100, st_at = -100, st_n = 0, bs_at = -1;   // (the first layer's bias loads go out right behind the first four stages)
  const int brow0 = fr, brow1 = 32 + fr;
  // this lane's bias values: channels w * 64 + i * 32 + 8 q + 4 fh .. + 3. Lane-dependent addresses ON PURPOSE: eight vector loads per
  // layer, a known number for the counted waits below (left to the compiler, wave-uniform addresses may or may not become scalar loads);
  // the next layer's are requested half a layer ahead
  auto load_bias = [&](int layer, float4 (&bv)[2][4]) {
    const float* bias = a.params + (int64_t)layer * 262656 + 262144 + w * 64 + 4 * fh;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const float4*>(bias + i * 32 + 8 * q);
  };
  float4 bv[2][4], bvn[2][4];
  load_bias(0, bv);
  for (int layer = 0; layer < a.n_layers; ++layer) {
    const HeadInferLayer L = a.layer[layer];
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    uint2 res[2][2][4];
    for (int ks = 0; ks < HI_KS; ++ks) {
      const int g = layer * HI_KS + ks;
      int allowed = 2 * min(HI_NS - 2, G - 1 - g);
      if (g - ld_at >= 1 && g - ld_at <= HI_NS - 1) allowed += 16;
      if (g - st_at >= 1 && g - st_at <= HI_NS - 1) allowed += st_n;
      if (g - bs_at >= 1 && g - bs_at <= HI_NS - 1) allowed += 8;
      wait_vmcnt_dyn(allowed);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of stage g - 1 / its epilogue writes are complete
      __builtin_amdgcn_s_barrier();                          // stage g (and at g = 0 the input tile) has landed everywhere; stage g - 1 is read
      if (g + HI_NS - 1 < G) issue(g + HI_NS - 1);
      if (ks == 0 && L.res_in) {   // the residual values in the accumulator layout, on their way while the layer multiplies
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
              const uint2* rp = reinterpret_cast<const uint2*>(L.res_in + (size_t)min(m0 + j * 32 + fr, n - 1) * 512 + w * 64 + i * 32 + 8 * q + 4 * fh);
              if (HI_NT) { res[i][j][q].x = __builtin_nontemporal_load(&rp->x); res[i][j][q].y = __builtin_nontemporal_load(&rp->y); }
              else res[i][j][q] = *rp;
            }
        ld_at = g;
      }
      if (ks == HI_KS / 2 && layer + 1 < a.n_layers) { load_bias(layer + 1, bvn); bs_at = g; }
      const uint16_t* sW = ring + (g % HI_NS) * HI_STAGE + (fh * 512 + w * 64 + fr) * 8;
      const int chunk = ks * 2 + fh;
      const typename E::frag fa0 = *reinterpret_cast<const typename E::frag*>(sW);
      const typename E::frag fa1 = *reinterpret_cast<const typename E::frag*>(sW + 32 * 8);
      const typename E::frag fb0 = *reinterpret_cast<const typename E::frag*>(&act[brow0 * 512 + ((chunk ^ (brow0 & 15)) << 3)]);
      const typename E::frag fb1 = *reinterpret_cast<const typename E::frag*>(&act[brow1 * 512 + ((chunk ^ (brow1 & 15)) << 3)]);
      acc[0][0] = E::mfma32(fa0, fb0, acc[0][0]);
      acc[1][0] = E::mfma32(fa1, fb0, acc[1][0]);
      acc[0][1] = E::mfma32(fa0, fb1, acc[0][1]);
      acc[1][1] = E::mfma32(fa1, fb1, acc[1][1]);
    }
    // ---- layer epilogue, in place: every wave has read the whole tile before any wave overwrites its 64 channels of it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 32 + fr;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = w * 64 + i * 32 + 8 * q + 4 * fh;
          const float4 b = bv[i][q];
          uint2 y = E::pk4(fmaxf(acc[i][j][4 * q + 0] + b.x, 0.f), fmaxf(acc[i][j][4 * q + 1] + b.y, 0.f), fmaxf(acc[i][j][4 * q + 2] + b.z, 0.f),
                           fmaxf(acc[i][j][4 * q + 3] + b.w, 0.f));
          if (L.res_in) {   // x = res + relu(conv): the rounded activation plus the residual, rounded again (rowgemm80_body AUX_RESIDUAL)
            float yf[4], rf[4];
            E::un4(y, yf);
            E::un4(res[i][j][q], rf);
            y = E::pk4(yf[0] + rf[0], yf[1] + rf[1], yf[2] + rf[2], yf[3] + rf[3]);
          }
          *reinterpret_cast<uint2*>(&act[row * 512 + ((((ch >> 3) ^ (row & 15)) << 3) | (ch & 7))]) = y;
        }
    }
    uint16_t* const gout = (layer + 1 == a.n_layers) ? a.out : L.res_out;
    if (gout) {   // the finished tile -> HBM in full 1 KiB rows (the last layer's output; a residual stream a later layer adds)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      uint4 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = 8 * w + r;
        v[r] = *reinterpret_cast<const uint4*>(&act[row * 512 + ((l ^ (row & 15)) << 3)]);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = 8 * w + r;
        if (m0 + row < n) {
          uint4* gp = reinterpret_cast<uint4*>(gout + (size_t)(m0 + row) * 512 + l * 8);
          if (HI_NT) { __builtin_nontemporal_store(v[r].x, &gp->x); __builtin_nontemporal_store(v[r].y, &gp->y); __builtin_nontemporal_store(v[r].z, &gp->z); __builtin_nontemporal_store(v[r].w, &gp->w); }
          else *gp = v[r];
        }
      }
      st_at = (layer + 1) * HI_KS - 1;                    // (issued behind the last iteration of this layer)
      st_n = min(8, max(0, n - (m0 + 8 * w)));            // this wave's stores: exactly its valid rows
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[i][q] = bvn[i][q];
  }
}

}  // namespace acez
