// head_fused.hip -- persistent row-tile forward of the head: one workgroup walks ALL wide layers for its 32 rows.
//
// Why: a 5120x512x512 layer is only 2.7 GFLOP; as 8 dependent launches the forward chain cost 8 x (8.7 us kernel +
// boundary) although each launch holds < 3 us of MFMA/load work (tools/ablate_rowgemm.hip). Rows are independent
// through the whole MLP (ace_network.py:120-137 is a chain of 1x1 convolutions), so a workgroup can keep its rows'
// activations in LDS and never synchronise with other workgroups:
//
//   LDS (160 KiB, one workgroup per CU):  R | P0 | P1  = three [32 rows][512 ch] bf16 activation tiles (96 KiB)
//                                         ring         = two [256 out-ch][64 k] bf16 weight stages      (64 KiB)
//   per layer: 16 stages (8 K-steps x 2 output halves) streamed L2 -> LDS by global_load_lds_dwordx4 (double buffered:
//   stage g+1 is in flight while stage g is multiplied); MFMA 32x32x16 bf16 as D[out-ch][row]; the epilogue
//   (bias, ReLU, residual add in place on R, bf16) writes the next layer's input tile straight into LDS and the
//   tile is copied to HBM in full 1 KiB rows for the backward pass one barrier later.
//   The gather of the input rows (ace_trainer.py:485-494) is the tile load of layer 0.
//
// The weight stream is the bound: every workgroup reads all 512 KiB of a layer, 160 workgroups -> 84 MB per layer
// from L2 (~2.6 us at the measured ~32 TB/s aggregate L2 rate), vs 1.1 us of MFMA time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "head_kernels.h"

namespace acez {

struct FusedLayer {
  int in_buf, out_buf;     // 0 = R, 1 = P0, 2 = P1
  int residual;            // after the activation: R += out (in place), ace_network.py:126,133
  uint16_t* g_out;         // [n][512] post-ReLU output in HBM (null: not stored)
  uint16_t* g_res;         // [n][512] new residual stream in HBM (residual layers; null: not stored)
};

struct HeadFwdArgs {
  const uint16_t* feat;    // feature rows
  const int64_t* idx;      // row gather indices into feat, or null (rows m0.. taken directly)
  uint16_t* g_in;          // [n][512] copy of the gathered input rows (needed by the weight-gradient pass) or null
  int n, n_layers;
  const uint16_t* Wb;      // [L][512][512] bf16
  const float* params;     // flat fp32 parameters (bias of layer l at l*262656 + 262144)
  FusedLayer layer[MAX_LAYERS];
  const TrainState* st;
};

// activation tile addressing: act_off() of head_kernels.hip

__global__ __launch_bounds__(256, 1) void headfwd_kernel(HeadFwdArgs a) {
  if (a.st && !a.st->active) return;
  __shared__ __attribute__((aligned(16))) uint16_t smem[3 * 32 * 512 + 2 * 256 * 64];
  uint16_t* const acts = smem;                  // 3 tiles
  uint16_t* const ring = smem + 3 * 32 * 512;   // 2 stages
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m0 = blockIdx.x * 32;
  const int n = a.n;

  // ---- input tile -> R (and its HBM copy): wave w loads rows 8w .. 8w+7, one 1 KiB row per instruction
  {
    uint4 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = w * 8 + r;
      const int m = min(m0 + row, n - 1);
      const int64_t src = a.idx ? a.idx[m] : (int64_t)m;
      v[r] = *reinterpret_cast<const uint4*>(a.feat + src * 512 + l * 8);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = w * 8 + r;
      *reinterpret_cast<uint4*>(&acts[act_off(row, l * 8)]) = v[r];
      if (a.g_in && m0 + row < n) *reinterpret_cast<uint4*>(a.g_in + (size_t)(m0 + row) * 512 + l * 8) = v[r];
    }
  }

  const int G = a.n_layers * 16;
  // DMA of stage g = (layer, kt, half): wave w moves rows (w*8+j)*8 .. +7 of the 256-row half, j = 0..7
  auto issue = [&](int g) {
    const int layer = g >> 4, s = g & 15, kt = s >> 1, half = s & 1;
    const uint16_t* Wl = a.Wb + (size_t)layer * 262144 + (size_t)(half * 256) * 512 + kt * 64;
    uint16_t* dst = ring + (g & 1) * (256 * 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = (w * 8 + j) * 8 + (l >> 3);
      const int c = (l & 7) ^ ((row >> 1) & 7);
      __builtin_amdgcn_global_load_lds((gvoid_t*)(Wl + (size_t)row * 512 + c * 8), (lvoid_t*)(dst + (w * 8 + j) * 8 * 64), 16, 0, 0);
    }
  };
  // copy a finished [32][512] tile to HBM, 1 KiB per row
  auto store_tile = [&](const uint16_t* tile, uint16_t* g) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = w * 8 + r;
      if (m0 + row < n) *reinterpret_cast<uint4*>(g + (size_t)(m0 + row) * 512 + l * 8) = *reinterpret_cast<const uint4*>(&tile[act_off(row, l * 8)]);
    }
  };

  f32x16 acc[4];
  // one stage: wait for its DMA, barrier, start the next DMA, multiply. Written as a macro-like lambda taking the two
  // accumulators of the stage's output half by reference so that register indices stay static (a runtime
  // accumulator index makes the compiler shuffle the whole accumulator file through VGPRs every iteration).
  auto stage = [&](int g, f32x16& acc0, f32x16& acc1, const FusedLayer& L) {
    const int kt = (g & 15) >> 1;
    // my share of stage g has landed, my earlier tile stores have retired, and my LDS writes (input tile / layer
    // epilogue) are complete -- a raw s_barrier does not wait for any of these by itself
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();    // ... everyone's has; everyone is done with stage g-1 and with the last epilogue
    if (g + 1 < G) issue(g + 1);
    const uint16_t* sW = ring + (g & 1) * (256 * 64);
    const uint16_t* sIn = acts + L.in_buf * (32 * 512);
    const int brow = l & 31;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ck = kt * 8 + kk * 2 + (l >> 5);
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&sIn[brow * 512 + ((ck ^ (brow & 15)) << 3)]);
      const int c = 2 * kk + (l >> 5);
      const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(&sW[swz(w * 64 + (l & 31), c)]);
      const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(&sW[swz(w * 64 + 32 + (l & 31), c)]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb, acc1, 0, 0, 0);
    }
  };
  issue(0);
  for (int layer = 0; layer < a.n_layers; ++layer) {
    const FusedLayer L = a.layer[layer];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    for (int kt = 0; kt < 8; ++kt) {
      const int g = layer * 16 + kt * 2;
      stage(g, acc[0], acc[1], L);
      if (kt == 0 && layer > 0) {    // the previous layer's output tile is complete: copy it out for the backward pass
        const FusedLayer P = a.layer[layer - 1];
        if (P.g_out) store_tile(acts + P.out_buf * (32 * 512), P.g_out);
        if (P.residual && P.g_res) store_tile(acts, P.g_res);
      }
      stage(g + 1, acc[2], acc[3], L);
    }
    // layer epilogue: bias + ReLU (+ residual) -> bf16 -> the next layer's input tile in LDS
    uint16_t* sOut = acts + L.out_buf * (32 * 512);
    const float* bias = a.params + (int64_t)layer * 262656 + 262144;
    const int row = l & 31, hh = l >> 5;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int nbase = (u >> 1) * 256 + w * 64 + (u & 1) * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nn = nbase + 8 * q + 4 * hh;
        const float4 b = *reinterpret_cast<const float4*>(bias + nn);
        const float v0 = fmaxf(acc[u][4 * q + 0] + b.x, 0.f), v1 = fmaxf(acc[u][4 * q + 1] + b.y, 0.f);
        const float v2 = fmaxf(acc[u][4 * q + 2] + b.z, 0.f), v3 = fmaxf(acc[u][4 * q + 3] + b.w, 0.f);
        const uint2 y = pack4(v0, v1, v2, v3);
        const int off = act_off(row, nn);
        *reinterpret_cast<uint2*>(&sOut[off]) = y;
        if (L.residual) {
          float yf[4], rf[4];
          unpack4(y, yf);
          unpack4(*reinterpret_cast<const uint2*>(&acts[off]), rf);
          *reinterpret_cast<uint2*>(&acts[off]) = pack4(yf[0] + rf[0], yf[1] + rf[1], yf[2] + rf[2], yf[3] + rf[3]);
        }
      }
    }
  }
  __syncthreads();
  {
    const FusedLayer P = a.layer[a.n_layers - 1];
    if (P.g_out) store_tile(acts + P.out_buf * (32 * 512), P.g_out);
    if (P.residual && P.g_res) store_tile(acts, P.g_res);
  }
}

}  // namespace acez
