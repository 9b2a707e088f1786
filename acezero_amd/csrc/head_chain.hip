// head_chain.hip -- the training step's dependent chain as ONE launch: gather, the 8 forward layers, fc3 + loss + its
// gradient, and the 7 input-gradient layers (ace_trainer.py:485-613, ace_network.py:120-147 and their autograd).
//
// Why: as 1 + 8 + 1 + 7 dependent launches the chain cost 7.7 + 57.8 + 21.8 + 60.8 us per 5120-row step although it holds
// 16 us of MFMA work: every launch pays its boundary, the dispatch of 256 large-LDS workgroups, a cold operand pipeline and
// an epilogue whose stores must drain (round-1 profile). The rows of a batch are independent through the whole MLP and
// through the loss, so a workgroup that owns 32 rows can walk all of it without ever synchronising with another workgroup.
//
// Structure (one 512-thread workgroup per 32 rows, 160 workgroups at batch 5120, one per CU):
//   LDS 160 KiB = X  [32][512] bf16   current activations / gradients of the rows, updated IN PLACE by every layer (32 KiB)
//                 R  [32][512] bf16   residual stream (forward) / its gradient (backward); loss-phase scratch       (32 KiB)
//                 ring 3 x [256 out-ch][64 k] bf16 weight stages                                                   (96 KiB)
//   waves 4..7  only issue LDS-DMA (global_load_lds_dwordx4): the gathered input rows, then the weight stages of ALL layers as
//               one flat stream of 16 stages per layer (8 K-steps x 2 output halves), two stages ahead of the multipliers,
//               across layer boundaries and across the loss phase -- weights do not depend on the rows;
//   waves 0..3  multiply (v_mfma_f32_16x16x32_bf16, wave w owns output channels {64w..64w+63} + {256+64w..}, all 32 rows:
//               16 accumulator tiles) and run the epilogues: bias/ReLU/residual (forward), residual-gradient add, ReLU mask,
//               bias-gradient column sums (backward) -- the same rounding points as rowgemm80_kernel, and the same K order,
//               so activations and gradients are bitwise those of the per-layer launches.
//   ReLU masks are 64 bits per lane and layer (same lane <-> (row, channel) map in both directions), parked in a global scratch.
//   What leaves the workgroup for HBM: the layer inputs and dZ tiles the weight-gradient kernel needs (copied out of X in
//   full 1 KiB rows while the next layer multiplies), the per-workgroup partials of the bias / fc3 gradients and statistics.
//
// Bound: each workgroup streams all 512 KiB of every layer's weights L2 -> LDS (15 layers: 7.5 MiB), i.e. the per-CU
// LDS-DMA fill rate, against 1.7 us of MFMA time per layer. The 2-D tiled launches move 2.5x fewer bytes per CU but need an
// all-to-all between layers (a kernel boundary or an in-launch hand-off, both > 2.5 us).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "head_kernels.h"

namespace acez {

constexpr int CHAIN_ROWS = 32;
constexpr int CHAIN_STAGE = 256 * 64;   // elements of one ring slot
constexpr int CHAIN_THREADS = 512;

struct ChainStep {
  const uint16_t* W;      // [512][512] bf16: W (forward) or W^T (input gradient) of the layer
  const float* bias;      // forward: fp32 bias
  uint16_t* g_out;        // [n][512] copy of the step's output tile in HBM, or null
  int residual;           // forward: R += out; the next input is R  (ace_network.py:126,133)
  int add;                // backward: the residual gradient R is added before rounding
  int aux;                // backward: R = the unmasked rounded sum (gradient of the residual stream)
  int mask_layer;         // forward: ReLU mask bits of this layer are written (-1: none); backward: they are applied
  int bias_slot;          // backward: layer whose bias-gradient partial this step produces
};

struct ChainArgs {
  const uint16_t* src;    // phases & 1: feature buffer, rows gathered through idx; else the fc2 output [n][512]
  const int64_t* idx;
  uint16_t* g_in;         // [n][512] copy of the gathered rows (the first layer's weight gradient needs them)
  uint2* maskbits;        // [layers][workgroups][256]
  float* bias_partials;   // [layers][bias_layer_stride], one [512] row per workgroup
  int64_t bias_layer_stride;
  int n, n_fwd, n_bwd;
  int phases;             // 1 = gather + forward, 2 = loss + backward, 3 = everything
  const TrainState* st;
  LossArgs loss;
  ChainStep step[2 * MAX_LAYERS - 1];
};

__device__ __forceinline__ void chain_dma(const uint16_t* g, uint16_t* lds) {
  __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)lds, 16, 0, 0);
}
// bit e set <=> bf16 element e of the packed quadruple is > 0 (sign clear, magnitude non-zero): the ReLU mask of rowgemm_kernel
__device__ __forceinline__ uint32_t pos4(uint2 y) {
  const uint32_t a = y.x & 0xffffu, b = y.x >> 16, c = y.y & 0xffffu, d = y.y >> 16;
  return (uint32_t)(a != 0 && a < 0x8000u) | ((uint32_t)(b != 0 && b < 0x8000u) << 1) | ((uint32_t)(c != 0 && c < 0x8000u) << 2) |
         ((uint32_t)(d != 0 && d < 0x8000u) << 3);
}

__global__ __launch_bounds__(CHAIN_THREADS) void chain_kernel(ChainArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[2 * CHAIN_ROWS * 512 + 3 * CHAIN_STAGE];   // 160 KiB
  uint16_t* const X = smem;
  uint16_t* const R = smem + CHAIN_ROWS * 512;
  uint16_t* const ring = smem + 2 * CHAIN_ROWS * 512;
  if (a.st && !a.st->active) return;   // schedule ended (ace_trainer.py:509-510): nothing is written; uniform for the grid
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m0 = blockIdx.x * CHAIN_ROWS, n = a.n;
  const bool fwd_phase = (a.phases & 1) != 0, bwd_phase = (a.phases & 2) != 0;
  const int s_begin = fwd_phase ? 0 : a.n_fwd;
  const int nsteps = (fwd_phase ? a.n_fwd : 0) + (bwd_phase ? a.n_bwd : 0);
  const bool loss_first = !fwd_phase;                              // loss + backward only: X is loaded with the fc2 output
  const int loss_after = (fwd_phase && bwd_phase) ? a.n_fwd - 1 : -1;   // step after whose epilogue the loss phases run
  const int G = nsteps * 16;

  // Barrier protocol (all eight waves execute the same sequence of s_barrier):
  //   [loss_first: LB0, LB1]  then per step: 16 stage barriers, E1 (everyone is done reading X; the epilogue overwrites it),
  //   [after step loss_after: LB0 (X = fc2 output complete), LB1 (inside loss_body)],  finally one barrier (last X complete).
  if (w >= 4) {
    // ---------------------------------------------------------------------------------------------- loader waves
    const int lw = w - 4;
    // input rows 8 lw .. 8 lw + 7: one 1 KiB row per DMA instruction; physical 16-byte chunk l receives logical chunk l ^ (row & 15)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = lw * 8 + r;
      const int m = min(m0 + row, n - 1);   // rows past the end compute on a copy of the last row and are never stored
      const int64_t srow = (fwd_phase && a.idx) ? a.idx[m] : (int64_t)m;
      const uint16_t* g = a.src + srow * 512 + ((l ^ (row & 15)) << 3);
      chain_dma(g, X + row * 512);
      if (fwd_phase) chain_dma(g, R + row * 512);
    }
    // weight stage (kt, half) = rows half*256 .. +255, columns kt*64 .. +63 of the step's matrix; this loader moves the 8-row
    // groups q = 8 lw + j; lane: row q*8 + (l >> 3), physical chunk l & 7 <- logical chunk (l & 7) ^ ((row >> 1) & 7)  (swz)
    int woff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = (lw * 8 + j) * 8 + (l >> 3);
      woff[j] = row * 512 + (((l & 7) ^ ((row >> 1) & 7)) << 3);
    }
    auto issue = [&](int g, int slot) {
      const int s = s_begin + (g >> 4), kt = (g & 15) >> 1, half = g & 1;
      const uint16_t* Wl = a.step[s].W + half * (256 * 512) + kt * 64;
      uint16_t* dst = ring + slot * CHAIN_STAGE + lw * (8 * 8 * 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) chain_dma(Wl + woff[j], dst + j * (8 * 64));
    };
    issue(0, 0);
    issue(1, 1);
    if (loss_first) {
      ACEZ_VMCNT(16);                    // the input tile has landed (in-order completion; two stages may stay in flight)
      __builtin_amdgcn_s_barrier();      // LB0
      __builtin_amdgcn_s_barrier();      // LB1
    }
    int slot_next = 2;
    for (int g = 0; g < G; ++g) {
      if (g + 1 < G) ACEZ_VMCNT(8);      // stage g (and everything older) has landed; stage g + 1 may stay in flight
      else ACEZ_VMCNT(0);
      __builtin_amdgcn_s_barrier();      // ... for every loader; the multipliers are done with stage g - 1
      if (g + 2 < G) {
        issue(g + 2, slot_next);         // into the slot of stage g - 1
        slot_next = (slot_next == 2) ? 0 : slot_next + 1;
      }
      if ((g & 15) == 15) {
        __builtin_amdgcn_s_barrier();    // E1
        if ((g >> 4) == loss_after) {
          __builtin_amdgcn_s_barrier();  // LB0
          __builtin_amdgcn_s_barrier();  // LB1
        }
      }
    }
    __builtin_amdgcn_s_barrier();        // last X complete
    return;
  }

  // ------------------------------------------------------------------------------------------------ multiplier waves
  const int fr = l & 15, fq = l >> 4;
  LossPre pre{0, 0, 0, 0.f, 0.f};
  if (bwd_phase) pre = loss_prefetch(a.loss, m0 + w * LOSS_ROWS, l);

  // copy the complete tile X to HBM in full 1 KiB rows (wave w: rows 8w .. 8w+7)
  auto copy_tile = [&](uint16_t* g) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // two batches of four rows: eight 16-byte values in flight spill registers
      uint4 v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = w * 8 + h * 4 + r;
        v[r] = *reinterpret_cast<const uint4*>(&X[row * 512 + ((l ^ (row & 15)) << 3)]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = w * 8 + h * 4 + r;
        if (m0 + row < n) *reinterpret_cast<uint4*>(g + (size_t)(m0 + row) * 512 + l * 8) = v[r];
      }
    }
  };
  // one weight stage: 2 K-chunks of 32 x (4 channel tiles x 2 row tiles); `ac` = the accumulators of the stage's output half
  auto stage = [&](int kt, f32x4 (&ac)[4][2], int slot) {
    const uint16_t* sW = ring + slot * CHAIN_STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + fq;
      bf16x8 fa[4], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(&X[(j * 16 + fr) * 512 + (((kt * 8 + c) ^ fr) << 3)]);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(&sW[swz(w * 64 + i * 16 + fr, c)]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], ac[i][j], 0, 0, 0);
    }
  };

  uint16_t* copy_dst = fwd_phase ? a.g_in : nullptr;
  if (loss_first) {
    __builtin_amdgcn_s_barrier();   // LB0: the fc2 output tile has landed
    loss_body<true>(a.loss, blockIdx.x, w, l, X, reinterpret_cast<float*>(R), pre);   // contains LB1; dZ -> X and HBM
  }
  int slot = 0;
  for (int si = 0; si < nsteps; ++si) {
    const ChainStep& S = a.step[s_begin + si];
    const bool is_fwd = (s_begin + si) < a.n_fwd;
    f32x4 acc[2][4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[h][i][j][r] = 0.f;
    float4 bias[2][4];
    uint2 mbits = make_uint2(0u, 0u);
    for (int kt = 0; kt < 8; ++kt) {
      __builtin_amdgcn_s_barrier();
      if (kt == 0) {
        // X is complete and read-only until E1: copy the previous step's output out, fetch this step's epilogue inputs
        if (copy_dst) copy_tile(copy_dst);
        if (is_fwd) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) bias[h][i] = *reinterpret_cast<const float4*>(S.bias + h * 256 + w * 64 + i * 16 + 4 * fq);
        } else {
          mbits = a.maskbits[((size_t)S.mask_layer * gridDim.x + blockIdx.x) * 256 + w * 64 + l];
        }
      }
      stage(kt, acc[0], slot);
      slot = (slot == 2) ? 0 : slot + 1;
      __builtin_amdgcn_s_barrier();
      stage(kt, acc[1], slot);
      slot = (slot == 2) ? 0 : slot + 1;
    }
    __builtin_amdgcn_s_barrier();   // E1
    if (is_fwd) {
      uint32_t bits[2] = {0u, 0u};
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int off = act_off(j * 16 + fr, h * 256 + w * 64 + i * 16 + 4 * fq);
            const float4 b = bias[h][i];
            const float v0 = fmaxf(acc[h][i][j][0] + b.x, 0.f), v1 = fmaxf(acc[h][i][j][1] + b.y, 0.f);
            const float v2 = fmaxf(acc[h][i][j][2] + b.z, 0.f), v3 = fmaxf(acc[h][i][j][3] + b.w, 0.f);
            const uint2 y = pack4(v0, v1, v2, v3);
            bits[h] |= pos4(y) << ((i * 2 + j) * 4);
            if (S.residual) {
              // R = bf16( float(bf16(y)) + float(R) ), and the next layer reads R   (ace_network.py:126,133)
              float yf[4], rf[4];
              unpack4(y, yf);
              unpack4(*reinterpret_cast<const uint2*>(&R[off]), rf);
              const uint2 r = pack4(yf[0] + rf[0], yf[1] + rf[1], yf[2] + rf[2], yf[3] + rf[3]);
              *reinterpret_cast<uint2*>(&R[off]) = r;
              *reinterpret_cast<uint2*>(&X[off]) = r;
            } else {
              *reinterpret_cast<uint2*>(&X[off]) = y;
            }
          }
      if (S.mask_layer >= 0) a.maskbits[((size_t)S.mask_layer * gridDim.x + blockIdx.x) * 256 + w * 64 + l] = make_uint2(bits[0], bits[1]);
    } else {
      const uint32_t mb[2] = {mbits.x, mbits.y};
      float cs[2][4][4];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[h][i][e] = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int off = act_off(j * 16 + fr, h * 256 + w * 64 + i * 16 + 4 * fq);
            float v[4] = {acc[h][i][j][0], acc[h][i][j][1], acc[h][i][j][2], acc[h][i][j][3]};
            if (S.add) {
              float ad[4];
              unpack4(*reinterpret_cast<const uint2*>(&R[off]), ad);
              v[0] += ad[0]; v[1] += ad[1]; v[2] += ad[2]; v[3] += ad[3];
            }
            uint2 y = pack4(v[0], v[1], v[2], v[3]);
            if (S.aux) *reinterpret_cast<uint2*>(&R[off]) = y;
            // relu backward: keep the gradient where the forward activation was > 0
            const uint32_t b = (mb[h] >> ((i * 2 + j) * 4)) & 15u;
            if (!(b & 1u)) y.x &= 0xffff0000u;
            if (!(b & 2u)) y.x &= 0x0000ffffu;
            if (!(b & 4u)) y.y &= 0xffff0000u;
            if (!(b & 8u)) y.y &= 0x0000ffffu;
            *reinterpret_cast<uint2*>(&X[off]) = y;
            float q[4];
            unpack4(y, q);
#pragma unroll
            for (int e = 0; e < 4; ++e) cs[h][i][e] += q[e];   // bias gradient: the bf16-rounded values, rows fr then fr + 16
          }
        }
      // column sums over the workgroup's 32 rows: butterfly over the 16 lanes that share fq (fixed order -> deterministic)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) cs[h][i][e] += __shfl_xor(cs[h][i][e], o);
      if (fr == 0) {
        float* bp = a.bias_partials + (size_t)S.bias_slot * a.bias_layer_stride + (size_t)blockIdx.x * 512;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(bp + h * 256 + w * 64 + i * 16 + 4 * fq) = make_float4(cs[h][i][0], cs[h][i][1], cs[h][i][2], cs[h][i][3]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile writes are complete before the next barrier
    copy_dst = S.g_out;
    if (si == loss_after) {
      __builtin_amdgcn_s_barrier();   // LB0: X = fc2 output, complete
      loss_body<true>(a.loss, blockIdx.x, w, l, X, reinterpret_cast<float*>(R), pre);   // contains LB1; dZ -> X and HBM
      copy_dst = nullptr;
    }
  }
  __builtin_amdgcn_s_barrier();       // last X complete
  if (copy_dst) copy_tile(copy_dst);
}

}  // namespace acez
