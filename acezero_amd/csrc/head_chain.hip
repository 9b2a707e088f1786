// head_chain.hip -- the training step's dependent chain as ONE launch: gather, the 8 forward layers, fc3 + loss + its
// gradient, and the 7 input-gradient layers (ace_trainer.py:485-613, ace_network.py:120-147 and their autograd).
//
// Why: as 1 + 8 + 1 + 7 dependent launches the chain cost 7.7 + 57.8 + 21.8 + 60.8 us per 5120-row step although it holds
// 16 us of MFMA work: every launch pays its boundary, the dispatch of 256 large-LDS workgroups, a cold operand pipeline and
// an epilogue whose stores must drain (round-1 profile). The rows of a batch are independent through the whole MLP and
// through the loss, so a workgroup that owns 32 rows can walk all of it without ever synchronising with another workgroup.
//
// Structure (one 768-thread workgroup per 32 rows, 160 workgroups at batch 5120, one per CU):
//   LDS 160 KiB = X    [32][512] bf16   current activations / gradients of the rows, updated IN PLACE by every layer   (32 KiB)
//                 ring 7 x [128 out-ch][64 k] bf16 weight stages                                                      (112 KiB)
//                 misc the loss phases' scratch                                                                        (16 KiB)
//   waves 8..11 (loaders) only issue LDS-DMA (global_load_lds_dwordx4): the gathered input rows, then the weight stages of ALL
//               layers as one flat stream of 32 stages per layer (8 K-steps x 2 sub-blocks x 2 output halves; every loader
//               moves a quarter of every stage), as far ahead of the multipliers as the ring allows (up to five stages in
//               flight), across layer boundaries and across the loss phase -- weights do not depend on the rows;
//   waves 0..7  (multipliers, v_mfma_f32_16x16x32_bf16): group h = w >> 2 owns output half h and consumes the stages of
//               parity h; wave w owns 2 x 32 output channels of all 32 rows (8 accumulator tiles) and runs the epilogues:
//               bias / ReLU / residual (forward), residual-gradient add, ReLU mask, bias-gradient column sums (backward) --
//               the same rounding points and the same K order as rowgemm80_kernel, so activations and gradients are bitwise
//               those of the per-layer launches.
//   Synchronisation is s_barrier only: one per stage (loaders: "stage g has landed", in-order vmcnt; multipliers: "done with
//   stage g - 1", whose slot is refilled with stage g + 6) and two per layer around the in-place epilogue. Measured on MI355X
//   (tools/chain_prims.hip): s_barrier 13-21 cycles with 8-16 waves arriving together, against 155 cycles for one LDS flag
//   poll, 250-500 for a flag hand-off and 1300-2600 for an all-to-all flag barrier -- a flag-based version of this kernel
//   (no barriers, sequence numbers in LDS) was built first and ran at 13 us per layer.
//   The residual stream (forward) / its gradient (backward) lives in HBM and is fetched into registers in accumulator
//   layout one layer ahead; ReLU masks are 32 bits per lane and layer (same lane <-> (row, channel) map in both directions),
//   parked in a global scratch. What leaves the workgroup for HBM: the layer inputs and dZ tiles the weight-gradient kernel
//   needs (copied out of X in full 1 KiB rows while the next layer multiplies), the per-workgroup partials of the bias / fc3
//   gradients and statistics.
//
// Bound: each workgroup streams all 512 KiB of every layer's weights L2 -> LDS (15 layers: 7.5 MiB), i.e. the per-CU
// LDS-DMA fill rate (measured 131 GB/s = 3.9 us per layer), against 1.7 us of MFMA time per layer. The 2-D tiled launches
// move 2.5x fewer bytes per CU but need an all-to-all between layers (a kernel boundary or an in-launch hand-off, > 2.5 us).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "head_kernels.h"

namespace acez {

constexpr int CHAIN_ROWS = 32;
constexpr int CHAIN_STAGE = 128 * 64;   // elements of one ring slot (16 KiB)
constexpr int CHAIN_RING = 8;           // ring slots (four pairs): 32 + 128 = all 160 KiB of LDS
constexpr int CHAIN_MULT = 8, CHAIN_LOAD = 4;
constexpr int CHAIN_THREADS = 64 * (CHAIN_MULT + CHAIN_LOAD);

struct ChainStep {
  const uint16_t* W;       // [512][512] bf16: W (forward) or W^T (input gradient) of the layer
  const float* bias;       // forward: fp32 bias
  uint16_t* g_out;         // [n][512] copy of the step's output tile in HBM, or null
  const uint16_t* r_in;    // [n][512] forward: residual stream added to the output; backward: residual gradient added; or null
  uint16_t* r_out;         // [n][512] backward: the unmasked rounded sum (gradient of the residual stream) is stored here; or null
  int mask_layer;          // forward: ReLU mask bits of this layer are written (-1: none); backward: they are applied
  int bias_slot;           // backward: layer whose bias-gradient partial this step produces
};

struct ChainArgs {
  const uint16_t* src;     // phases & 1: feature buffer, rows gathered through idx; else the fc2 output [n][512]
  const int64_t* idx;
  uint16_t* g_in;          // [n][512] copy of the gathered rows (the first layer's weight gradient and the residual need them)
  uint16_t* g_dz_last;     // [n][512] dZ of fc2 (output of the loss phases)
  uint32_t* maskbits;      // [layers][workgroups][512]
  float* bias_partials;    // [layers][bias_layer_stride], one [512] row per workgroup
  int64_t bias_layer_stride;
  int n, n_fwd, n_bwd;
  int phases;              // 1 = gather + forward, 2 = loss + backward, 3 = everything
  const TrainState* st;
  int* err;                // unused (kept for the host ABI of the trainer): the kernel has no spin loops
  unsigned long long* trace;   // null, or [2][256] s_memtime stamps of workgroup 0 (tools/chain_trace.py): row 0 = multiplier wave 0
                               // (per step: X ready, fetch issued, K loop done, everyone done, epilogue done), row 1 = loader 0
  int dbg;                 // ablation (tools/chain_timing.py, ACEZ_CHAIN_DBG): 1 = no MFMA / fragment reads, 2 = stages read from contiguous
                           // 16 KiB blocks (timing only), 4 = no weight DMA,
                           // 8 = no epilogue, 16 = no tile copies to HBM, 32 = no bias / mask / residual fetch
  LossArgs loss;
  ChainStep step[2 * MAX_LAYERS - 1];
};

__device__ __forceinline__ void chain_dma(const uint16_t* g, uint16_t* lds) {
  __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)lds, 16, 0, 0);
}
// bit e set <=> bf16 element e of the packed quadruple of ReLU outputs (never negative) is > 0: magnitude bits non-zero
// (+0 / -0 excluded, NaN included -- the predicate "m != 0 && m < 0x8000" of rowgemm_kernel's mask on such values)
__device__ __forceinline__ uint32_t pos4_relu(uint2 y) {
  return (uint32_t)((y.x & 0x7fffu) != 0) | ((uint32_t)((y.x & 0x7fff0000u) != 0) << 1) | ((uint32_t)((y.y & 0x7fffu) != 0) << 2) |
         ((uint32_t)((y.y & 0x7fff0000u) != 0) << 3);
}

// ---- sequence flags in LDS (ints). A signal is one ds_write by lane 0 after the wave's own LDS / DMA work is complete; a wait
// is a broadcast ds_read in a loop. Plain volatile accesses + compiler barriers: the LDS executes a wave's operations in
// order, and a C++ release / acquire here would drain vmcnt (the loaders' DMA pipeline, the multipliers' tile stores).
__device__ __forceinline__ void chain_loss(const LossArgs& a, int block, int wv, int t, uint16_t* Xt, float* scratch, int64_t pre_p, int pre_view,
                                           int pre_img, float pre_tu, float pre_tv) {
  loss_body<true, false>(a, block, wv, t, Xt, scratch, LossPre{pre_p, pre_view, pre_img, pre_tu, pre_tv}, [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's ds rows are written
    __builtin_amdgcn_s_barrier();                        // LB1
  });
}

// DBG = false is the production instantiation: the ablation switches of ChainArgs::dbg are compiled out of the per-stage paths
// (with a barrier per stage pair every scalar branch in them is on the critical path).
template <bool DBG>
__global__ __launch_bounds__(CHAIN_THREADS) void chain_kernel(ChainArgs a) {
  const int dbg = DBG ? a.dbg : 0;
  __shared__ __attribute__((aligned(16))) uint16_t smem[CHAIN_ROWS * 512 + CHAIN_RING * CHAIN_STAGE];   // 160 KiB
  uint16_t* const X = smem;
  uint16_t* const ring = smem + CHAIN_ROWS * 512;
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m0 = blockIdx.x * CHAIN_ROWS, n = a.n;
  const bool fwd_phase = (a.phases & 1) != 0, bwd_phase = (a.phases & 2) != 0;
  const int s_begin = fwd_phase ? 0 : a.n_fwd;
  const int nsteps = (fwd_phase ? a.n_fwd : 0) + (bwd_phase ? a.n_bwd : 0);
  const bool loss_first = !fwd_phase;                                   // loss + backward only: X is loaded with the fc2 output
  const int loss_after = (fwd_phase && bwd_phase) ? a.n_fwd - 1 : -1;   // step after whose epilogue the loss phases run
  // the loaders fetch their gather indices before the schedule flag is looked at (two dependent HBM round trips otherwise)
  int64_t srow[8];
  if (w >= CHAIN_MULT) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = min(m0 + (w - CHAIN_MULT) * 8 + r, n - 1);   // rows past the end compute on a copy of the last row and are never stored
      srow[r] = (fwd_phase && a.idx) ? a.idx[m] : (int64_t)m;
    }
  }
  if (a.st && !a.st->active) return;   // schedule ended (ace_trainer.py:509-510): nothing is written; uniform for the grid

  // Barrier protocol -- all twelve waves execute the same sequence of s_barrier (13-21 cycles each when the waves arrive
  // together, tools/chain_prims.hip; an LDS-flag hand-off costs 250-500 cycles and an 8-wave flag barrier 1300):
  //   [loss_first: LB0, LB1]  then per step: 32 stage barriers (stage g has landed / everyone is done with stage g - 1),
  //   E1 (everyone is done reading X; the epilogue overwrites it), XC (X holds the next input),
  //   [after step loss_after: LB1 inside loss_body, XC again].
  if (w >= CHAIN_MULT) {
    // ---------------------------------------------------------------------------------------------- loader wave c
    const int c = w - CHAIN_MULT;
    // input rows 8c .. 8c+7: one 1 KiB row per DMA instruction; physical 16-byte chunk l receives logical chunk l ^ (row & 15)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = c * 8 + r;
      chain_dma(a.src + srow[r] * 512 + ((l ^ (row & 15)) << 3), X + row * 512);
    }
    // stage g (global index over the whole launch): step g >> 5, K-step kt = (g & 31) >> 2, sub-block (g >> 1) & 1 of output
    // half g & 1: rows 256 h + 128 sub .. + 127, columns 64 kt .. + 63 of the step's matrix, as [128][64] with the swz() chunk
    // swizzle. This loader moves rows 32 c .. 32 c + 31 of every stage: DMA instruction j covers rows 32 c + 8 j .. + 7; lane: row
    // + (l >> 3), physical chunk l & 7 <- logical chunk (l & 7) ^ ((row >> 1) & 7)
    int woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = c * 32 + j * 8 + (l >> 3);
      woff[j] = row * 512 + (((l & 7) ^ ((row >> 1) & 7)) << 3);
    }
    const int G = nsteps * 32;
    // The per-stage path of a loader is kept to a few dozen instructions: a wave issues about one instruction per 4-5 cycles,
    // and with a barrier per stage the slowest wave's path IS the stage time (a first version with ~150 instructions of
    // bookkeeping per stage -- if-chains for the wait count, 64-bit pointer arithmetic, ablation switches -- cost 5 us per layer).
    const uint16_t* Wcur = a.step[s_begin].W;                                  // matrix of the step whose stages are being issued
    const uint16_t* Wnext = (nsteps > 1) ? a.step[s_begin + 1].W : Wcur;      // ... of the step after it (fetched a step ahead)
    int qi = 0;                    // stage index inside the step being issued
    int gi = 0;                    // stages issued so far
    uint16_t* dst = ring + c * (32 * 64);   // this loader's quarter of the next slot to fill
    int slot_i = 0;
    const bool no_dma = (dbg & 4) != 0;
    auto issue = [&]() {
      if (!no_dma) {
        // rows 256 h + 128 sub, columns 64 kt of the matrix: h = qi & 1, sub = (qi >> 1) & 1, kt = qi >> 2
        const uint16_t* Wl = Wcur + (((qi & 1) << 17) + ((qi & 2) << 15) + ((qi >> 2) << 6));
#pragma unroll
        for (int j = 0; j < 4; ++j) chain_dma(Wl + woff[j], dst + j * (8 * 64));
      }
      ++gi;
      if (++qi == 32) {
        qi = 0;
        Wcur = Wnext;
        const int nx = (gi >> 5) + 1;
        if (nx < nsteps) Wnext = a.step[s_begin + nx].W;
      }
      if (++slot_i == CHAIN_RING) { slot_i = 0; dst -= (CHAIN_RING - 1) * CHAIN_STAGE; }
      else dst += CHAIN_STAGE;
    };
    // Stages are consumed in PAIRS (2p for group 0, 2p + 1 for group 1, one barrier per pair). Barrier p certifies pairs p AND
    // p + 1 (the multipliers fetch the W fragments of pair p + 1 while they multiply pair p), pair p + 2 is in flight and pair
    // p + 3 is issued into the slots pair p - 1 has just left: a ring of four pairs = 8 x 16 KiB. The loss phases' 1.5 KiB of
    // scratch live in the one pair position that is free while they run (the pair after the two prefetched ones).
#pragma unroll 1
    for (int i = 0; i < 6; ++i) issue();   // pairs 0, 1, 2  (G >= 32)
    if (loss_first) {
      ACEZ_VMCNT(24);                    // the input tile has landed (in-order completion; six stages may stay in flight)
      __builtin_amdgcn_s_barrier();      // LB0: X complete
      __builtin_amdgcn_s_barrier();      // LB1
      __builtin_amdgcn_s_barrier();      // XC
    }
#pragma unroll 1
    for (int si = 0; si < nsteps; ++si) {
      const int ntail = (si + 1 == nsteps) ? 2 : 0;   // the last two pairs of the launch have nothing in flight behind them
#pragma unroll 1
      for (int q = 0; q < 16 - ntail; ++q) {
        if (!(dbg & 512)) ACEZ_VMCNT(8);   // pairs p, p + 1 and everything older have landed: pair p + 2 may be in flight
        __builtin_amdgcn_s_barrier();    // ... for every loader; the multipliers are done with pair p - 1
        if (gi < G) { issue(); issue(); }
      }
      if (ntail) {
        ACEZ_VMCNT(0); __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_s_barrier();      // E1
      __builtin_amdgcn_s_barrier();      // XC
      if (si == loss_after) {
        __builtin_amdgcn_s_barrier();    // LB1
        __builtin_amdgcn_s_barrier();    // XC
      }
    }
    return;
  }

  // ------------------------------------------------------------------------------------------------ multiplier wave w
  const int fr = l & 15, fq = l >> 4;
  const int gh = w >> 2, ww = w & 3;   // output half (= stage parity) of this wave's group, position in the group
  // accumulator tile (sub, i, j): rows 16 j + fr, channels ch0(sub, i) + 4 fq .. + 3
  auto ch0 = [&](int sub, int i) { return gh * 256 + sub * 128 + ww * 32 + i * 16; };
  LossPre pre{0, 0, 0, 0.f, 0.f};
  if (bwd_phase && w < 4) pre = loss_prefetch(a.loss, m0 + w * LOSS_ROWS, l);

  // copy the complete tile X to HBM in full 1 KiB rows (wave w: rows 4w .. 4w+3), two rows at a time
  auto copy_tile = [&](uint16_t* g) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int r0 = w * 4 + h2 * 2, r1 = r0 + 1;
      const uint4 v0 = *reinterpret_cast<const uint4*>(&X[r0 * 512 + ((l ^ (r0 & 15)) << 3)]);
      const uint4 v1 = *reinterpret_cast<const uint4*>(&X[r1 * 512 + ((l ^ (r1 & 15)) << 3)]);
      if (m0 + r0 < n) *reinterpret_cast<uint4*>(g + (size_t)(m0 + r0) * 512 + l * 8) = v0;
      if (m0 + r1 < n) *reinterpret_cast<uint4*>(g + (size_t)(m0 + r1) * 512 + l * 8) = v1;
    }
  };
  auto run_loss = [&](float* loss_scratch) {   // contains LB1 (every wave must execute it: waves 4..7 only take part in the barrier)
    if (w < 4) chain_loss(a.loss, blockIdx.x, w, l, X, loss_scratch, pre.p, pre.view, pre.img, pre.tu, pre.tv);
    else __builtin_amdgcn_s_barrier();
  };

  int tp = 0;
  auto stamp = [&]() {
    if (a.trace && blockIdx.x == 0 && w == 0 && l == 0 && tp < 256) a.trace[tp++] = __builtin_amdgcn_s_memtime();
  };
  stamp();
  // per-lane LDS offsets (elements), constant for the whole launch
  int xb[2], xx[2], wo[2][2], eo[2][2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) xb[j] = (j * 16 + fr) * 512;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    xx[kk] = ((kk * 4 + fq) ^ fr) << 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) wo[kk][i] = swz(ww * 32 + i * 16 + fr, kk * 4 + fq);
  }
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) eo[sub][i][j] = act_off(j * 16 + fr, ch0(sub, i) + 4 * fq);
  int pair = 0;                                    // ring position (in pairs) of the next pair whose W fragments are fetched
  const uint16_t* sWp = ring + gh * CHAIN_STAGE;   // ... and this group's stage of it
  bf16x8 faA[2][2], faB[2][2];                     // W fragments of the pair being multiplied / the one after it
  bool have_fa = false;
  uint16_t* copy_dst = fwd_phase ? a.g_in : nullptr;
  if (loss_first) {
    __builtin_amdgcn_s_barrier();   // LB0: the fc2 output tile has landed
    run_loss(reinterpret_cast<float*>(ring + 6 * CHAIN_STAGE));   // pairs 0, 1, 2 are in flight: the fourth pair position is free
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // XC
    copy_dst = a.g_dz_last;
  }
  for (int si = 0; si < nsteps; ++si) {
    const ChainStep S = a.step[s_begin + si];   // one wide scalar load per step
    const bool is_fwd = (s_begin + si) < a.n_fwd;
    f32x4 acc[2][2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[sub][i][j][r] = 0.f;
    float4 bias[2][2];
    uint2 rin[2][2][2];
    uint32_t mbits = 0u;
    stamp();
    // One stage of this wave's group: 2 K-chunks of 32 x (2 channel tiles x 2 row tiles). The X fragments of a K-step serve both
    // of its stages (sub = 0, 1) and are fetched one K-step ahead (X does not change inside a layer), so that only the four W
    // fragment reads sit between a barrier and its MFMAs. LDS addresses: X fragment (kk, j) at xb[j] + ((kt << 6) ^ xx[kk]) (the
    // act_off swizzle with the K-step folded in), W fragment (kk, i) at the stage base + wo[kk][i].
    auto load_x = [&](int kt, bf16x8 (&fb)[2][2]) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int xo = (kt << 6) ^ xx[kk];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[kk][j] = *reinterpret_cast<const bf16x8*>(&X[xb[j] + xo]);
      }
    };
    auto load_w = [&](bf16x8 (&fa)[2][2]) {   // the W fragments of the next pair (this group's stage of it), then advance
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[kk][i] = *reinterpret_cast<const bf16x8*>(&sWp[wo[kk][i]]);
      if (++pair == CHAIN_RING / 2) { pair = 0; sWp = ring + gh * CHAIN_STAGE; }
      else sWp += 2 * CHAIN_STAGE;
    };
    auto stage = [&](f32x4 (&ac)[2][2], const bf16x8 (&fa)[2][2], const bf16x8 (&fb)[2][2]) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][i], fb[kk][j], ac[i][j], 0, 0, 0);
    };
    auto fetch_epilogue_inputs = [&]() {
      // accumulator layout; fetched two K-steps before the epilogue so that they do not occupy registers through the whole K loop
      if (dbg & 32) return;
      if (is_fwd) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int i = 0; i < 2; ++i) bias[sub][i] = *reinterpret_cast<const float4*>(S.bias + ch0(sub, i) + 4 * fq);
      } else {
        mbits = a.maskbits[((size_t)S.mask_layer * gridDim.x + blockIdx.x) * 512 + w * 64 + l];
      }
      if (S.r_in) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              rin[sub][i][j] = *reinterpret_cast<const uint2*>(S.r_in + (size_t)min(m0 + j * 16 + fr, n - 1) * 512 + ch0(sub, i) + 4 * fq);
      }
    };
    bf16x8 fb[2][2];
    const bool mm = !(dbg & 1);
#pragma unroll 1
    for (int kt = 0; kt < 8; ++kt) {
      __builtin_amdgcn_s_barrier();   // pairs (kt, 0) and (kt, 1) have landed; everyone is done with the pair before
      if (kt == 0) {
        // X is complete (XC; for the first step: the input rows have landed with the first pair) and read-only until E1
        if (!have_fa && mm) { load_w(faA); have_fa = true; }   // the very first pair of the launch: nobody could prefetch it
        if (mm) load_x(0, fb);
        if (copy_dst && !(dbg & 16)) copy_tile(copy_dst);   // the previous step's output
      }
      if (mm) { load_w(faB); stage(acc[0], faA, fb); }        // fetch pair (kt, 1) behind the MFMAs of pair (kt, 0)
      __builtin_amdgcn_s_barrier();   // pairs (kt, 1) and (kt + 1, 0) -- possibly the next step's first -- have landed
      if (mm) {
        if (!(kt == 7 && si + 1 == nsteps)) load_w(faA);
        stage(acc[1], faB, fb);
        if (kt < 7) load_x(kt + 1, fb);   // behind this K-step's MFMAs: lands while they execute
      }
      if (kt == 5) fetch_epilogue_inputs();
    }
    stamp();
    // This wave's HBM stores of the step's start (the tile copy) and of the previous epilogue (residual gradient, mask bits) are
    // complete before E1: other waves of the workgroup re-read them from HBM layers later. (Waiting here, not after the epilogue's
    // own stores, keeps their latency off the critical path; the loads above are needed now anyway.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // E1: everyone is done reading X, the epilogue may overwrite it
    stamp();
    if (dbg & 8) {
    } else if (is_fwd) {
      uint32_t bits = 0u;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int off = eo[sub][i][j];
            const float4 b = bias[sub][i];
            const float v0 = fmaxf(acc[sub][i][j][0] + b.x, 0.f), v1 = fmaxf(acc[sub][i][j][1] + b.y, 0.f);
            const float v2 = fmaxf(acc[sub][i][j][2] + b.z, 0.f), v3 = fmaxf(acc[sub][i][j][3] + b.w, 0.f);
            uint2 y = pack4(v0, v1, v2, v3);
            bits |= pos4_relu(y) << (((sub * 2 + i) * 2 + j) * 4);
            if (S.r_in) {
              // the residual stream continues as bf16( float(bf16(y)) + float(R) ), and the next layer reads it  (ace_network.py:126,133)
              float yf[4], rf[4];
              unpack4(y, yf);
              unpack4(rin[sub][i][j], rf);
              y = pack4(yf[0] + rf[0], yf[1] + rf[1], yf[2] + rf[2], yf[3] + rf[3]);
            }
            *reinterpret_cast<uint2*>(&X[off]) = y;
          }
      if (S.mask_layer >= 0) a.maskbits[((size_t)S.mask_layer * gridDim.x + blockIdx.x) * 512 + w * 64 + l] = bits;
    } else {
      float* bp = a.bias_partials + (size_t)S.bias_slot * a.bias_layer_stride + (size_t)blockIdx.x * 512;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int off = eo[sub][i][j];
            const bool live = m0 + j * 16 + fr < n;   // rows past the end stay zero rows in the backward pass
            float v[4] = {acc[sub][i][j][0], acc[sub][i][j][1], acc[sub][i][j][2], acc[sub][i][j][3]};
            if (S.r_in && live) {
              float ad[4];
              unpack4(rin[sub][i][j], ad);
              v[0] += ad[0]; v[1] += ad[1]; v[2] += ad[2]; v[3] += ad[3];
            }
            uint2 y = pack4(v[0], v[1], v[2], v[3]);
            if (S.r_out && live)   // gradient of the residual stream: the unmasked rounded sum
              *reinterpret_cast<uint2*>(S.r_out + (size_t)(m0 + j * 16 + fr) * 512 + ch0(sub, i) + 4 * fq) = y;
            // relu backward: keep the gradient where the forward activation was > 0
            const uint32_t b = (mbits >> (((sub * 2 + i) * 2 + j) * 4)) & 15u;
            if (!(b & 1u)) y.x &= 0xffff0000u;
            if (!(b & 2u)) y.x &= 0x0000ffffu;
            if (!(b & 4u)) y.y &= 0xffff0000u;
            if (!(b & 8u)) y.y &= 0x0000ffffu;
            *reinterpret_cast<uint2*>(&X[off]) = y;
            float q[4];
            unpack4(y, q);
#pragma unroll
            for (int e = 0; e < 4; ++e) cs[e] += q[e];   // bias gradient: the bf16-rounded values, rows fr then fr + 16
          }
          // column sums over the workgroup's 32 rows: the 16 lanes that share fq
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[e] = row16_sum(cs[e]);
          if (fr == 0) *reinterpret_cast<float4*>(bp + ch0(sub, i) + 4 * fq) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile writes are complete
    __builtin_amdgcn_s_barrier();   // XC: X holds the next input
    stamp();
    copy_dst = S.g_out;
    if (si == loss_after) {
      // resident or in flight now: the next step's pairs 0 (its W fragments are already fetched: `pair` points behind it), 1 and
      // 2; the position after them is free until the next step's first barrier
      run_loss(reinterpret_cast<float*>(ring + ((pair + 2) & 3) * 2 * CHAIN_STAGE));   // contains LB1; dZ -> X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); // XC
      copy_dst = a.g_dz_last;
    }
  }
  if (copy_dst) copy_tile(copy_dst);
}

}  // namespace acez
