// head_kernels.hip -- gfx950 kernels of the ACE scene-coordinate head (training + inference).
//
// What is computed (reference file:line):
//   gather                ace_trainer.py:485-494   training_buffer['features'][random_batch_indices]
//   rowgemm (fwd)         ace_network.py:122-136   relu(conv1x1(x)) chains, residual adds at :126,:133
//   loss kernel           ace_network.py:137-147   fc3 + softplus de-homogenisation + mean
//                         ace_trainer.py:521-613   projection, masks, ReproLoss (ace_loss.py:39-91), loss/B
//                         + the analytic gradient of all of it wrt the fc3 outputs
//   rowgemm (dgrad)       autograd of the 1x1 convs wrt their inputs (+ relu masks, residual fan-in)
//   wgrad                 autograd of the 1x1 convs wrt weight and bias, all layers in one launch
//   grad_reduce / adamw   ace_schedule.py:106-126 (AdamW step), torch.optim.AdamW semantics
//   sched                 ace_schedule.py:72-101,115-126 (cool-down trigger, LinearLR / OneCycleLR)
//
// Data layout: activations are row-major [rows][512] bf16 (a row = one patch = 1 KiB); weights are kept as
// fp32 masters plus two bf16 compute copies, W [out][in] and W^T [in][out], so that the forward GEMM and the
// input-gradient GEMM both see a reduction-contiguous second operand. MFMA: v_mfma_f32_32x32x16_bf16, fp32
// accumulate, computed as D[i = output channel][j = row] so that each lane ends up holding 4 consecutive
// channels of one row (an 8-byte row-major store).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "head_kernels.h"
#include "gemm_common.h"

namespace acez {


// ---------------------------------------------------------------------------------------------------
// gather: out[r][:] = features[idx[r]][:]   (one wave copies one 1 KiB row per instruction)
// ---------------------------------------------------------------------------------------------------
// Rows wave, wave + nwaves, ... of a batch, GR at a time: the loads of a level are issued for all GR rows before anything of the next
// level (one row after the other, every row paid its own idx -> row round trips; two at a time -- round 3 -- the optimiser launch's 1720
// gather waves still walked their three rows in two passes of three dependent levels each). With meta.dst the per-row metadata the
// loss kernel needs (GatherMeta) is looked up beside the copy: a third level for the image index, overlapped with the row stores.
template <int GR = 4>
__device__ __forceinline__ void gather_rows(const uint16_t* __restrict__ feat, const int64_t* __restrict__ idx, uint16_t* __restrict__ out, int n,
                                            int wave, int nwaves, int lane, const GatherMeta& meta) {
  // (meta.hold: the trainer's sticky fault word, a wave-uniform load that travels with the first level and is only looked at in front of
  // the stores; while it is set nothing is stored)
  const bool held = meta.hold && *meta.hold;
  for (int r0 = wave; r0 < n; r0 += GR * nwaves) {
    int r[GR];
    bool ok[GR];
    int64_t s[GR];
#pragma unroll
    for (int j = 0; j < GR; ++j) { r[j] = r0 + j * nwaves; ok[j] = r[j] < n; s[j] = idx[ok[j] ? r[j] : r0]; }
    uint4 v[GR];
#pragma unroll
    for (int j = 0; j < GR; ++j) v[j] = *reinterpret_cast<const uint4*>(feat + s[j] * 512 + lane * 8);
    int view[GR];
    float2 tp[GR];
    if (meta.dst) {
#pragma unroll
      for (int j = 0; j < GR; ++j) { view[j] = meta.view_idx[s[j]]; tp[j] = *reinterpret_cast<const float2*>(meta.target_px + s[j] * 2); }
    }
#pragma unroll
    for (int j = 0; j < GR; ++j)
      if (ok[j] && !held) *reinterpret_cast<uint4*>(out + (size_t)r[j] * 512 + lane * 8) = v[j];
    if (meta.dst) {
      int img[GR];
#pragma unroll
      for (int j = 0; j < GR; ++j) img[j] = meta.view_image[view[j]];
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < GR; ++j)
          if (ok[j] && !held) meta.dst[r[j]] = make_int4(view[j], img[j], __float_as_int(tp[j].x), __float_as_int(tp[j].y));
      }
    }
  }
}

__global__ __launch_bounds__(256) void gather_kernel(const uint16_t* __restrict__ feat, const int64_t* __restrict__ idx,
                                                     uint16_t* __restrict__ out, int n, const TrainState* st, GatherMeta meta) {
  if (st && !st->active) return;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  gather_rows(feat, idx, out, n, wave, nwaves, lane, meta);
}

// ---------------------------------------------------------------------------------------------------
// rowgemm80: Out[m][n] = epi( sum_k In[m][k] * W[n][k] ), K = 512 (ACEZ_HEAD_CHANNELS) = 8 K-steps of 64, on 80-row x 128-column
// tiles so that a 5120-row batch gives 64 x 4 = 256 workgroups = one per CU (a 128 x 128 tiling -- round 1, in the git history --
// fills only 160 of the 256 CUs). MFMA v_mfma_f32_16x16x32_{bf16,f16}. LDS tiles are [row][64] 16-bit with the 16-byte chunk index
// XOR-swizzled by (row >> 1) & 7 (conflict-free ds_read_b128 fragment reads); rows past M are clamped (never stored).
// Eight waves with fixed roles (waves w and w + 4 share a SIMD):
//   waves 0..3  multiply: wave w owns output columns 32w .. 32w+31 (2 column fragments x 5 row fragments = 10
//               accumulator tiles) and never touches global memory inside the K loop;
//   waves 4..7  only issue LDS-DMA: the 8 K-stages [W 128 x 64 | In 96 x 64] into a 4-slot ring (112 KiB, 7 DMA
//               instructions per loader and stage) and then the epilogue's input tiles (`add`, `mask` / `res`) into two
//               [80][128] staging tiles, so that their latency is hidden behind the K loop too.
// A wave's instruction stream is in-order: when every wave both loaded and multiplied, the time its DMA instructions
// spent getting through the memory pipeline was added to its MFMA time.
// The epilogue exchanges the whole 80 x 128 tile through the staging tiles (16-byte chunk index XOR row & 15: conflict
// free both for the accumulator layout and for the row-wise copy), so every global access is a full 256-byte row.
// ---------------------------------------------------------------------------------------------------

// fp16 gradient-scale control: every kernel that stores propagated gradients folds the largest magnitude it produced (before the
// conversion, in scaled units) into the trainer's 64 striped words (TrainState::dz_absmax_slots) -- a wavefront maximum, then an atomic
// max on the bit pattern (non-negative floats order like unsigned integers; anything above 65504.f is the overflow signal, f16_overflow) of the word of
// this workgroup's stripe. Read (absmax_all) and reset by sched_post_wave.
// The magnitudes are taken BEFORE the conversion, so the overflow test is "larger than fp16's largest finite value", not "is inf": a
// propagated value of 7e4 is a finite float that the conversion stores as inf (round 6: a 25 000-iteration fp16 refit went through
// exactly that window once the scale had climbed to 2^16 -- the update was not skipped and NaN reached the fp32 masters).
__device__ __forceinline__ bool f16_overflow(uint32_t amax_bits) { return amax_bits > 0x477fe000u; }   // 65504.f; inf and NaN patterns are above it
__device__ __forceinline__ void absmax_publish(uint32_t* slots, float amax) {
  amax = wave_max63_nonneg(amax);   // (DPP: no LDS round trips; the maximum is in lane 63)
  if ((threadIdx.x & 63) == 63 && amax > 0.f) atomicMax(slots + (blockIdx.x & 63) * 32, __float_as_uint(amax));
}
// The maximum over 64 stripe words already in registers (one per lane, all 64 lanes call it, every lane returns the result). The bit
// patterns are non-negative floats (or +inf), which order like the unsigned integers they are: a DPP float maximum + one v_readlane, no LDS
// round trips (absmax_all's six ds_bpermute rounds sat between the K loop and the epilogue of every wave of wgrad_opt_kernel in fp16 mode).
__device__ __forceinline__ uint32_t absmax_reduce(uint32_t m) {
  const float f = wave_max63_nonneg(__uint_as_float(m));
  return (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(f), 63);
}
// the maximum over the 64 stripes; all 64 lanes of the wavefront must call it, every lane returns the result
__device__ __forceinline__ uint32_t absmax_all(const TrainState* st, int lane) {
  uint32_t m = __builtin_nontemporal_load(&st->dz_absmax_slots[lane * 32]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
  return m;
}

// One layer of one workgroup. SEQ = false: the whole of rowgemm80_kernel. SEQ = true: one link of rowseq_kernel (below), where
// the layers of a dependent chain run in ONE launch and the kernel boundary is replaced by a same-XCD hand-off (SeqLink).
#ifndef ACEZ_SEQ_SLEEP
#define ACEZ_SEQ_SLEEP "1"   // s_sleep argument between two polls of a hand-off counter (64 clocks each)
#endif
struct SeqLink {
  uint32_t* flag;        // per-row-tile counter in this XCD's L2 (one 128-byte line each), monotonically increasing
  uint32_t target;       // value of *flag when the four column tiles of the layer before have stored their outputs
  bool first, wait;      // first layer of the launch (requests its own W stages) / In is produced inside this launch
  bool signal;           // another layer follows: bump *flag after the stores
  const uint16_t* next_W;  // weights of the layer after (null: none): its first four W stages are requested before the epilogue
  uint32_t flag_index;   // flag = flags + flag_index; the trainer's sticky fault word is flags[64 * 32]: set when the bounded poll
                         // below expires (the host then falls back to per-layer launches, head_api.hip seq_fault_check)
  uint32_t limit;        // the poll budget (a number of polls), a kernel argument
};
constexpr int RG80_STAGE = (128 + 96) * 64;                        // elements per ring slot
constexpr int RG80_SMEM = 4 * RG80_STAGE + 2 * 80 * 128;           // ring + two staging tiles
constexpr int RG80_SMEM_SEQ = RG80_SMEM + 1024;                    // + the bias-partial combine buffer (the ring is busy in SEQ mode)

template <bool BIAS_RELU, bool HAS_ADD, bool HAS_MASK, int AUX, bool SEQ, class E = EltBf16>
__device__ __forceinline__ void rowgemm80_body(const RowGemmArgs& a, uint16_t* smem, const int active, const int mt, const int n0,
                                               const SeqLink& q) {
  constexpr int STAGE = RG80_STAGE;
  uint16_t* const stA = smem + 4 * STAGE;   // `add` in / aux out
  uint16_t* const stB = stA + 80 * 128;     // mask | res in / main out
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m0 = mt * 80;
  const int M = a.M, N = a.N;
  constexpr int K = 512, KT = 8;
  constexpr bool HAS_IN2 = AUX == AUX_RESIDUAL;   // (the ReLU mask of an input-gradient layer is a lane-private bit word since round 6, not a tile)
  constexpr int EPI = 5 * ((HAS_ADD ? 1 : 0) + (HAS_IN2 ? 1 : 0));  // epilogue-input DMA instructions per loader

  if (w >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - 4;
    if (ACEZ_DBG(a.dbg) & 4) {
      for (int kt = 0; kt < KT + 1; ++kt) __builtin_amdgcn_s_barrier();
    } else {
      // W instructions 4lw .. 4lw+3 (8 rows each), In instructions 3lw .. 3lw+2; lane: row + (l>>3), slot l&7
      const uint16_t* gW[4];
      const uint16_t* gI[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (lw * 4 + j) * 8 + (l >> 3);
        gW[j] = a.W + (size_t)(n0 + row) * K + ((l & 7) ^ ((row >> 1) & 7)) * 8;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int row = (lw * 3 + j) * 8 + (l >> 3);
        gI[j] = a.In + (size_t)min(m0 + row, M - 1) * K + ((l & 7) ^ ((row >> 1) & 7)) * 8;
      }
      // (Round 6: K stages started at stage (row tile & 7), so that the eight row tiles an XCD holds of one column tile pull eight different
      // weight stages at any moment -- the rotation head_maps.hip gains 15 % from -- measured +3.5 us on the forward chain and +1.5 us on the
      // input-gradient chain (profiles/r06_krot_experiments.log): in lockstep one L2 fill serves every workgroup that shares the stage.)
      auto issueW = [&](int kt) {
        uint16_t* slot = smem + (kt & 3) * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_global_load_lds((gvoid_t*)(gW[j] + kt * 64), (lvoid_t*)(slot + (lw * 4 + j) * 8 * 64), 16, 0, 0);
      };
      auto issueI = [&](int kt) {
        uint16_t* slot = smem + (kt & 3) * STAGE;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          __builtin_amdgcn_global_load_lds((gvoid_t*)(gI[j] + kt * 64), (lvoid_t*)(slot + 128 * 64 + (lw * 3 + j) * 8 * 64), 16, 0, 0);
      };
      auto issue = [&](int kt) { issueW(kt); issueI(kt); };
      // epilogue inputs: 4-row groups 5lw .. 5lw+4 of the [80][128] tile; lane: row + (l>>4), physical chunk l&15 receives
      // the logical chunk (l&15) ^ (row & 15)
      auto issue_tile = [&](const uint16_t* src, uint16_t* dst) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int row = (lw * 5 + j) * 4 + (l >> 4);
          const uint16_t* g = src + (size_t)min(m0 + row, M - 1) * N + n0 + (((l & 15) ^ (row & 15)) << 3);
          __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(dst + (lw * 5 + j) * 4 * 128), 16, 0, 0);
        }
      };
      if (!SEQ) {
        issue(0); issue(1); issue(2); issue(3);
      } else {
        // W stages 0..3 first (requested by the layer before unless this is the first), then -- once the four column tiles
        // of the producing layer have landed in this XCD's L2 -- the In stages
        if (q.first) { issueW(0); issueW(1); issueW(2); issueW(3); }
        if (q.wait) {
          // Bounded: a sibling that never arrives (workgroups of one row tile on different XCDs -- every L2 then holds its own copy
          // of the counter and none of them ever reaches the target --, or a sibling that is never dispatched) must not hang the
          // stream. When the budget expires the wave raises the sticky fault word and goes on with whatever is in memory: the
          // results of this launch are garbage, the optimiser and schedule kernels of the step see the word and do nothing, and
          // the host switches the trainer to per-layer launches at its next state read (head_api.hip).
          // The whole poll is ONE asm statement: written as a C loop with a second exit, the compiler re-scheduled the K loops and
          // epilogues of the input-gradient instantiation (+3 us per chain, same instruction mix; found by diffing the ISA of a
          // build without the bound). sc1: past this CU's L1; the counter and the tiles live in the L2 all four workgroups share,
          // no L2 invalidate. Every lane loads the same word: the loop condition is scalar. The budget is counted in polls
          // (>= ~0.5 us each: an L2 round trip + the sleep).
          uint32_t vseen, sseen, spins, timed;
          asm volatile(
              "s_mov_b32 %[spins], 0\n\t"
              "s_mov_b32 %[timed], 0\n"
              "1:\n\t"
              "global_load_dword %[vseen], %[flag], off sc1\n\t"
              "s_waitcnt vmcnt(0)\n\t"
              "v_readfirstlane_b32 %[sseen], %[vseen]\n\t"
              "s_sub_i32 %[sseen], %[sseen], %[target]\n\t"
              "s_cmp_ge_i32 %[sseen], 0\n\t"
              "s_cbranch_scc1 2f\n\t"
              "s_sleep " ACEZ_SEQ_SLEEP "\n\t"
              "s_add_u32 %[spins], %[spins], 1\n\t"
              "s_cmp_lt_u32 %[spins], %[limit]\n\t"
              "s_cbranch_scc1 1b\n\t"
              "s_mov_b32 %[timed], 1\n"
              "2:"
              : [vseen] "=&v"(vseen), [sseen] "=&s"(sseen), [spins] "=&s"(spins), [timed] "=&s"(timed)
              : [flag] "v"(q.flag), [target] "s"(q.target), [limit] "s"(q.limit)
              : "memory", "scc");
#ifndef ACEZ_SEQ_UNBOUNDED   // (timing experiments only, tools/lib_variant.sh: what the fault path costs)
          if (timed && l == 0) {   // the fault word, and the step is switched off: every later kernel of this trainer starts with `if (!st->active) return`
            __hip_atomic_store(q.flag - q.flag_index + 64 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.st) __hip_atomic_store(const_cast<int*>(&a.st->active), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#endif
        }
        issueI(0); issueI(1); issueI(2); issueI(3);
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        // in-order completion: everything younger than stage kt may still be in flight
        if (SEQ && kt == 0) ACEZ_VMCNT(9);
        else if (SEQ && kt == 1) ACEZ_VMCNT(6);
        else if (SEQ && kt == 2) ACEZ_VMCNT(10);
        else if (kt == 0) ACEZ_VMCNT(21);
        else if (kt <= 4) ACEZ_VMCNT(14);
        else if (kt == 5) ACEZ_VMCNT_C(14 + EPI);
        else if (kt == 6) ACEZ_VMCNT_C(7 + EPI);
        else ACEZ_VMCNT_C(EPI);
        __builtin_amdgcn_s_barrier();   // stage kt has landed; the multipliers are done with stage kt - 1
        if (kt >= 1 && kt + 3 < KT) issue(kt + 3);
        if (kt == 4) {
          if (HAS_ADD) issue_tile(a.add, stA);
          if (HAS_IN2) issue_tile(a.res, stB);
        }
      }
      ACEZ_VMCNT(0);
      __builtin_amdgcn_s_barrier();     // epilogue inputs have landed; the ring is free
      if (SEQ && q.next_W) {            // overlaps the epilogue: 64 KiB of the next layer's 208 KiB (requested behind the hand-off
#pragma unroll                          // instead, so that the loader waves' vmcnt(0) there covers stores only: measured +3 us per chain)
        for (int j = 0; j < 4; ++j) gW[j] += q.next_W - a.W;
        issueW(0); issueW(1); issueW(2); issueW(3);
      }
    }
    if (!SEQ && ((ACEZ_DBG(a.dbg) & 1) || !active)) return;
    __builtin_amdgcn_s_barrier();       // the multipliers have written the output tiles
  } else {
    // ------------------------------------------------------------------ multiplier waves
    f32x4 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const int fr = l & 15, fq = l >> 4;
    // this lane's 40 ReLU mask bits (RowGemmArgs::mask_in): requested now, used after the K loop. Inline asm: a plain load would be sunk
    // to its first use by the scheduler, i.e. into the epilogue, where its round trip would be exposed
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 mb = {0u, 0u};
    if (HAS_MASK) {
      const uint2* mp = a.mask_in + ((size_t)(mt * 4 + (n0 >> 7)) * 4 + w) * 64 + l;
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(mb) : "v"(mp) : "memory");
    }
    float4 bias[2];
    if (BIAS_RELU) {
#pragma unroll
      for (int i = 0; i < 2; ++i) bias[i] = *reinterpret_cast<const float4*>(a.bias + n0 + w * 32 + i * 16 + 4 * fq);
    }
    // The K loop is straight-line code (the ablation bits are compile-time 0 in the product build): without this the scheduler sinks
    // the second half's MFMAs below the next s_barrier -- their ds_reads would still be in flight when the loader waves, released by
    // that barrier, refill the slot (only the DMA latency protects them: a rare last-bit corruption under two processes on one GPU,
    // found with tools/seq_stress.py). Keep the fragment reads of a stage complete before the wave arrives at the barrier.
    auto stage_barrier = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      stage_barrier();
      if (ACEZ_DBG(a.dbg) & 2) continue;
      const uint16_t* sW = smem + (kt & 3) * STAGE;
      const uint16_t* sI = sW + 128 * 64;
      if constexpr (!HAS_MASK) {
        // forward instantiations: the scheduler's own order (reads of the second K step behind the first MFMAs) is the better one
        // (A/B, round 5: the explicit order below costs the forward chain 0.5-1.5 us)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int c = kk * 4 + fq;
          typename E::frag fa[2], fb[5];
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const typename E::frag*>(&sW[swz(w * 32 + i * 16 + fr, c)]);
#pragma unroll
          for (int j = 0; j < 5; ++j) fb[j] = *reinterpret_cast<const typename E::frag*>(&sI[swz(j * 16 + fr, c)]);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = E::mfma16(fa[i], fb[j], acc[i][j]);
        }
      } else {
      // Input-gradient instantiations: all fourteen fragment reads of the stage are requested before its first MFMA and every MFMA
      // waits for exactly the reads it needs. A multiplier wave is ALONE on its SIMD (its neighbour only issues DMA), so every LDS
      // round trip left between two MFMAs is matrix-pipe idle time: left to the scheduler, these instantiations read a fragment, waited
      // for it -- lgkmcnt(0) -- and multiplied, two to three times per stage (60 such read-wait pairs in rowseq_kernel<true>'s K loops
      // against 3 in the forward chain's; ISA scan, round 5: -2.0 ... -2.6 us on the input-gradient chain). Same MFMAs, same order.
      typename E::frag fa[2][2], fb[2][5];
      // The counted waits below assume ONE ds_read_b128 per fragment (fourteen reads in flight): a fragment is 16 bytes at a 16-byte aligned
      // LDS address (swz() returns multiples of eight elements). A wider / split fragment would turn the counts into wrong hints.
      static_assert(sizeof(typename E::frag) == 16 && alignof(typename E::frag) == 16, "one ds_read_b128 per fragment: ACEZ_KSTEP's lgkmcnt ladder");
      // request order = use order: fa[kk][0], fb[kk][0..4], fa[kk][1] for kk = 0, 1 (LDS returns in order)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int c = kk * 4 + fq;
        fa[kk][0] = *reinterpret_cast<const typename E::frag*>(&sW[swz(w * 32 + fr, c)]);
#pragma unroll
        for (int j = 0; j < 5; ++j) fb[kk][j] = *reinterpret_cast<const typename E::frag*>(&sI[swz(j * 16 + fr, c)]);
        fa[kk][1] = *reinterpret_cast<const typename E::frag*>(&sW[swz(w * 32 + 16 + fr, c)]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // each MFMA waits for exactly the reads it needs: 14 in flight, the first needs two of them, ... (counts = reads that may remain)
#define ACEZ_KSTEP(KK, BASE)                                                                                   \
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((BASE) + 5) : "memory"); acc[0][0] = E::mfma16(fa[KK][0], fb[KK][0], acc[0][0]); \
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((BASE) + 4) : "memory"); acc[0][1] = E::mfma16(fa[KK][0], fb[KK][1], acc[0][1]); \
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((BASE) + 3) : "memory"); acc[0][2] = E::mfma16(fa[KK][0], fb[KK][2], acc[0][2]); \
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((BASE) + 2) : "memory"); acc[0][3] = E::mfma16(fa[KK][0], fb[KK][3], acc[0][3]); \
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((BASE) + 1) : "memory"); acc[0][4] = E::mfma16(fa[KK][0], fb[KK][4], acc[0][4]); \
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(BASE) : "memory");                                           \
      acc[1][0] = E::mfma16(fa[KK][1], fb[KK][0], acc[1][0]); acc[1][1] = E::mfma16(fa[KK][1], fb[KK][1], acc[1][1]); \
      acc[1][2] = E::mfma16(fa[KK][1], fb[KK][2], acc[1][2]); acc[1][3] = E::mfma16(fa[KK][1], fb[KK][3], acc[1][3]); \
      acc[1][4] = E::mfma16(fa[KK][1], fb[KK][4], acc[1][4]);
      ACEZ_KSTEP(0, 7)
      __builtin_amdgcn_sched_barrier(0);
      ACEZ_KSTEP(1, 0)
#undef ACEZ_KSTEP
      }
    }
    stage_barrier();                    // epilogue inputs have landed
    if (!SEQ && ((ACEZ_DBG(a.dbg) & 1) || !active)) return;
    float amax = 0.f;                   // fp16 gradient layers: largest |value| before the conversion (absmax_publish)
    // The epilogue inputs this lane needs from the staging tiles (`add` values in stA, mask / residual values in stB: up to 2 x 10 eight-byte
    // reads), ALL requested before the first is used: written as read-compute-write per fragment the compiler could not move a read above
    // the previous fragment's write to the same tile (it cannot see that the addresses differ) and followed every read by a full
    // lgkmcnt(0) -- ten to twenty serial LDS round trips per layer in the input-gradient chain (85 read-wait pairs in rowseq_kernel<true>'s
    // ISA against 2 in the forward chain's; found with a scan for load-wait pairs, round 5). Same values, same arithmetic.
    uint2 ra[2][5], rb[2][5];
    if (HAS_ADD || HAS_IN2) {
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int o = st_off(j * 16 + fr, w * 32 + i * 16 + 4 * fq);
          if (HAS_ADD) ra[i][j] = *reinterpret_cast<const uint2*>(&stA[o]);
          if (HAS_IN2) rb[i][j] = *reinterpret_cast<const uint2*>(&stB[o]);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (HAS_MASK) asm volatile("s_waitcnt vmcnt(0)" : "+v"(mb)::"memory");   // the mask bits requested before the K loop (long since landed)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int ml = j * 16 + fr;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int nl = w * 32 + i * 16 + 4 * fq;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        uint16_t* pa = &stA[st_off(ml, nl)];
        uint16_t* pb = &stB[st_off(ml, nl)];
        if (BIAS_RELU) {
          v[0] += bias[i].x; v[1] += bias[i].y; v[2] += bias[i].z; v[3] += bias[i].w;
        }
        if (HAS_ADD) {
          float ad[4];
          E::un4(ra[i][j], ad);
          v[0] += ad[0]; v[1] += ad[1]; v[2] += ad[2]; v[3] += ad[3];
        }
        if (BIAS_RELU) {
          v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        if (E::is_f16 && HAS_MASK) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        uint2 y = E::pk4(v[0], v[1], v[2], v[3]);
        if (AUX == AUX_RESIDUAL) {
          float yf[4], rf[4];
          E::un4(y, yf);
          E::un4(rb[i][j], rf);
          *reinterpret_cast<uint2*>(pa) = E::pk4(yf[0] + rf[0], yf[1] + rf[1], yf[2] + rf[2], yf[3] + rf[3]);
        } else if (AUX == AUX_UNMASKED) {
          *reinterpret_cast<uint2*>(pa) = y;
        }
        const int f = 2 * j + i;   // (compile-time: both loops are unrolled)
        if (HAS_MASK) {
          // keep a half word iff its bit is set: sign-extended one-bit fields -> 0 / ~0, merged per half
          const uint32_t kx = ((uint32_t)__builtin_amdgcn_sbfe((int)mb.x, 9 - f, 1) & 0x0000ffffu) | ((uint32_t)__builtin_amdgcn_sbfe((int)mb.x, 25 - f, 1) & 0xffff0000u);
          const uint32_t ky = ((uint32_t)__builtin_amdgcn_sbfe((int)mb.y, 9 - f, 1) & 0x0000ffffu) | ((uint32_t)__builtin_amdgcn_sbfe((int)mb.y, 25 - f, 1) & 0xffff0000u);
          y.x &= kx; y.y &= ky;
        }
        *reinterpret_cast<uint2*>(pb) = y;
      }
    }
    if (E::is_f16 && HAS_MASK && a.absmax) absmax_publish(a.absmax, amax);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // output tiles complete
  }
  // ------------------------------------------------------------------ all eight waves: copy the tiles out
  // All LDS reads first, into distinct registers, then the stores: as a read-store loop the compiler reused one register quadruple
  // and every read waited -- vmcnt(0) -- for the previous store to COMPLETE (its data register must not be overwritten while the
  // store is in flight): 2-5 serial store round trips at the end of each of the step's 15 launches.
  {
    const int q0 = t, q1 = t + 512, q2 = t + 1024;
    const int so0 = (q0 >> 4) * 128 + (((q0 & 15) ^ ((q0 >> 4) & 15)) << 3);
    const int so1 = (q1 >> 4) * 128 + (((q1 & 15) ^ ((q1 >> 4) & 15)) << 3);
    const int so2 = (q2 < 1280) ? (q2 >> 4) * 128 + (((q2 & 15) ^ ((q2 >> 4) & 15)) << 3) : 0;
    const uint4 m0v = *reinterpret_cast<const uint4*>(&stB[so0]);
    const uint4 m1v = *reinterpret_cast<const uint4*>(&stB[so1]);
    const uint4 m2v = *reinterpret_cast<const uint4*>(&stB[so2]);
    uint4 a0v = m0v, a1v = m1v, a2v = m2v;
    if (AUX != AUX_NONE) {
      a0v = *reinterpret_cast<const uint4*>(&stA[so0]);
      a1v = *reinterpret_cast<const uint4*>(&stA[so1]);
      a2v = *reinterpret_cast<const uint4*>(&stA[so2]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (also keeps the reads above this point)
    const int r0 = m0 + (q0 >> 4), r1 = m0 + (q1 >> 4), r2 = m0 + (q2 >> 4);
    const size_t o0 = (size_t)r0 * N + n0 + (q0 & 15) * 8, o1 = (size_t)r1 * N + n0 + (q1 & 15) * 8, o2 = (size_t)r2 * N + n0 + (q2 & 15) * 8;
    // SEQ: the tile the NEXT layer reads goes first -- the main tile, or after a residual add the aux tile (the residual stream is the next
    // block's input) -- and the hand-off follows its acknowledgement; the other tile (pre-residual activation kept for the backward pass /
    // unmasked gradient with its tile-local consumer) and the bias column sums come after the signal
    constexpr bool AUX_FIRST = AUX == AUX_RESIDUAL;
    uint16_t* const first = AUX_FIRST ? a.out_aux : a.out_main;
    uint16_t* const second = AUX_FIRST ? a.out_main : a.out_aux;
    const uint4 f0 = AUX_FIRST ? a0v : m0v, f1v = AUX_FIRST ? a1v : m1v, f2v = AUX_FIRST ? a2v : m2v;
    const uint4 s0 = AUX_FIRST ? m0v : a0v, s1 = AUX_FIRST ? m1v : a1v, s2 = AUX_FIRST ? m2v : a2v;
    if (r0 < M) *reinterpret_cast<uint4*>(first + o0) = f0;
    if (r1 < M) *reinterpret_cast<uint4*>(first + o1) = f1v;
    if (q2 < 1280 && r2 < M) *reinterpret_cast<uint4*>(first + o2) = f2v;
    if (SEQ && q.signal) {
      ACEZ_VMCNT(0);   // this wave's stores are acknowledged by the L2 = visible to the three sibling workgroups (same XCD)
      if (l == 0) {
        const uint32_t one = 1;
        asm volatile("global_atomic_add %0, %1, off" ::"v"(q.flag), "v"(one) : "memory");
      }
    }
    if (AUX != AUX_NONE && second) {   // (null: a training forward does not keep the pre-residual activation -- only its mask bits are read again)
      if (r0 < M) *reinterpret_cast<uint4*>(second + o0) = s0;
      if (r1 < M) *reinterpret_cast<uint4*>(second + o1) = s1;
      if (q2 < 1280 && r2 < M) *reinterpret_cast<uint4*>(second + o2) = s2;
    }
  }
  if (BIAS_RELU && a.mask_out && w < 4) {
    // ReLU mask bits of this launch's output tile, for the input-gradient layer that will need them (RowGemmArgs::mask_out): computed BEHIND
    // the hand-off, from the main-output staging tile, in the time the multiplier waves would otherwise spend waiting for the next layer's
    // first K stage (in the epilogue proper -- 60 instructions per lane in front of the signal -- they cost the forward chain 2 us per
    // step; a store in front of the signal another 1.6 us per LAYER: its acknowledgement was waited for by the signalling vmcnt(0)).
    // The staging tile is intact until the next layer's loaders refill it behind that layer's K stage 4.
    // A 16-bit output is > 0 iff it is non-zero (the ReLU leaves no sign bit: v_max_f32(-0, +0) = +0): the per-half unsigned minimum with
    // 1 is the flag (v_pk_min_u16; inline asm: written as a vector minimum the compiler scalarised it into a dozen 16-bit compares and
    // selects per word); fragment f = 2 j + i ends at bits 9 - f and 25 - f of its word.
    const int fr = l & 15, fq = l >> 4;
    uint2 yv[5][2];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) yv[j][i] = *reinterpret_cast<const uint2*>(&stB[st_off(j * 16 + fr, w * 32 + i * 16 + 4 * fq)]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint2 mbits = make_uint2(0u, 0u);
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint32_t fx, fy;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(fx) : "v"(yv[j][i].x), "s"(0x00010001u));
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(fy) : "v"(yv[j][i].y), "s"(0x00010001u));
        mbits.x = (mbits.x << 1) | fx;
        mbits.y = (mbits.y << 1) | fy;
      }
    a.mask_out[((size_t)(mt * 4 + (n0 >> 7)) * 4 + w) * 64 + l] = mbits;   // 512 contiguous bytes per wave
  }
  if (HAS_MASK && a.bias_partials) {
    // bias gradient partial of this 80-row tile = column sums of the bf16 output tile: four 20-row groups in parallel,
    // combined in a fixed order (the ring is free by now)
    const int col = t & 127, g = t >> 7;
    const int rows = min(80, M - m0);
    // (all twenty reads first, then the adds in row order: as a read-add loop every read was followed by a full lgkmcnt(0) -- twenty serial
    // LDS round trips per layer, found with the same scan as the epilogue reads above)
    uint16_t colv[20];
#pragma unroll
    for (int r = 0; r < 20; ++r) colv[r] = stB[st_off(g * 20 + r, col)];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float sacc = 0.f;
#pragma unroll
    for (int r = 0; r < 20; ++r) {
      const int row = g * 20 + r;
      const float v = E::to_f(colv[r]);
      sacc += (row < rows) ? v : 0.f;
    }
    float* red = reinterpret_cast<float*>(SEQ ? smem + RG80_SMEM : smem);
    red[g * 128 + col] = sacc;
    if (SEQ) {   // not __syncthreads(): its fence would also wait for the next layer's W stages
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
    if (t < 128) a.bias_partials[(size_t)mt * 512 + n0 + t] = ((red[t] + red[128 + t]) + red[256 + t]) + red[384 + t];
  }
}

template <bool BIAS_RELU, bool HAS_ADD, bool HAS_MASK, int AUX, class E = EltBf16>
__global__ __launch_bounds__(512) void rowgemm80_kernel(RowGemmArgs a) {
  const int active = a.st ? a.st->active : 1;
  __shared__ __attribute__((aligned(16))) uint16_t smem[RG80_SMEM];
  const int mtiles = (a.M + 79) / 80;
  const int per_xcd = (mtiles + 7) >> 3;
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + (jx >> 2);
  if (mt >= mtiles) return;
  rowgemm80_body<BIAS_RELU, HAS_ADD, HAS_MASK, AUX, false, E>(a, smem, active, mt, (jx & 3) * 128, SeqLink{});
}

// ---------------------------------------------------------------------------------------------------
// rowseq: the layers of a dependent GEMM chain (the 8 forward layers, or the 7 input-gradient layers) in ONE launch with
// rowgemm80's tiling, roles, ring, K order and epilogues -- bit-identical outputs. Layer l + 1 of row tile mt reads layer l's
// 80 x 512 output = the four column tiles of mt, which the grid decode places on ONE XCD, so the hand-off stays inside that
// XCD's L2: every wave bumps the row tile's counter once its stores are acknowledged, the next layer's loader waves poll it
// past their L1. No agent-scope fence (a buffer_inv sc1 / buffer_wbl2 sc1 pair per seam costs 5 - 40 us: measured with
// tools/seam_proto.hip). Tile-local epilogue inputs (`add`, `res`) were written by the same CU or before the launch; `In` is
// never read before it is produced, so no stale L1 line can exist. What the seam saves over a kernel boundary: the launch
// ramp, and the next layer's first four W stages are in flight while the epilogue runs (tools/seam_proto.hip: 6.13 -> 4.88 us
// per forward layer at 5120 rows). All workgroups must be resident (they wait for their siblings): the host only uses this
// kernel when the grid is not larger than the number of CUs.
// ---------------------------------------------------------------------------------------------------
struct SeqLayer {
  const uint16_t *In, *W;
  const float* bias;
  const uint16_t *add, *res;
  uint2* mask_out;          // RowGemmArgs::mask_out / mask_in
  const uint2* mask_in;
  uint16_t *out_main, *out_aux;
  float* bias_partials;
  int aux_mode, pad;
};
constexpr int SEQ_MAX_LAYERS = 8;
struct RowSeqArgs {
  SeqLayer layer[SEQ_MAX_LAYERS];
  int n_layers, M;
  const TrainState* st;
  uint32_t* flags;     // [64 row tiles][32] hand-off counters + [64 * 32] the fault word
  uint32_t base[64];   // per row tile: seams completed by earlier launches (a launch only touches the row tiles of ITS batch)
  uint32_t spin_limit; // poll budget of a hand-off (polls of >= ~0.5 us); expired: fault word + st->active = 0
  uint32_t* xcc_dbg;   // null, or [8 + 256]: words 0..7 |= 1 << XCC_ID of the workgroups with blockIdx & 7 = word (sticky); word
                       // 8 + 4 mt + nt = XCC_ID of the workgroup that owned tile (mt, nt) in the last launch (tests: the four column
                       // tiles of a row tile must report the same XCD)
                       // flags[64 * 32] = the sticky fault word (non-zero: a poll of this trainer has expired, every launch returns at
                       // once), flags[64 * 32 + 1] = a copy of the poll budget (diagnostics)
};

template <bool BWD, class E = EltBf16>
__global__ __launch_bounds__(512) void rowseq_kernel(RowSeqArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[RG80_SMEM_SEQ];
  const int mtiles = (a.M + 79) / 80;
  const int per_xcd = (mtiles + 7) >> 3;
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + (jx >> 2);
  if (mt >= mtiles) return;
  if (a.xcc_dbg && threadIdx.x == 0) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
    atomicOr(a.xcc_dbg + (blockIdx.x & 7), 1u << xcc);
    a.xcc_dbg[8 + mt * 4 + (jx & 3)] = xcc;
  }
  // (after an expired poll the fault handler has cleared `active`: nothing of this trainer runs until the host has fallen back)
  // (... or a hand-off poll of this trainer has expired in an EARLIER launch -- the sticky fault word behind the counters: the buffers of
  // the faulted step are left alone until the host has fallen back; both words are wave-uniform loads issued together)
  if ((a.st && !a.st->active) || a.flags[64 * 32]) {   // training has ended on the device: no work, but the counters keep step with the host's bases
    if (threadIdx.x == 0) {      // (the same L2-local atomic as the hand-off itself)
      const uint32_t inc = 8u * (uint32_t)(a.n_layers - 1);
      asm volatile("global_atomic_add %0, %1, off" ::"v"(a.flags + mt * 32), "v"(inc) : "memory");
    }
    return;
  }
  const int n0 = (jx & 3) * 128;
  for (int layer = 0; layer < a.n_layers; ++layer) {
    const SeqLayer& y = a.layer[layer];
    RowGemmArgs g;
    g.In = y.In; g.W = y.W; g.bias = y.bias; g.add = y.add; g.mask_out = y.mask_out; g.mask_in = y.mask_in; g.res = y.res; g.out_main = y.out_main; g.out_aux = y.out_aux;
    g.bias_partials = y.bias_partials; g.M = a.M; g.N = 512; g.K = 512; g.relu = BWD ? 0 : 1; g.aux_mode = y.aux_mode; g.st = a.st; g.dbg = 0;
    g.absmax = (BWD && a.st) ? const_cast<uint32_t*>(a.st->dz_absmax_slots) : nullptr;
    SeqLink q;
    q.flag = a.flags + mt * 32; q.target = (a.base[mt] + (uint32_t)layer) * 32u;   // 4 workgroups x 8 waves per seam
    q.first = layer == 0; q.wait = layer > 0; q.signal = layer + 1 < a.n_layers;
    q.next_W = q.signal ? a.layer[layer + 1].W : nullptr;
    q.flag_index = (uint32_t)(mt * 32); q.limit = a.spin_limit;
    if (!BWD) {
      if (y.aux_mode == AUX_RESIDUAL) rowgemm80_body<true, false, false, AUX_RESIDUAL, true, E>(g, smem, 1, mt, n0, q);
      else rowgemm80_body<true, false, false, AUX_NONE, true, E>(g, smem, 1, mt, n0, q);
    } else {
      if (y.add) rowgemm80_body<false, true, true, AUX_UNMASKED, true, E>(g, smem, 1, mt, n0, q);
      else if (y.aux_mode == AUX_UNMASKED) rowgemm80_body<false, false, true, AUX_UNMASKED, true, E>(g, smem, 1, mt, n0, q);
      else rowgemm80_body<false, false, true, AUX_NONE, true, E>(g, smem, 1, mt, n0, q);
    }
  }
}

// Placement probe (acez_trainer_create): a launch with rowseq_kernel's grid, block size and LDS footprint that only records
// which XCD ran tile (mt, nt) under rowseq_kernel's decode. The host enables the one-launch chains only if the four column tiles
// of every row tile report the same XCD. (A probe cannot promise anything about later launches -- HIP makes no placement
// contract -- which is why the hand-off itself is bounded and fault-checked; the probe keeps a part or a partition mode on which
// the mapping does not hold from ever taking the fault path.)
__global__ __launch_bounds__(512) void seq_probe_kernel(uint32_t* rec /*[256]*/, int mtiles) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[RG80_SMEM_SEQ];
  const int per_xcd = (mtiles + 7) >> 3;
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + (jx >> 2);
  if (threadIdx.x == 0) smem[0] = (uint16_t)blockIdx.x;   // keeps the allocation
  __syncthreads();
  if (mt >= mtiles || threadIdx.x != 0) return;
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
  rec[mt * 4 + (jx & 3)] = (xcc << 16) | smem[0];
}

// host-side dispatch on the epilogue shape (the flags of RowGemmArgs select the instantiation)
static inline void launch_rowgemm(const RowGemmArgs& g, hipStream_t s, bool f16 = false) {
  const bool br = g.bias != nullptr;
  const int mtiles = (g.M + 79) / 80;
  const dim3 grid(8 * 4 * ((mtiles + 7) / 8));  // N = 512 -> 4 column tiles; XCD-aware decode inside the kernels
#define ACEZ_RG(...)                                                                                               \
  do {                                                                                                             \
    if (f16) hipLaunchKernelGGL((rowgemm80_kernel<__VA_ARGS__, EltF16>), grid, dim3(512), 0, s, g);                \
    else hipLaunchKernelGGL((rowgemm80_kernel<__VA_ARGS__>), grid, dim3(512), 0, s, g);                            \
  } while (0)
  if (br && g.aux_mode == AUX_NONE) ACEZ_RG(true, false, false, AUX_NONE);
  else if (br && g.aux_mode == AUX_RESIDUAL) ACEZ_RG(true, false, false, AUX_RESIDUAL);
  else if (!br && g.mask_in && !g.add && g.aux_mode == AUX_NONE) ACEZ_RG(false, false, true, AUX_NONE);
  else if (!br && g.mask_in && !g.add && g.aux_mode == AUX_UNMASKED) ACEZ_RG(false, false, true, AUX_UNMASKED);
  else if (!br && g.mask_in && g.add && g.aux_mode == AUX_UNMASKED) ACEZ_RG(false, true, true, AUX_UNMASKED);
  else if (!br && g.mask_in && g.add && g.aux_mode == AUX_NONE) ACEZ_RG(false, true, true, AUX_NONE);
  else abort();  // no other epilogue shape exists in the head
#undef ACEZ_RG
}

// ---------------------------------------------------------------------------------------------------
// wgrad: dW_l[n][c] = sum_m dZ_l[m][n] * In_l[m][c] (the bias gradients are column sums taken where dZ is produced), all wide layers in one
// launch (grid.y = layer), split-K over rows (slab s = blockIdx.x / 16). Both operands are row-major with
// the reduction index m as the slow dimension, so the MFMA fragments (8 consecutive m for one column) are
// read with ds_read_b64_tr_b16 from an untransposed [64 m][128 cols] LDS tile (row pitch 320 B: the four
// rows a 32-lane group touches fall on disjoint banks).
// ---------------------------------------------------------------------------------------------------
// LDS stage = [64 m][128 cols] bf16 per operand (256-byte rows, no padding: the image is written by LDS-DMA in
// lane order). The four rows a 32-lane tr-read group touches would share banks, so the 32-byte segment index of a
// row is XOR-ed with (row & 3) << 1 -- applied to the per-lane DMA source chunk and, identically, to the reads.
// element offset, inside a [64][128] stage, of the first of the two transposed 8-byte reads lane l issues for the
// fragment tile[k0 + 8*(l>>5) + e][col0 + (l & 31)], e = 0..7, for k0 = 0 (other k0: + k0 * 128)
__device__ __forceinline__ int tr_base(int col0, int l) {
  const int q = l >> 4, i16 = l & 15;
  const int row = 8 * (q >> 1) + (i16 >> 2);                     // row & 3 == i16 >> 2 (also for row + 4)
  const int colb = (col0 + 16 * (q & 1) + 4 * (i16 & 3)) * 2;    // logical byte offset inside the 256-byte row
  const int phys = ((((colb >> 5) ^ ((i16 >> 2) << 1)) << 5) | (colb & 31)) >> 1;
  return row * 128 + phys;
}
template <class E = EltBf16>
__device__ __forceinline__ typename E::frag tr_frag(const uint16_t* p) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 128));
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(typename E::frag, r);
}

#ifndef ACEZ_WGRAD_RING
#define ACEZ_WGRAD_RING 4
#endif
constexpr int WGRAD_LOADERS = 8;                    // loader waves per workgroup (beside the 4 multiplier waves)
constexpr int WGRAD_THREADS = 256 + 64 * WGRAD_LOADERS;
// s_waitcnt vmcnt(n) for a wave-uniform n that is only known at run time (1 .. 31; anything else waits for everything)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {
    case 1: ACEZ_VMCNT(1); break;
    case 2: ACEZ_VMCNT(2); break;
    case 3: ACEZ_VMCNT(3); break;
    case 4: ACEZ_VMCNT(4); break;
    case 5: ACEZ_VMCNT(5); break;
    case 6: ACEZ_VMCNT(6); break;
    case 7: ACEZ_VMCNT(7); break;
    case 8: ACEZ_VMCNT(8); break;
    case 9: ACEZ_VMCNT(9); break;
    case 10: ACEZ_VMCNT(10); break;
    case 11: ACEZ_VMCNT(11); break;
    case 12: ACEZ_VMCNT(12); break;
    case 13: ACEZ_VMCNT(13); break;
    case 14: ACEZ_VMCNT(14); break;
    case 15: ACEZ_VMCNT(15); break;
    case 16: ACEZ_VMCNT(16); break;
    case 17: ACEZ_VMCNT(17); break;
    case 18: ACEZ_VMCNT(18); break;
    case 19: ACEZ_VMCNT(19); break;
    case 20: ACEZ_VMCNT(20); break;
    case 21: ACEZ_VMCNT(21); break;
    case 22: ACEZ_VMCNT(22); break;
    case 23: ACEZ_VMCNT(23); break;
    case 24: ACEZ_VMCNT(24); break;
    case 25: ACEZ_VMCNT(25); break;
    case 26: ACEZ_VMCNT(26); break;
    case 27: ACEZ_VMCNT(27); break;
    case 28: ACEZ_VMCNT(28); break;
    case 29: ACEZ_VMCNT(29); break;
    case 30: ACEZ_VMCNT(30); break;
    case 31: ACEZ_VMCNT(31); break;
    default: ACEZ_VMCNT(0); break;
  }
}

// The K loop of one (layer, row slab, 128 x 128 tile) workgroup, shared by wgrad_kernel and wgrad_opt_kernel. Returns true in the
// loader waves (their loop is over: every stage has landed and every barrier of the loop has been passed), false in the four multiplier
// waves, whose accumulators then hold the slab's partial tile. KT = number of 64-row stages of this slab.
// PFN > 0: `prefetch()` issues exactly PFN vector-memory loads (wgrad_opt_kernel: the optimiser state of the workgroup's half tile) from
// inside the loader loop, about a dozen stages before its end, so that they travel beside the operand stream instead of after it; loads
// complete in order, so the counted waits of the stages requested BEFORE the prefetch allow PFN more instructions in flight.
template <class E, int PFN, class PF>
__device__ __forceinline__ bool wgrad_kloop(const WgradArgs& a, uint16_t (*smem)[2][64 * 128], const int layer, const int slab, const int tile,
                                            f32x16 (&acc)[2][2], int& KT_out, PF&& prefetch) {
  constexpr int RING = ACEZ_WGRAD_RING;   // slots of the [dZ | In] stage ring, 32 KiB each (RING - 1 stages in flight per CU)
  static_assert(2 * (16 / WGRAD_LOADERS) * (RING - 1) + PFN <= 31, "wait_vmcnt_dyn covers 1 .. 31");
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  // Role split: waves 0..3 multiply (2 x 2 grid of 64 x 64 sub-tiles: 4 fragment reads feed 4 MFMAs), waves 4..11 only
  // issue the LDS-DMA. A wave's instruction stream is in-order: with every wave loading AND multiplying, the ~0.4 us a
  // stage's DMA instructions need to get through the memory pipeline was added to the MFMA time instead of hidden
  // behind it (ablation: 18 us loads-only + 23 us MFMA-only = 41 us). Waves w and w + 4 share a SIMD, so every SIMD
  // has one multiplier and two loaders.
  const bool loader = w >= 4;
  const int cw = w & 3, wn = cw >> 1, wc = cw & 1;
  const int lw = w - 4;                         // loader index
  constexpr int GPL = 16 / WGRAD_LOADERS;       // 4-row DMA groups (x 2 operands) per loader and stage
  const int n0 = (tile >> 2) * 128, c0 = (tile & 3) * 128;
  const uint16_t* __restrict__ Z = a.dZ[layer];
  const uint16_t* __restrict__ X = a.In[layer];
  const int M = a.M;
  // rows of this slab: [mb, me), multiple of 64 per slab except the tail
  const int rows_per_slab = ((M + a.nslabs * 64 - 1) / (a.nslabs * 64)) * 64;
  const int mb = slab * rows_per_slab;
  const int me = min(M, mb + rows_per_slab);
  const int KT = (me > mb) ? (me - mb + 63) >> 6 : 0;
  KT_out = KT;

  // DMA instruction j of loader wave lw covers stage rows (lw*GPL+j)*4 .. +3 of both operands; this lane: row + (l>>4),
  // physical 16-byte chunk l&15, which must receive the logical chunk whose 32-byte segment index is XOR-swizzled
  const int prow = l >> 4, pq = l & 15;
  auto issue = [&](int kt) {
    const int slot = kt % RING;
#pragma unroll
    for (int j = 0; j < GPL; ++j) {
      const int srow = (lw * GPL + j) * 4 + prow;
      const int lchunk = ((((pq >> 1) ^ ((srow & 3) << 1)) << 1) | (pq & 1)) * 8;  // element offset of the source chunk
      const int m = mb + kt * 64 + srow;
      const bool ok = m < me;
      // rows past the slab end must contribute zeros: they are fetched from a zero page
      const uint16_t* gz = ok ? Z + (size_t)m * 512 + n0 + lchunk : a.zeros + pq * 8;
      const uint16_t* gx = ok ? X + (size_t)m * 512 + c0 + lchunk : a.zeros + pq * 8;
      __builtin_amdgcn_global_load_lds((gvoid_t*)gz, (lvoid_t*)&smem[slot][0][(lw * GPL + j) * 4 * 128], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gvoid_t*)gx, (lvoid_t*)&smem[slot][1][(lw * GPL + j) * 4 * 128], 16, 0, 0);
    }
  };

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Two separate loops (one per role) with the same number of barriers: sharing one loop body makes the compiler
  // carry the 64 accumulator registers through the loader's control flow (moves on every iteration).
  if (loader) {
    if (ACEZ_DBG(a.dbg) & 4) {
      for (int kt = 0; kt < KT; ++kt) __builtin_amdgcn_s_barrier();
      if (PFN > 0) prefetch();
      return true;
    }
    for (int kt = 0; kt < RING && kt < KT; ++kt) issue(kt);
    // stage kt: wait until it has landed, meet the multipliers (they are done with stage kt - 1), refill the slot of stage kt - 1
    auto stage = [&](int kt, int extra) {
      // stages issued so far: 0..RING-1 at kt = 0, 0..kt+RING-2 afterwards; a loader wave has 2 * GPL DMA instructions per stage in flight
      const int later = (kt == 0) ? min(RING - 1, KT - 1) : min(RING - 2, KT - 1 - kt);
      wait_vmcnt_dyn(2 * GPL * later + extra);
      __builtin_amdgcn_s_barrier();
      if (kt >= 1 && kt + RING - 1 < KT) issue(kt + RING - 1);
    };
    if (PFN == 0) {
      for (int kt = 0; kt < KT; ++kt) stage(kt, 0);
    } else {
      // the prefetch goes out behind the issue of iteration P (P < 0: behind the first RING stages); the last stage requested before
      // it is S: stages kt <= S are older than the prefetch, their waits allow it to be in flight
      const int P = KT > 16 ? KT - 13 : -1;
      for (int kt = 0; kt <= P; ++kt) stage(kt, 0);
      prefetch();
      const int S = P < 0 ? min(RING, KT) - 1 : min(P + RING - 1, KT - 1);
      for (int kt = P + 1; kt < KT; ++kt) stage(kt, kt <= S ? PFN : 0);
    }
    return true;
  }
  const int offA[2] = {tr_base(wn * 64, l), tr_base(wn * 64 + 32, l)};
  const int offB[2] = {tr_base(wc * 64, l), tr_base(wc * 64 + 32, l)};
  for (int kt = 0; kt < KT; ++kt) {
    __builtin_amdgcn_s_barrier();
    if (ACEZ_DBG(a.dbg) & 2) continue;
    const int slot = kt % RING;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      typename E::frag fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = tr_frag<E>(&smem[slot][0][offA[i] + kk * 16 * 128]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = tr_frag<E>(&smem[slot][1][offB[j] + kk * 16 * 128]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = E::mfma32(fa[i], fb[j], acc[i][j]);
    }
  }
  return false;
}

template <class E = EltBf16>
__global__ __launch_bounds__(WGRAD_THREADS) void wgrad_kernel(WgradArgs a) {
  const int active = a.st ? a.st->active : 1;  // tested before the stores only (see rowgemm80_body)
  __shared__ __attribute__((aligned(16))) uint16_t smem[ACEZ_WGRAD_RING][2][64 * 128];
  const int l = threadIdx.x & 63;
  const int cw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 3, wn = cw >> 1, wc = cw & 1;
  // XCD-aware decode: the 16 output tiles of one (layer, slab) group re-read the same dZ / In rows (4x each); they
  // are placed on ONE XCD (workgroup b runs on XCD b % 8) so that the re-reads hit that XCD's L2 instead of the
  // fabric. Placement only affects speed.
  const int b = blockIdx.x;
  const int xcd = b & 7, jx = b >> 3;
  const int group = xcd + 8 * (jx >> 4), tile = jx & 15;
  if (group >= a.n_layers * a.nslabs) return;
  const int layer = group / a.nslabs, slab = group - layer * a.nslabs;
  const int n0 = (tile >> 2) * 128, c0 = (tile & 3) * 128;
  f32x16 acc[2][2];
  int KT;
  if (wgrad_kloop<E, 0>(a, smem, layer, slab, tile, acc, KT, [] {})) return;

  if (ACEZ_DBG(a.dbg) & 1) { if (acc[0][0][0] == 1.2345e30f) a.slabs[0] = 1; return; }
  if (!active) return;
  float* __restrict__ G = a.slabs + (size_t)slab * a.slab_stride;
  const int h = l >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + wc * 64 + j * 32 + (l & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        G[a.w_off[layer] + (size_t)n * 512 + c] = acc[i][j][r];
      }
    }
}


// ---------------------------------------------------------------------------------------------------
// loss kernel (32 rows per workgroup, four wavefronts):
//   A  fc3 forward, wave wv: rows 8 wv .. 8 wv + 7, one row per pass, fp32 accumulate
//   B  one thread per row: de-homogenise, project, masks, robust loss and d(loss)/d(fc3 outputs)
//   C  wave wv: channels 128 wv .. +127 of all 32 rows, one thread per channel pair: fc3 weight-gradient partials and the
//      masked input gradient dZ
// ---------------------------------------------------------------------------------------------------
constexpr int LOSS_ROWS = 8;
__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// [32][512] bf16 activation tile of the row-persistent kernels (head_fused.hip, head_chain.hip): 16-byte chunk index XOR
// (row & 15), so that the 16 rows a ds_read_b128 lane group touches land on 16 different slots of a 256-byte bank row
__device__ __forceinline__ int act_off(int row, int ch) { return row * 512 + ((((ch >> 3) ^ (row & 15)) << 3) | (ch & 7)); }

// phase B's per-row inputs sit behind a chain of dependent loads (idx -> view -> image): they are fetched early (at kernel
// entry) so that the latency hides under phase A / under the forward GEMMs of the chain kernel
struct LossPre {
  int64_t p;
  int view, img;
  float tu, tv;
};
__device__ __forceinline__ LossPre loss_prefetch(const LossArgs& a, int m0, int t, int rows = LOSS_ROWS) {
  LossPre q{0, 0, 0, 0.f, 0.f};
  if (a.idx && t < rows && m0 + t < a.n) {
    q.p = a.idx[m0 + t];
    if (a.meta) {   // looked up where the batch was gathered (GatherMeta): one load level instead of three
      const int4 m = a.meta[m0 + t];
      q.view = m.x; q.img = m.y; q.tu = __int_as_float(m.z); q.tv = __int_as_float(m.w);
    } else {
      q.view = a.view_idx[q.p];
      q.img = a.view_image[q.view];
      q.tu = a.target_px[q.p * 2 + 0];
      q.tv = a.target_px[q.p * 2 + 1];
    }
  }
  return q;
}

// LDS scratch of the loss phases, in floats: s_s / s_ds / s_red [4 waves][LOSS_ROWS][4]
constexpr int LOSS_SCRATCH_FLOATS = 3 * 4 * LOSS_ROWS * 4;

// The three phases for the 32 rows of workgroup `block` (four wavefronts wv = 0..3 of 64 lanes t). LDSACT = false: the fc2
// output is read from a.act and dZ goes to a.dZ (loss_kernel). LDSACT = true: both live in the swizzled LDS tile `Xt`
// (act_off) of the chain kernel and dZ overwrites the activations in place (the caller copies the tile to a.dZ). The four waves
// meet once, through `sync` (loss_kernel: __syncthreads; the chain kernel: its flag barrier).
// PRE_INSIDE (loss_kernel): the per-row index chain of phase B (idx -> view -> image) is started here, BEHIND the loads of phase A:
// vmcnt completes in order, so a dependent chain issued first holds every later load behind its three round trips.
// LR rows per wavefront (4 LR per workgroup). The chain kernel's tile is 32 rows (LR = 8); loss_kernel runs LR = 4 by default: 320
// workgroups for a 5120-row batch instead of 160 on 256 CUs, half the serial row loop per wave (the summation order of the
// fc3 / bias partials follows the workgroup size, so the two sizes agree to rounding, not bitwise).
template <bool LDSACT, bool PRE_INSIDE, class E = EltBf16, int LR = 8, class Sync>
__device__ __forceinline__ void loss_body(const LossArgs& a, const int block, const int wv, const int t, uint16_t* Xt, float* scratch,
                                          LossPre pre, Sync sync) {
  constexpr int LOSS_ROWS = LR;
  float (*s_s)[4] = reinterpret_cast<float (*)[4]>(scratch + wv * LOSS_ROWS * 4);
  float (*s_ds)[4] = reinterpret_cast<float (*)[4]>(scratch + (4 + wv) * LOSS_ROWS * 4);
  float (*s_red)[4] = reinterpret_cast<float (*)[4]>(scratch + (8 + wv) * LOSS_ROWS * 4);
  const int l = t;
  const int m0 = (block * 4 + wv) * LOSS_ROWS;
  const int n = a.n, no = a.no;

  // ---- phase A. All of its loads are issued before anything is computed (as written row by row, every row's load was followed by
  // a full wait: twelve serial round trips).
  // (chain kernel, LDSACT: the rows are LDS reads issued where they are used -- batching them costs 60 registers the 768-thread
  // kernel does not have)
  uint4 w3raw[4], xraw[LDSACT ? 1 : LOSS_ROWS];
#pragma unroll
  for (int j = 0; j < 4; ++j) w3raw[j] = *reinterpret_cast<const uint4*>(a.W3 + (size_t)(j < no ? j : 0) * 512 + l * 8);
  if (!LDSACT) {
#pragma unroll
    for (int rr = 0; rr < LOSS_ROWS; ++rr) xraw[LDSACT ? 0 : rr] = *reinterpret_cast<const uint4*>(a.act + (size_t)min(m0 + rr, n - 1) * 512 + l * 8);
  }
  if (PRE_INSIDE) pre = loss_prefetch(a, m0, t, LOSS_ROWS);
  const int64_t pre_p = pre.p;
  const int pre_view = pre.view, pre_img = pre.img;
  const float pre_tu = pre.tu, pre_tv = pre.tv;
  // phase B's per-view / per-image matrices (augmentation, pose, intrinsics: 37 floats of a row's lane): with PRE_INSIDE they are
  // fetched here, as soon as view and image are known, and arrive while phase A computes
  float Am[12], Tm[16], Km[9];
  const bool early_mats = PRE_INSIDE && a.idx && t < LOSS_ROWS && m0 + t < n;
  if (early_mats) {
#pragma unroll
    for (int i = 0; i < 12; ++i) Am[i] = a.view_aug_inv[(size_t)pre_view * 12 + i];
#pragma unroll
    for (int i = 0; i < 16; ++i) Tm[i] = a.image_pose_inv[(size_t)pre_img * 16 + i];
#pragma unroll
    for (int i = 0; i < 9; ++i) Km[i] = a.view_K[(size_t)pre_view * 9 + i];
  }
  // phase C's loads too (training, loss_kernel): the rows come back from the L1 / L2 this workgroup has just pulled them through, but
  // requested behind the workgroup's meeting point they were one more round trip on the kernel's critical path
  constexpr bool EARLY_C = !LDSACT;
  const int chC = wv * 128 + 2 * t;
  uint32_t w3c[4] = {0u, 0u, 0u, 0u}, xv[LDSACT ? 1 : 4 * LOSS_ROWS];
  if (EARLY_C && a.idx) {
    const int mb = block * 4 * LOSS_ROWS;
#pragma unroll
    for (int j = 0; j < 4; ++j) w3c[j] = *reinterpret_cast<const uint32_t*>(a.W3 + (size_t)(j < no ? j : 0) * 512 + chC);
#pragma unroll
    for (int r = 0; r < 4 * LOSS_ROWS; ++r)   // rows past the end carry ds == 0 and are not stored
      xv[LDSACT ? 0 : r] = *reinterpret_cast<const uint32_t*>(a.act + (size_t)min(mb + r, n - 1) * 512 + chC);
  }
  {
    float w3[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      E::un4(make_uint2(w3raw[j].x, w3raw[j].y), &w3[j][0]);
      E::un4(make_uint2(w3raw[j].z, w3raw[j].w), &w3[j][4]);
      if (j >= no) {
#pragma unroll
        for (int e = 0; e < 8; ++e) w3[j][e] = 0.f;
      }
    }
    float b3v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b3v[j] = (j < no) ? a.b3[j] : 0.f;   // (not inside the row loop: a dependent load per row)
#pragma unroll
    for (int rr = 0; rr < LOSS_ROWS; ++rr) {
      const int r = rr, m = m0 + r;
      float x[8];
      const uint4 xr = LDSACT ? *reinterpret_cast<const uint4*>(&Xt[act_off(wv * LOSS_ROWS + r, l * 8)]) : xraw[LDSACT ? 0 : rr];
      E::un4(make_uint2(xr.x, xr.y), &x[0]);
      E::un4(make_uint2(xr.z, xr.w), &x[4]);
      if (!(m < n)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
      }
      float p[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float sacc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sacc = fmaf(x[e], w3[j][e], sacc);
        p[j] = sacc;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = wave_sum63(p[j]);   // DPP (the 24 ds_bpermute per row of a shuffle butterfly were most of phase A)
      if (l == 63) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s_s[r][j] = (j < no) ? p[j] + b3v[j] : 0.f;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (ACEZ_DBG(a.dbg) == 1) return;

  // ---- phase B
  const float gscale = (a.idx && a.st) ? a.st->grad_scale : 1.f;
  if (t < LOSS_ROWS) {
    const int r = t, m = m0 + r;
    float loss = 0.f, inl = 0.f, fgrad = 0.f, ds[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < n) {
      const float s0 = s_s[r][0], s1 = s_s[r][1], s2 = s_s[r][2], s3 = s_s[r][3];
      float X[3], hval = 1.f, bx = 0.f;
      bool hclamped = false;
      if (a.use_homogeneous) {
        bx = a.h_beta * s3;
        const float sp = (bx > 20.f) ? s3 : log1pf(expf(bx)) / a.h_beta;  // F.softplus(beta), ace_network.py:142
        const float hraw = sp + a.max_inv_scale;
        hclamped = hraw > a.min_inv_scale;                                // clamp_(max=min_inv_scale) :143
        hval = hclamped ? a.min_inv_scale : hraw;
        X[0] = s0 / hval + a.mean[0];
        X[1] = s1 / hval + a.mean[1];
        X[2] = s2 / hval + a.mean[2];
      } else {
        X[0] = s0 + a.mean[0];
        X[1] = s1 + a.mean[1];
        X[2] = s2 + a.mean[2];
      }
      if (a.out_xyz) {
        if (a.planar_hw > 0) {   // Regressor.forward's [B,3,H,W] layout (ace_network.py:265-270), what RANSAC reads
          const int64_t mg = (int64_t)a.row_offset + m;
          const int64_t fr = mg / a.planar_hw, px = mg - fr * a.planar_hw;
          float* o = a.out_xyz + fr * 3 * a.planar_hw + px;
          o[0] = X[0];
          o[(size_t)a.planar_hw] = X[1];
          o[(size_t)2 * a.planar_hw] = X[2];
        } else {
          a.out_xyz[(size_t)m * 3 + 0] = X[0];
          a.out_xyz[(size_t)m * 3 + 1] = X[1];
          a.out_xyz[(size_t)m * 3 + 2] = X[2];
        }
      }
      if (a.idx) {  // training: geometry + loss (skipped for pure inference)
        const float tu = pre_tu, tv = pre_tv;
        const int view = pre_view;
        if (!PRE_INSIDE) {
#pragma unroll
          for (int i = 0; i < 12; ++i) Am[i] = a.view_aug_inv[(size_t)view * 12 + i];
#pragma unroll
          for (int i = 0; i < 16; ++i) Tm[i] = a.image_pose_inv[(size_t)pre_img * 16 + i];
#pragma unroll
          for (int i = 0; i < 9; ++i) Km[i] = a.view_K[(size_t)view * 9 + i];
        }
        const float* A = Am;
        const float* T = Tm;
        float K[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) K[i] = Km[i];
        float kscale = 0.f;
        if (a.refine_calibration) {
          // refine_calibration.py:34-53: K[:2,:2] = (1+g) * f0 * (K00/f0) * I2
          kscale = K[0] / a.focal_init;
          const float f = (1.f + (float)a.st->calib_g) * a.focal_init * kscale;
          K[0] = f; K[1] = 0.f; K[3] = 0.f; K[4] = f;
        }
        // P = A(3x4) . T(4x4)    ace_trainer.py:530
        float P[12];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fmaf(A[i * 4 + k], T[k * 4 + j], acc);
            P[i * 4 + j] = acc;
          }
        float Xc[3], pp[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) Xc[i] = fmaf(P[i * 4 + 0], X[0], fmaf(P[i * 4 + 1], X[1], fmaf(P[i * 4 + 2], X[2], P[i * 4 + 3])));
#pragma unroll
        for (int i = 0; i < 3; ++i) pp[i] = fmaf(K[i * 3 + 0], Xc[0], fmaf(K[i * 3 + 1], Xc[1], K[i * 3 + 2] * Xc[2]));
        const bool zclamped = pp[2] < a.depth_min;                 // clamp_(min=depth_min) :545
        const float pz = zclamped ? a.depth_min : pp[2];
        const float u = pp[0] / pz, v = pp[1] / pz;
        const float du = u - tu, dv = v - tv;
        const float e = fabsf(du) + fabsf(dv);                     // L1 norm :552
        bool invalid = (Xc[2] < a.depth_min) || (e > a.hard_clamp) || (Xc[2] > a.depth_max);  // :558-565
        // depth-supervised mapping (use_depth, ace_trainer.py:567-574): a prediction further than 10 cm from an available
        // ground-truth scene coordinate is treated as invalid too
        float tcd[3] = {0.f, 0.f, 0.f}, tdist = 0.f;
        bool tavail = false;
        if (a.target_crds) {
          const float* tc = a.target_crds + pre_p * 3;
          tavail = (fabsf(tc[0]) + fabsf(tc[1]) + fabsf(tc[2])) > 0.00001f;
#pragma unroll
          for (int k = 0; k < 3; ++k) tcd[k] = tc[k] - X[k];
          tdist = sqrtf(tcd[0] * tcd[0] + tcd[1] * tcd[1] + tcd[2] * tcd[2]);
          if (tavail && tdist > 0.1f) invalid = true;
        }
        float dXc[3], dXs[3] = {0.f, 0.f, 0.f};
        if (!invalid) {
          float ge;
          const float wgt = a.st->loss_weight;
          if (a.loss_type == LOSS_TANH || a.loss_type == LOSS_DYNTANH) {
            const float th = tanhf(e / wgt);
            loss = wgt * th;
            ge = 1.f - th * th;
          } else if (a.loss_type == LOSS_L1) {
            const bool big = e > wgt;
            loss = big ? 0.f : e;
            ge = big ? 0.f : 1.f;
          } else if (a.loss_type == LOSS_L1_SQRT) {
            const bool big = e > wgt;
            loss = big ? sqrtf(wgt * e) : e;
            ge = big ? 0.5f * wgt / sqrtf(wgt * e) : 1.f;
          } else {
            const bool big = e > wgt;
            loss = big ? logf(1.f + wgt * e) : e;
            ge = big ? wgt / (1.f + wgt * e) : 1.f;
          }
          inl = (e < a.inlier_px) ? 1.f : 0.f;                      // :585
          const float gu = ge * sgn(du), gv = ge * sgn(dv);
          float dp[3];
          dp[0] = gu / pz;
          dp[1] = gv / pz;
          dp[2] = zclamped ? 0.f : -(gu * pp[0] + gv * pp[1]) / (pz * pz);
          if (a.refine_calibration) fgrad = (dp[0] * Xc[0] + dp[1] * Xc[1]) * a.focal_init * kscale;
#pragma unroll
          for (int j = 0; j < 3; ++j) dXc[j] = K[0 * 3 + j] * dp[0] + K[1 * 3 + j] * dp[1] + K[2 * 3 + j] * dp[2];
        } else if (a.target_crds) {
          // use_depth (ace_trainer.py:601-609): L2 distance to the ground-truth coordinate where there is one, nothing otherwise
          dXc[0] = dXc[1] = dXc[2] = 0.f;
          if (tavail) {
            loss = tdist;
            const float inv = tdist > 0.f ? 1.f / tdist : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) dXs[k] = -tcd[k] * inv;
          }
        } else {
          // proxy target at constant depth, ace_trainer.py:592-600
          const float* Ki = a.view_Kinv + (size_t)view * 9;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const float tgt = a.depth_target * fmaf(Ki[i * 3 + 0], tu, fmaf(Ki[i * 3 + 1], tv, Ki[i * 3 + 2]));
            const float d = tgt - Xc[i];
            loss += fabsf(d);
            dXc[i] = -sgn(d);
          }
        }
        const float invB = a.inv_batch;
        if (a.row_dT) {
          // d(loss)/dT[k][j] = sum_i A[i][k] * dXc[i] * Xh[j], k < 3 (row 3 of T is the constant 0 0 0 1)
          const float Xh[4] = {X[0], X[1], X[2], 1.f};
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float ak = (A[0 * 4 + k] * dXc[0] + A[1 * 4 + k] * dXc[1] + A[2 * 4 + k] * dXc[2]) * invB;
#pragma unroll
            for (int j = 0; j < 4; ++j) a.row_dT[(size_t)m * 12 + k * 4 + j] = ak * Xh[j];
          }
          a.row_image[m] = pre_img;
        }
        float dX[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) dX[k] = (P[0 * 4 + k] * dXc[0] + P[1 * 4 + k] * dXc[1] + P[2 * 4 + k] * dXc[2] + dXs[k]) * invB;
        fgrad *= invB;
        if (a.use_homogeneous) {
          ds[0] = dX[0] / hval;
          ds[1] = dX[1] / hval;
          ds[2] = dX[2] / hval;
          float dh = -(dX[0] * s0 + dX[1] * s1 + dX[2] * s2) / (hval * hval);
          if (hclamped) dh = 0.f;
          if (bx > 20.f) ds[3] = dh;
          else {
            const float z = expf(bx);
            ds[3] = dh * z / (z + 1.f);
          }
        } else {
          ds[0] = dX[0]; ds[1] = dX[1]; ds[2] = dX[2]; ds[3] = 0.f;
        }
      }
    }
    // (fp16 operands: the gradient is propagated scaled by grad_scale and un-scaled where it meets the fp32 optimiser; 1 for bf16)
#pragma unroll
    for (int j = 0; j < 4; ++j) s_ds[r][j] = E::is_f16 ? ds[j] * gscale : ds[j];   // (bf16: no multiply at all -- even an exact one changes how
                                                                                 // the compiler contracts the divisions above: last bits of ds)
    s_red[r][0] = loss; s_red[r][1] = inl; s_red[r][2] = fgrad; s_red[r][3] = 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (!a.idx) return;  // inference: no gradients
  if (ACEZ_DBG(a.dbg) == 2) return;

  // ---- the four waves meet once: phase C reads every row's ds and, in the chain kernel, overwrites activations that the other
  // waves' phase A has read
  sync();

  // ---- phase C: wave wv owns channels 128 wv .. 128 wv + 127 of ALL 32 rows, lane t two of them; the row sums of the fc3 weight
  // gradient and of the fc2 bias gradient (the bf16-rounded dZ values) run over the rows in order inside one lane, so a workgroup
  // emits ONE partial per 32 rows without any cross-wave combine
  {
    const int ch = wv * 128 + 2 * t;
    const int mb = block * 4 * LOSS_ROWS;
    float w3[4][2], gw[4][2], bsum[2] = {0.f, 0.f};
    float amax = 0.f;   // fp16: largest |dZ| this lane produces (scaled units) -> the gradient-scale control of the schedule wave
    // every load of the phase first (row by row, a load was followed by a full wait: 32 serial round trips); loss_kernel: they were
    // requested in front of phase B already (EARLY_C)
    if (!EARLY_C) {
#pragma unroll
      for (int j = 0; j < 4; ++j) w3c[j] = *reinterpret_cast<const uint32_t*>(a.W3 + (size_t)(j < no ? j : 0) * 512 + ch);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t v = (j < no) ? w3c[j] : 0u;
      E::un2(v, w3[j][0], w3[j][1]);
      gw[j][0] = gw[j][1] = 0.f;
    }
    const float (*ds_all)[4] = reinterpret_cast<const float (*)[4]>(scratch + 4 * LOSS_ROWS * 4);
    constexpr int C_UNROLL = LDSACT ? 4 : 4 * LOSS_ROWS;
#pragma unroll C_UNROLL
    for (int r = 0; r < 4 * LOSS_ROWS; ++r) {
      const int m = min(mb + r, n - 1);
      const uint32_t v = LDSACT ? *reinterpret_cast<const uint32_t*>(&Xt[act_off(r, ch)]) : xv[LDSACT ? 0 : r];
      float x[2];
      E::un2(v, x[0], x[1]);
      float d[2] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dsj = ds_all[r][j];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          d[e] = fmaf(dsj, w3[j][e], d[e]);
          gw[j][e] = fmaf(dsj, x[e], gw[j][e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (!(x[e] > 0.f)) d[e] = 0.f;  // relu mask of the fc2 output
      if (E::is_f16) amax = fmaxf(amax, fmaxf(fabsf(d[0]), fabsf(d[1])));
      const uint32_t pk = E::pk2(d[0], d[1]);
      // chain kernel: the gradient replaces the activations in the LDS tile (rows past the end become zero rows); the tile is
      // copied to a.dZ by the caller
      if (LDSACT) *reinterpret_cast<uint32_t*>(&Xt[act_off(r, ch)]) = pk;
      else if (mb + r < n) *reinterpret_cast<uint32_t*>(a.dZ + (size_t)m * 512 + ch) = pk;
      if (mb + r < n) {
        float r0, r1;
        E::un2(pk, r0, r1);
        bsum[0] += r0;   // bias gradient of fc2: the rounded values, in row order
        bsum[1] += r1;
      }
    }
    if (E::is_f16 && a.absmax) absmax_publish(a.absmax, amax);
    *reinterpret_cast<float2*>(a.bias_partials + (size_t)block * 512 + ch) = make_float2(bsum[0], bsum[1]);
    float* gp = a.fc3_partials + (size_t)block * a.fc3_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < no) *reinterpret_cast<float2*>(gp + (size_t)j * 512 + ch) = make_float2(gw[j][0], gw[j][1]);
    if (wv == 0) {
      // fc3 bias gradient and the statistics: sums over the 32 rows in row order
      const float (*red_all)[4] = reinterpret_cast<const float (*)[4]>(scratch + 8 * LOSS_ROWS * 4);
      if (t < no) {
        float fb3 = 0.f;
        for (int r = 0; r < 4 * LOSS_ROWS; ++r) fb3 += ds_all[r][t];
        gp[(size_t)no * 512 + t] = fb3;
      }
      if (t < 3) {
        float stat = 0.f;
        for (int r = 0; r < 4 * LOSS_ROWS; ++r) stat += red_all[r][t];
        a.stat_partials[(size_t)block * 4 + t] = stat;
      }
    }
  }
}

template <class E = EltBf16, int LR = 8>
__global__ __launch_bounds__(256) void loss_kernel(LossArgs a) {
  if ((a.st && !a.st->active) || (a.fault && *a.fault)) return;
  __shared__ float scratch[LOSS_SCRATCH_FLOATS];
  const int wv = threadIdx.x >> 6, t = threadIdx.x & 63;
  loss_body<false, true, E, LR>(a, blockIdx.x, wv, t, nullptr, scratch, LossPre{0, 0, 0, 0.f, 0.f}, [] { __syncthreads(); });
}

// loss_kernel's launch with the batch gather of the NEXT step as extra workgroups behind it (acez_train_step_next on the wgrad_opt path:
// the optimiser launch that used to carry the gather no longer exists). The loss kernel is a latency chain that leaves most wave slots
// of the chip empty (320 workgroups of 16 rows on 256 CUs, two fit per CU), the gather another (index -> row -> metadata): the
// two overlap. The rows go to the OTHER of the trainer's two input buffers / metadata tables -- this step's weight-gradient launch
// still reads the current ones. Eight rows per wavefront and pass: the free slots hold the whole batch in one pass.
template <class E = EltBf16, int LR = 8>
__global__ __launch_bounds__(256) void loss_gather_kernel(LossArgs a, int nblk, const uint16_t* __restrict__ feat, const int64_t* __restrict__ idx_next,
                                                          uint16_t* __restrict__ out, int n_next, GatherMeta meta) {
  const int ng = (int)gridDim.x - nblk;   // the gather workgroups come FIRST: three dependent load levels, the longest chain of the launch
  if ((int)blockIdx.x < ng) {
    const int lane = threadIdx.x & 63;
    const int wave = ((int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x) >> 6;
    const int nwaves = (ng * (int)blockDim.x) >> 6;
    gather_rows<8>(feat, idx_next, out, n_next, wave, nwaves, lane, meta);
    return;
  }
  if ((a.st && !a.st->active) || (a.fault && *a.fault)) return;
  __shared__ float scratch[LOSS_SCRATCH_FLOATS];
  const int wv = threadIdx.x >> 6, t = threadIdx.x & 63;
  loss_body<false, true, E, LR>(a, (int)blockIdx.x - ng, wv, t, nullptr, scratch, LossPre{0, 0, 0, 0.f, 0.f}, [] { __syncthreads(); });
}

// ---------------------------------------------------------------------------------------------------
// grad_reduce: flat gradient = fixed-order sum of the wgrad slabs (wide layers) and of the per-workgroup
// partials of the loss kernel (fc3 + statistics). Deterministic: no atomics anywhere in the step.
// ---------------------------------------------------------------------------------------------------
// tail outputs, one wavefront per output: the biases of the wide layers (column-sum partials written where each dZ is
// produced), fc3 weights/bias, then the 4 statistics. Lane-strided partial sums followed by the fixed butterfly order:
// deterministic; every lane returns the sum. (A two-level "32 outputs x 8 groups" variant with coalesced reads measured
// 2-4x slower: its 80-deep dependent add chains are latency bound.) Shared by grad_reduce_kernel and the fused adamw_kernel.
__device__ __forceinline__ float tail_output(const GradReduceArgs& a, int64_t k, int lane, int64_t& dst) {
  const int64_t n_bias = (int64_t)a.n_layers * 512;
  const int64_t n_fc3 = a.n_params - a.n_wide;
  float acc = 0.f;
  if (k < n_bias) {
    const int layer = (int)(k >> 9), c = (int)(k & 511);
    const float* p = a.bias_partials + (size_t)layer * a.bias_layer_stride + c;
    const int cnt = a.bias_count[layer];
    for (int b = lane; b < cnt; b += 64) acc += p[(size_t)b * 512];
    dst = (int64_t)layer * 262656 + 262144 + c;
  } else if (k < n_bias + n_fc3) {
    const int64_t kk = k - n_bias;
    for (int b = lane; b < a.n_loss_blocks; b += 64) acc += a.fc3_partials[(size_t)b * a.fc3_stride + kk];
    dst = a.n_wide + kk;
  } else {
    const int64_t kk = k - n_bias - n_fc3;
    if (kk < 3) {
      // every wave of the fused optimiser sums the loss for itself (NaN check) before it may store: the first 320 partial rows in ONE
      // memory round trip (a loop of dependent load -> add rounds cost the optimiser 1 us per 64 rows); same additions, same order
      float v[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int b = lane + 64 * u;
        const float x = a.stat_partials[(size_t)min(b, max(a.n_loss_blocks - 1, 0)) * 4 + kk];
        v[u] = b < a.n_loss_blocks ? x : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 5; ++u)
        if (lane + 64 * u < a.n_loss_blocks) acc += v[u];
      for (int b = lane + 320; b < a.n_loss_blocks; b += 64) acc += a.stat_partials[(size_t)b * 4 + kk];
    }
    // slot 3: this rank's rowseq fault word. It rides in the all-reduced bucket, so that a fault on ONE rank makes EVERY rank skip
    // the optimiser step (adamw_kernel) and the replicas stay identical
    else if (a.fault) {
      const uint32_t amax_bits = absmax_all(a.st, lane);
      if (lane == 0) acc = (*a.fault ? 1.f : 0.f) + (f16_overflow(amax_bits) ? 1024.f : 0.f);   // (+ fp16 overflow)
    }
    dst = a.n_params + kk;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  return k < n_bias + n_fc3 ? acc * a.st->inv_grad_scale : acc;   // (fp16: gradients arrive scaled; 1 for bf16, an exact product)
}

// (grad_reduce_kernel itself follows SmallCols below: its tail is reduced with it)

// ---------------------------------------------------------------------------------------------------
// adamw: torch.optim.AdamW single-tensor semantics on the flat fp32 masters, fused with the bf16 recast of
// W and W^T. One workgroup per 64 x 64 weight tile (transposed copy staged through LDS); the biases and
// fc3 are handled by the trailing workgroups.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float adamw_one(float p, float g, float& m, float& v, const AdamScalars& s) {
  // no fma contraction here: the function is inlined at several sites (tile path, small-parameter path, fused path) and
  // every site must round identically -- acez_train_step is bitwise equal to backward + update
#pragma clang fp contract(off)
  p = p * s.decay;                       // p.mul_(1 - lr * wd)
  m = m + (g - m) * s.one_minus_beta1;   // exp_avg.lerp_(grad, 1 - beta1)
  v = v * s.beta2 + s.one_minus_beta2 * g * g;
  const float denom = sqrtf(v) / s.bc2_sqrt + s.eps;
  return p - s.step_size * (m / denom);
}

// Workgroups of the optimiser's part of a launch: the first adamw_small_blocks() handle the small parameters (biases, fc3) and, in
// the fused step, the statistics; then one workgroup per 64 x 64 weight tile. (Small first: each of them is a chain of dependent
// round trips -- partials -> butterfly -> p, m, v -> store -- that should run under the streaming tile blocks, not after them.)
__host__ __device__ inline int adamw_small_blocks(int n_layers, int64_t n_fc3, bool fused) {
  const int64_t n_small = (int64_t)n_layers * 512 + n_fc3;
  return fused ? (int)((n_small + 4 + 63) / 64)   // fused: 64 outputs per workgroup, four lanes each (adamw_small_columns)
               : (int)((n_small + 255) / 256);
}

// The small parameters of the fused step (biases of the wide layers, fc3) and the four statistics: 64 consecutive outputs per
// workgroup, FOUR lanes (a DPP quad) per output. An output is a column sum over partial rows (one per row tile of the GEMM chain, or per
// workgroup of the loss kernel): a load instruction of a wavefront reads 16 adjacent columns of 4 rows -- four full 64-byte lines --
// where tail_output (a wavefront per output, lanes striding the ROWS; round 3's tail_output8: eight outputs) touches 64 different lines with every instruction:
// 193 workgroups of those were the long pole of the optimiser launch (18 us by themselves at batch 5120, against 9.5 us for gather +
// schedule; round 4). The summation ORDER is tail_output's, so the bits are: partial j = rows j, j + 64, ... added in that order, then
// the xor butterfly 32, 16, ..., 1, which pairs partial j with j ^ off at every level -- a fixed binary tree (addition commutes). Lane q
// of the quad holds the partials j = 4 i + q: the levels 32 .. 4 pair i with i ^ 8, 4, 2, 1 inside the lane's registers, the levels 2
// and 1 are two quad permutes.
// LPO lanes per output (4, or 8: half the registers per lane -- wgrad_opt_kernel's multiplier waves request the rows from inside their
// K loop and keep them in registers next to the accumulators): lane q holds the partials j = LPO i + q; the butterfly levels that pair
// i with i ^ (32 / LPO), ... , 1 run in the lane's registers, the last log2(LPO) levels across the lanes of the output.
// issue() requests everything (p, m, v of the parameter and the first 320 partial rows: ONE memory round trip -- as a loop of
// load-then-add rounds the five rounds of the fc3 columns were five dependent round trips, 7 us); finish() reduces, applies the
// optimiser and stores. `late_skip()` (wgrad_opt_kernel: adamw_body's guards) is evaluated inside finish() before anything is stored.
// where the optimiser of the small parameters lives (null params: reduce only -- grad_reduce_kernel's tail)
struct SmallOpt {
  float *params, *m, *v;
  uint16_t* W3b;
  const int64_t* b_off;
  int64_t fc3_off;
  int no, f16;
};
__device__ __forceinline__ SmallOpt small_opt(const AdamArgs& a) { return SmallOpt{a.params, a.m, a.v, a.W3b, a.b_off, a.fc3_off, a.no, a.f16}; }
template <int LPO>
struct SmallCols {
  static_assert(LPO == 4 || LPO == 8, "lanes per output");
  static constexpr int NI = 64 / LPO, TD = 5, PER_BLOCK = 256 / LPO;   // partials per lane; 64-row rounds requested at once; outputs per 256 threads
  float x[TD][NI];
  float p, m, v;
  struct Col {
    const float* base;
    int cnt;
    int64_t stride, dst, ome, k;
  };
  static __device__ __forceinline__ Col decode(const GradReduceArgs& r, const SmallOpt& o, const int b) {
    const int64_t n_bias = (int64_t)r.n_layers * 512, n_fc3 = r.n_params - r.n_wide, n_out = n_bias + n_fc3 + 4;
    Col c;
    c.k = (int64_t)b * PER_BLOCK + ((int)threadIdx.x / LPO);
    c.base = r.stat_partials; c.cnt = 0; c.stride = 0; c.dst = -1; c.ome = -1;
    if (c.k < n_bias) {
      // (the outputs of a wavefront never straddle a layer -- 512 is a multiple of the outputs per wavefront --, so the layer is a scalar:
      // bias_count[layer] / b_off[layer] become scalar loads of the kernel arguments instead of a per-lane load the addresses below wait for)
      const int layer = __builtin_amdgcn_readfirstlane((int)(c.k >> 9)), col = (int)(c.k & 511);
      c.base = r.bias_partials + (size_t)layer * r.bias_layer_stride + col; c.cnt = r.bias_count[layer]; c.stride = 512;
      c.dst = (int64_t)layer * 262656 + 262144 + col;
      if (o.params) c.ome = o.b_off[layer] + col;
    } else if (c.k < n_bias + n_fc3) {
      c.base = r.fc3_partials + (c.k - n_bias); c.cnt = r.n_loss_blocks; c.stride = r.fc3_stride;
      c.dst = r.n_wide + (c.k - n_bias);
      if (o.params) c.ome = o.fc3_off + (c.k - n_bias);
    } else if (c.k < n_out) {
      const int64_t kk = c.k - n_bias - n_fc3;
      if (kk < 3) { c.base = r.stat_partials + kk; c.cnt = r.n_loss_blocks; c.stride = 4; }
      c.dst = r.n_params + kk;
    }
    return c;
  }
  __device__ __forceinline__ void issue(const GradReduceArgs& r, const SmallOpt& o, const int b) {
    const int q = (int)threadIdx.x & (LPO - 1);
    const Col c = decode(r, o, b);
    p = 0.f; m = 0.f; v = 0.f;
    if (q == 0 && c.ome >= 0) { p = o.params[c.ome]; m = o.m[c.ome]; v = o.v[c.ome]; }
    const int last = max(c.cnt - 1, 0);
#pragma unroll
    for (int u = 0; u < TD; ++u) {
      if (__any(64 * u < c.cnt)) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int row = 64 * u + LPO * i + q;
          const float y = c.base[(size_t)min(row, last) * c.stride];   // unconditional load, masked afterwards
          x[u][i] = row < c.cnt ? y : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) x[u][i] = 0.f;
      }
    }
  }
  template <class G>
  __device__ __forceinline__ void finish(const GradReduceArgs& r, const SmallOpt& o, const int b, const AdamScalars& s, G&& late_skip) {
    const int lane = (int)threadIdx.x & 63, q = (int)threadIdx.x & (LPO - 1);
    const int64_t n_bias = (int64_t)r.n_layers * 512, n_fc3 = r.n_params - r.n_wide, n_out = n_bias + n_fc3 + 4;
    const Col c = decode(r, o, b);
    const bool skip = late_skip();
    float acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = 0.f;
#pragma unroll
    for (int u = 0; u < TD; ++u)
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] += x[u][i];   // (+0.f where tail_output adds nothing: acc is never -0, so the bits are the same)
    const int last = max(c.cnt - 1, 0);
    for (int u = TD; __any(64 * u < c.cnt); ++u) {
      float y[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = 64 * u + LPO * i + q;
        const float z = c.base[(size_t)min(row, last) * c.stride];
        y[i] = row < c.cnt ? z : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] += y[i];
    }
#pragma unroll
    for (int off = NI / 2; off >= 1; off >>= 1)
#pragma unroll
      for (int i = 0; i < off; ++i) acc[i] = acc[i] + acc[i + off];
    float g = acc[0];
    if (LPO == 8) g = g + __shfl_xor(g, 4);   // butterfly level 4
    g = ACEZ_DPP_ADD(g, 0x4E, 0xF);           // quad_perm [2,3,0,1]: butterfly level 2
    g = ACEZ_DPP_ADD(g, 0xB1, 0xF);           // quad_perm [1,0,3,2]: butterfly level 1
    if (b == (int)((n_out - 1) / PER_BLOCK) && r.fault) {   // statistics slot 3: the fault word + the fp16 overflow flag (see tail_output)
      const uint32_t amax_bits = absmax_all(r.st, lane);
      if (c.k == n_out - 1) g = (*r.fault ? 1.f : 0.f) + (f16_overflow(amax_bits) ? 1024.f : 0.f);
    }
    if (c.k < n_bias + n_fc3) g *= r.st->inv_grad_scale;   // (fp16: gradients arrive scaled; 1 for bf16, an exact product)
    if (q != 0 || c.dst < 0 || skip) return;
    r.grad[c.dst] = g;
    if (c.ome >= 0) {
      float pp = p, mm = m, vv = v;
      pp = adamw_one(pp, g, mm, vv, s);
      o.params[c.ome] = pp; o.m[c.ome] = mm; o.v[c.ome] = vv;
      const int64_t kf = c.k - n_bias;
      if (kf >= 0 && kf < (int64_t)o.no * 512) o.W3b[kf] = o.f16 ? EltF16::from_f(pp) : f2bf(pp);
    }
  }
};
// small-parameter workgroups of a fused optimiser launch (wgrad_opt_kernel: one per 32 outputs, in the multiplier waves of the first ones)
__host__ __device__ inline int small_cols_blocks(int n_layers, int64_t n_fc3, int lanes_per_output) {
  const int64_t per = 256 / lanes_per_output;
  return (int)(((int64_t)n_layers * 512 + n_fc3 + 4 + per - 1) / per);
}
__device__ __forceinline__ void adamw_small_columns(const AdamArgs& a, const int b, const AdamScalars& s) {
  SmallCols<4> sm;
  const SmallOpt o = small_opt(a);
  sm.issue(a.tail, o, b);
  sm.finish(a.tail, o, b, s, [] { return false; });
}

#ifndef GR_LPO
#define GR_LPO 8   // lanes per output of grad_reduce_kernel's tail (4: 97 workgroups with 80 loads per lane in flight; 8: 193 with 40)
#endif
__host__ __device__ inline int grad_reduce_tail_blocks(int n_layers, int64_t n_fc3) { return small_cols_blocks(n_layers, n_fc3, GR_LPO); }
// Grid: the tail workgroups FIRST (GR_LPO lanes per output: biases, fc3, statistics), then the wide part (split flow only). The tail is the
// launch's long pole -- every lane sums up to 80 partials behind two load levels -- and, dispatched behind the 2052 wide workgroups, it
// started when those were done; every load of a workgroup is requested before the schedule's `active` word is looked at (three dependent
// round trips per workgroup of a launch that is one wave of workgroups long). Round 6: 18.4 -> see profiles/r06_dp_host_probe.log.
__device__ __forceinline__ void grad_reduce_body(const GradReduceArgs& a, const int b) {
  const int tail_blocks = grad_reduce_tail_blocks(a.n_layers, a.n_params - a.n_wide);
#ifdef GR_ABL   // timing-only ablation builds (tools/lib_variant.sh): 1 = no tail, 2 = no wide part
  if ((GR_ABL & 1) && b < tail_blocks) return;
  if ((GR_ABL & 2) && b >= tail_blocks) return;
#endif
  if (b >= tail_blocks) {  // weights: 16-byte loads, slabs summed in slab order
    const int64_t i4 = ((int64_t)(b - tail_blocks) * 256 + threadIdx.x) * 4;
    if (a.skip_wide || i4 >= a.n_wide) return;
    if ((i4 % 262656) >= 262144) return;  // bias slots are produced by the tail path
    float4 acc = *reinterpret_cast<const float4*>(a.slabs + i4);
    float4 v1 = acc;
    if (a.nslabs > 1) v1 = *reinterpret_cast<const float4*>(a.slabs + (size_t)a.slab_stride + i4);
    const int active = a.st ? a.st->active : 1;
    const float inv = a.st->inv_grad_scale;
    if (!active) return;   // (a rowseq fault of this step: the tail's first workgroup raises statistics slot 3)
    for (int s = 1; s < a.nslabs; ++s) {
      const float4 v = s == 1 ? v1 : *reinterpret_cast<const float4*>(a.slabs + (size_t)s * a.slab_stride + i4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    *reinterpret_cast<float4*>(a.grad + i4) = acc;
    return;
  }
  // the tail: 64 outputs per workgroup, four lanes each, coalesced partial-row reads (SmallCols: tail_output's summation order, the same bits;
  // a wavefront per output -- 1538 workgroups whose lanes stride the partial ROWS -- touched 64 cache lines with every load)
  SmallCols<GR_LPO> sm;
  const SmallOpt none{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
  sm.issue(a, none, b);
  if (a.st && !a.st->active) {
    // switched off by a rowseq fault in THIS step (not by the end of the schedule): the bucket keeps whatever it held, but statistics
    // slot 3 must still tell the other ranks, which then all skip the optimiser step
    if (a.fault && *a.fault && b == 0 && threadIdx.x == 0) a.grad[a.n_params + 3] = 1.f;
    return;
  }
  sm.finish(a, none, b, AdamScalars{}, [] { return false; });
}
__global__ __launch_bounds__(256) void grad_reduce_kernel(GradReduceArgs a) { grad_reduce_body(a, (int)blockIdx.x); }

// One workgroup's share of the optimiser step (b = workgroup index inside the optimiser's part of a launch; tileT: 64 x 66 bf16 LDS).
// The body of adamw_kernel and of adamw_pose_kernel (pose_fused.hip), where the pose network's backward runs beside it.
__device__ __forceinline__ void adamw_body(const AdamArgs& a, const int b, uint16_t (*tileT)[66]) {
  const TrainState* st = a.st;
  const int t = threadIdx.x;
  const int nsmall = adamw_small_blocks(a.n_layers, a.n_fc3, a.slabs != nullptr);
  const bool tile = b >= nsmall;
  // ---- tile blocks: every load of the 64 x 64 tile is issued FIRST (before the loss reduction below, which is a memory round trip
  // of its own, and before any store: written as one load / compute / store loop per row group, the loads of row group i + 1 could
  // not be moved above the stores of row group i -- same arrays, no alias information -- and a workgroup made four dependent round
  // trips instead of one)
  const int tb = b - nsmall, layer = tile ? a.layer_lo + tb / 64 : 0, tl = tb % 64;   // (tile blocks cover layers layer_lo .. layer_hi - 1)
  const int r0 = (tl >> 3) * 64, c0 = (tl & 7) * 64, cc = (t & 15) * 4;
  const int64_t woff = a.w_off[layer];
  float4 p4[4], g4[4], m4[4], v4[4], q1[4];
  if (tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (t >> 4) + 16 * i;
      const int64_t o = woff + (int64_t)(r0 + rr) * 512 + c0 + cc;
      p4[i] = *reinterpret_cast<const float4*>(a.params + o);
      m4[i] = *reinterpret_cast<const float4*>(a.m + o);
      v4[i] = *reinterpret_cast<const float4*>(a.v + o);
      g4[i] = *reinterpret_cast<const float4*>((a.slabs ? a.slabs : a.grad) + o);
      if (a.slabs && a.nslabs > 1) q1[i] = *reinterpret_cast<const float4*>(a.slabs + (size_t)a.slab_stride + o);   // the second split-K slab, in the same round trip
    }
  }
  // Everything the guards below need is requested BEHIND the tile loads and before any of it is looked at: the schedule's `active` flag,
  // the chain fault word, the optimiser scalars (scalar loads) and the loss partials of the NaN check (vector loads) -- one memory round
  // trip for the whole prologue. (Round 3 read `active` first, then the tile, then the fault word, then the partials: three dependent
  // round trips in front of the first store of every workgroup of a launch that lives for a single wave of tiles.)
  const int active = st->active;
  const AdamScalars s = st->adam;
  const float inv_scale = st->inv_grad_scale;
  float lossv;
  int fault_now = 0;
  if (a.slabs) {   // fused step: the statistics are still partials; every wave sums the loss for itself
    fault_now = *a.fault;   // a hand-off poll of rowseq_kernel expired in this step: its gradients are garbage, nothing is updated
    int64_t d;
    lossv = tail_output(a.tail, (int64_t)a.n_layers * 512 + a.n_fc3, threadIdx.x & 63, d);
    if (!active || fault_now) return;
  } else {
    if (!active) return;
    lossv = a.grad[a.n_params];
    if (fmodf(a.grad[a.n_params + 3], 1024.f) != 0.f) {   // some rank's fault word, summed by the all-reduce: every rank skips the step and
      if (b == 0 && threadIdx.x == 0) {               // falls back at its next state read
        *a.fault = 1;
        const_cast<TrainState*>(st)->active = 0;
      }
      return;
    }
  }
  if (lossv != lossv) return;  // NaN loss: the reference aborts before the optimiser step (ace_trainer.py:615-617)
  // fp16: an infinity was stored somewhere in the gradient chain of this step -> no update (GradScaler.step, ace_schedule.py:112); the
  // schedule wave lowers the scale. In the split flow the flag arrives all-reduced in statistics slot 3 (+1024 per overflowing rank).
  if (a.f16 && (a.slabs ? f16_overflow(absmax_all(st, threadIdx.x & 63)) : a.grad[a.n_params + 3] >= 1024.f)) return;
  if (tile) {
    if (a.slabs) {   // same additions, in the same order, as grad_reduce_kernel's wide part
      for (int sl = 1; sl < a.nslabs; ++sl) {
        float4 q4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = (t >> 4) + 16 * i;
          if (sl == 1) q4[i] = q1[i];
          else q4[i] = *reinterpret_cast<const float4*>(a.slabs + (size_t)sl * a.slab_stride + woff + (int64_t)(r0 + rr) * 512 + c0 + cc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { g4[i].x += q4[i].x; g4[i].y += q4[i].y; g4[i].z += q4[i].z; g4[i].w += q4[i].w; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { g4[i].x *= inv_scale; g4[i].y *= inv_scale; g4[i].z *= inv_scale; g4[i].w *= inv_scale; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (t >> 4) + 16 * i;
      const int64_t o = woff + (int64_t)(r0 + rr) * 512 + c0 + cc;
      float4 p = p4[i], m = m4[i], v = v4[i];
      const float4 g = g4[i];
      p.x = adamw_one(p.x, g.x, m.x, v.x, s);
      p.y = adamw_one(p.y, g.y, m.y, v.y, s);
      p.z = adamw_one(p.z, g.z, m.z, v.z, s);
      p.w = adamw_one(p.w, g.w, m.w, v.w, s);
      *reinterpret_cast<float4*>(a.params + o) = p;
      *reinterpret_cast<float4*>(a.m + o) = m;
      *reinterpret_cast<float4*>(a.v + o) = v;
      const uint2 pk = a.f16 ? EltF16::pk4(p.x, p.y, p.z, p.w) : pack4(p.x, p.y, p.z, p.w);
      *reinterpret_cast<uint2*>(a.Wb + (size_t)layer * 262144 + (size_t)(r0 + rr) * 512 + c0 + cc) = pk;
      tileT[cc + 0][rr] = (uint16_t)(pk.x & 0xffff);
      tileT[cc + 1][rr] = (uint16_t)(pk.x >> 16);
      tileT[cc + 2][rr] = (uint16_t)(pk.y & 0xffff);
      tileT[cc + 3][rr] = (uint16_t)(pk.y >> 16);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (t >> 4) + 16 * i;  // row of W^T tile = column of W
      uint2 pk;
      pk.x = (uint32_t)tileT[rr][cc + 0] | ((uint32_t)tileT[rr][cc + 1] << 16);
      pk.y = (uint32_t)tileT[rr][cc + 2] | ((uint32_t)tileT[rr][cc + 3] << 16);
      *reinterpret_cast<uint2*>(a.WbT + (size_t)layer * 262144 + (size_t)(c0 + rr) * 512 + r0 + cc) = pk;
    }
  } else {
    // small parameters: biases of the wide layers, fc3 weight + bias
    const int64_t n_bias = (int64_t)a.n_layers * 512;
    if (a.slabs) {   // fused step: the partials are reduced here as well (grad_reduce_kernel's tail, output by output, the same bits)
      adamw_small_columns(a, b, s);
      return;
    }
    const int64_t k = (int64_t)b * 256 + t;
    int64_t o = -1;
    if (k < n_bias) o = a.b_off[k >> 9] + (k & 511);
    else if (k < n_bias + a.n_fc3) o = a.fc3_off + (k - n_bias);
    if (o >= 0) {
      float p = a.params[o], m = a.m[o], v = a.v[o];
      p = adamw_one(p, a.grad[o], m, v, s);
      a.params[o] = p; a.m[o] = m; a.v[o] = v;
      if (k >= n_bias && (k - n_bias) < (int64_t)a.no * 512) a.W3b[k - n_bias] = a.f16 ? EltF16::from_f(p) : f2bf(p);
    }
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  __shared__ uint16_t tileT[64][66];
  adamw_body(a, blockIdx.x, tileT);
}

// W^T of wide layers layer_lo .. from their 16-bit W (acez_trainer_import_weights16: the sharded data-parallel update receives the
// other ranks' layers as 16-bit copies and derives the transposed operand of the input-gradient GEMM locally). One 64 x 64 tile per block.
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ Wb, uint16_t* __restrict__ WbT, int layer_lo) {
  __shared__ uint16_t tileT[64][66];
  const int t = threadIdx.x, b = blockIdx.x;
  const int layer = layer_lo + b / 64, tl = b % 64;
  const int r0 = (tl >> 3) * 64, c0 = (tl & 7) * 64, cc = (t & 15) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = (t >> 4) + 16 * i;
    const uint2 pk = *reinterpret_cast<const uint2*>(Wb + (size_t)layer * 262144 + (size_t)(r0 + rr) * 512 + c0 + cc);
    tileT[cc + 0][rr] = (uint16_t)(pk.x & 0xffff);
    tileT[cc + 1][rr] = (uint16_t)(pk.x >> 16);
    tileT[cc + 2][rr] = (uint16_t)(pk.y & 0xffff);
    tileT[cc + 3][rr] = (uint16_t)(pk.y >> 16);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = (t >> 4) + 16 * i;
    uint2 pk;
    pk.x = (uint32_t)tileT[rr][cc + 0] | ((uint32_t)tileT[rr][cc + 1] << 16);
    pk.y = (uint32_t)tileT[rr][cc + 2] | ((uint32_t)tileT[rr][cc + 3] << 16);
    *reinterpret_cast<uint2*>(WbT + (size_t)layer * 262144 + (size_t)(c0 + rr) * 512 + r0 + cc) = pk;
  }
}

// recast only (no optimiser step): used after loading weights
// The sharded data-parallel update in ONE launch on the receiving side: every wide layer OUTSIDE [own_lo, own_hi) is taken from the
// all-gather buffer `src` ([L][512][512] 16-bit, all ranks' layers) into W and, transposed, into W^T (the rank's own layers were written
// by its optimiser launch). Replaces one copy + one transpose launch per peer (2 (G - 1) small launches per step at G ranks).
__global__ __launch_bounds__(256) void import16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ Wb, uint16_t* __restrict__ WbT,
                                                       int own_lo, int own_hi) {
  __shared__ uint16_t tileT[64][66];
  const int t = threadIdx.x, b = blockIdx.x;
  const int layer = b / 64, tl = b % 64;
  if (layer >= own_lo && layer < own_hi) return;
  const int r0 = (tl >> 3) * 64, c0 = (tl & 7) * 64, cc = (t & 15) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = (t >> 4) + 16 * i;
    const size_t o = (size_t)layer * 262144 + (size_t)(r0 + rr) * 512 + c0 + cc;
    const uint2 pk = *reinterpret_cast<const uint2*>(src + o);
    *reinterpret_cast<uint2*>(Wb + o) = pk;
    tileT[cc + 0][rr] = (uint16_t)(pk.x & 0xffff);
    tileT[cc + 1][rr] = (uint16_t)(pk.x >> 16);
    tileT[cc + 2][rr] = (uint16_t)(pk.y & 0xffff);
    tileT[cc + 3][rr] = (uint16_t)(pk.y >> 16);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = (t >> 4) + 16 * i;
    uint2 pk;
    pk.x = (uint32_t)tileT[rr][cc + 0] | ((uint32_t)tileT[rr][cc + 1] << 16);
    pk.y = (uint32_t)tileT[rr][cc + 2] | ((uint32_t)tileT[rr][cc + 3] << 16);
    *reinterpret_cast<uint2*>(WbT + (size_t)layer * 262144 + (size_t)(c0 + rr) * 512 + r0 + cc) = pk;
  }
}

__global__ __launch_bounds__(256) void recast_kernel(AdamArgs a) {
  __shared__ uint16_t tileT[64][66];
  const int t = threadIdx.x, b = blockIdx.x;
  if (b < a.n_layers * 64) {
    const int layer = b / 64, tl = b % 64;
    const int r0 = (tl >> 3) * 64, c0 = (tl & 7) * 64;
    const int64_t woff = a.w_off[layer];
    const int cc = (t & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (t >> 4) + 16 * i;
      const float4 p = *reinterpret_cast<const float4*>(a.params + woff + (int64_t)(r0 + rr) * 512 + c0 + cc);
      const uint2 pk = a.f16 ? EltF16::pk4(p.x, p.y, p.z, p.w) : pack4(p.x, p.y, p.z, p.w);
      *reinterpret_cast<uint2*>(a.Wb + (size_t)layer * 262144 + (size_t)(r0 + rr) * 512 + c0 + cc) = pk;
      tileT[cc + 0][rr] = (uint16_t)(pk.x & 0xffff);
      tileT[cc + 1][rr] = (uint16_t)(pk.x >> 16);
      tileT[cc + 2][rr] = (uint16_t)(pk.y & 0xffff);
      tileT[cc + 3][rr] = (uint16_t)(pk.y >> 16);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (t >> 4) + 16 * i;
      uint2 pk;
      pk.x = (uint32_t)tileT[rr][cc + 0] | ((uint32_t)tileT[rr][cc + 1] << 16);
      pk.y = (uint32_t)tileT[rr][cc + 2] | ((uint32_t)tileT[rr][cc + 3] << 16);
      *reinterpret_cast<uint2*>(a.WbT + (size_t)layer * 262144 + (size_t)(c0 + rr) * 512 + r0 + cc) = pk;
    }
  } else {
    const int64_t k = (int64_t)(b - a.n_layers * 64) * 256 + t;
    if (k < (int64_t)a.no * 512) a.W3b[k] = a.f16 ? EltF16::from_f(a.params[a.fc3_off + k]) : f2bf(a.params[a.fc3_off + k]);
  }
}

// ---------------------------------------------------------------------------------------------------
// schedule (one thread): ace_schedule.py:72-101 before the step, :115-126 after it. The learning rate is
// advanced with the same chainable recurrences torch.optim.lr_scheduler uses, in double.
// ---------------------------------------------------------------------------------------------------
__device__ double det_cos_pi(double x);  // cos(pi * x), x in [0, 1]

__device__ double onecycle_lr(const SchedConfig& c, int step_num) {
  // torch OneCycleLR(max_lr, total_steps, pct_start=0.3, anneal='cos', div_factor=25, final_div_factor=1e4,
  //                  three_phase=False, cycle_momentum=False)
  const double initial_lr = c.lr_max / 25.0;
  const double min_lr = initial_lr / 1e4;
  const double end1 = 0.3 * (double)c.iterations - 1.0;
  const double end2 = (double)c.iterations - 1.0;
  double start, end, pct;
  if ((double)step_num <= end1 || c.iterations <= 1) {
    start = initial_lr; end = c.lr_max;
    pct = (end1 > 0) ? (double)step_num / end1 : 1.0;
  } else {
    start = c.lr_max; end = min_lr;
    pct = ((double)step_num - end1) / (end2 - end1);
  }
  const double cos_out = det_cos_pi(pct) + 1.0;
  return end + (start - end) / 2.0 * cos_out;
}

// The scalar part of TrainState (everything but the criterion ring) as a struct of named locals: the schedule code runs on such a
// copy, loaded with ONE batch of loads and written back once. Operating on `st->field` directly made every access its own global
// round trip (no alias information: ~25 dependent L2 accesses, 5.5 us for one wavefront -- as long as the whole batch gather it
// shares a launch with).
struct SchedHot {
  int active, iteration, max_iterations, in_cooldown, warmup_epoch, cooldown_epoch, nan_flag, opt_steps, crit_count, crit_pos, calib_steps;
  float loss_weight, last_loss, last_inliers;
  double lr, calib_g, calib_m, calib_v, beta1_pow, beta2_pow;
  AdamScalars adam;
  int pose_enable, pose_opt_steps;
  double pose_b1pow, pose_b2pow;
  AdamScalars pose_adam;
  float grad_scale, inv_grad_scale;
};
__device__ __forceinline__ SchedHot load_hot(const TrainState* st) {
  SchedHot h;
  h.active = st->active; h.iteration = st->iteration; h.max_iterations = st->max_iterations; h.in_cooldown = st->in_cooldown;
  h.warmup_epoch = st->warmup_epoch; h.cooldown_epoch = st->cooldown_epoch; h.nan_flag = st->nan_flag; h.opt_steps = st->opt_steps;
  h.crit_count = st->crit_count; h.crit_pos = st->crit_pos; h.calib_steps = st->calib_steps;
  h.loss_weight = st->loss_weight; h.last_loss = st->last_loss; h.last_inliers = st->last_inliers;
  h.lr = st->lr; h.calib_g = st->calib_g; h.calib_m = st->calib_m; h.calib_v = st->calib_v; h.beta1_pow = st->beta1_pow; h.beta2_pow = st->beta2_pow;
  h.adam = st->adam;
  h.pose_enable = st->pose_enable; h.pose_opt_steps = st->pose_opt_steps; h.pose_b1pow = st->pose_b1pow; h.pose_b2pow = st->pose_b2pow;
  h.pose_adam = st->pose_adam;
  h.grad_scale = st->grad_scale; h.inv_grad_scale = st->inv_grad_scale;
  return h;
}
__device__ __forceinline__ void store_hot(TrainState* st, const SchedHot& h) {
  st->active = h.active; st->iteration = h.iteration; st->max_iterations = h.max_iterations; st->in_cooldown = h.in_cooldown;
  st->warmup_epoch = h.warmup_epoch; st->cooldown_epoch = h.cooldown_epoch; st->nan_flag = h.nan_flag; st->opt_steps = h.opt_steps;
  st->crit_count = h.crit_count; st->crit_pos = h.crit_pos; st->calib_steps = h.calib_steps;
  st->loss_weight = h.loss_weight; st->last_loss = h.last_loss; st->last_inliers = h.last_inliers;
  st->lr = h.lr; st->calib_g = h.calib_g; st->calib_m = h.calib_m; st->calib_v = h.calib_v; st->beta1_pow = h.beta1_pow; st->beta2_pow = h.beta2_pow;
  st->adam = h.adam;
  st->pose_enable = h.pose_enable; st->pose_opt_steps = h.pose_opt_steps; st->pose_b1pow = h.pose_b1pow; st->pose_b2pow = h.pose_b2pow;
  st->pose_adam = h.pose_adam;
  st->grad_scale = h.grad_scale; st->inv_grad_scale = h.inv_grad_scale;
}

__device__ __forceinline__ void sched_prepare_hot(SchedHot& h, const SchedConfig& c, float crit_min) {
  // everything the kernels of iteration `h.iteration` need: cool-down decision, active flag, loss weight, AdamW scalars
  const int it = h.iteration;
  // check_and_set_cooldown(iteration)   ace_schedule.py:72-101
  if (c.schedule == SCHED_1CYCLEPOLY && !h.in_cooldown && it >= c.warmup_iterations) {
    const bool by_duration = it >= (h.max_iterations - c.cooldown_iterations);
    const int cnt = h.crit_count;
    const bool dynamic = (cnt > 0) && ((double)crit_min > c.cooldown_trigger_percent);   // min(buffer) > trigger, ace_schedule.py:86-90
    if (by_duration || dynamic) {
      h.in_cooldown = 1;
      h.cooldown_epoch = 0;
      h.max_iterations = it + c.cooldown_iterations;
    }
  }
  h.active = (it < h.max_iterations && !h.nan_flag) ? 1 : 0;   // ace_trainer.py:509-510
  // loss weight of this iteration (ace_loss.py:55-69)
  float wgt = c.soft_clamp;
  if (c.loss_type == LOSS_DYNTANH) {
    double sw = (double)it / (double)c.iterations;
    if (c.circle_schedule) sw = 1.0 - sqrt(1.0 - sw * sw);
    wgt = (float)((1.0 - sw) * (double)c.soft_clamp + (double)c.soft_clamp_min);
  }
  h.loss_weight = wgt;
  // scalars of torch.optim.AdamW for this step (double like the Python side, then cast like the fp32 kernels);
  // beta^step is kept as a running product (one multiply per step, same value as pow to ~1e-16 relative)
  {
    const double lr = h.lr;
    const double b1p = h.beta1_pow * c.beta1, b2p = h.beta2_pow * c.beta2;  // beta^(opt_steps + 1)
    const double bc1 = 1.0 - b1p;
    const double bc2 = 1.0 - b2p;
    AdamScalars s;
    s.decay = (float)(1.0 - lr * c.weight_decay);
    s.one_minus_beta1 = (float)(1.0 - c.beta1);
    s.beta2 = (float)c.beta2;
    s.one_minus_beta2 = (float)(1.0 - c.beta2);
    s.bc2_sqrt = (float)sqrt(bc2);
    s.eps = (float)c.eps;
    s.step_size = (float)(lr / bc1);
    h.adam = s;
  }
  if (c.pose_refinement) {
    h.pose_enable = (it > c.pose_wait) ? 1 : 0;   // ace_trainer.py:634
    const double b1p = h.pose_b1pow * c.beta1, b2p = h.pose_b2pow * c.beta2;
    AdamScalars s;
    s.decay = (float)(1.0 - c.pose_lr * c.weight_decay);
    s.one_minus_beta1 = (float)(1.0 - c.beta1);
    s.beta2 = (float)c.beta2;
    s.one_minus_beta2 = (float)(1.0 - c.beta2);
    s.bc2_sqrt = (float)sqrt(1.0 - b2p);
    s.eps = (float)c.eps;
    s.step_size = (float)(c.pose_lr / (1.0 - b1p));
    h.pose_adam = s;
  }
}

__device__ void sched_prepare(TrainState* st, const SchedConfig& c, float crit_min) {
  SchedHot h = load_hot(st);
  sched_prepare_hot(h, c, crit_min);
  store_hot(st, h);
}

// after a rowseq fall-back (head_api.hip seq_fault_check): the fault handler had cleared `active`; restore what sched_prepare would set
__global__ void sched_reactivate_kernel(TrainState* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) st->active = (st->iteration < st->max_iterations && !st->nan_flag) ? 1 : 0;
}

// initial state: lr as left by the torch scheduler constructors (ace_schedule.py:12-70)
__global__ void sched_init_kernel(TrainState* st, SchedConfig c) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->active = 0; st->iteration = 0; st->max_iterations = c.iterations; st->in_cooldown = 0;
  st->warmup_epoch = 0; st->cooldown_epoch = 0; st->nan_flag = 0; st->opt_steps = 0; st->crit_count = 0; st->crit_pos = 0;
  st->calib_steps = 0; st->loss_weight = c.soft_clamp; st->last_loss = 0.f; st->last_inliers = 0.f;
  st->calib_g = 0.0; st->calib_m = 0.0; st->calib_v = 0.0; st->beta1_pow = 1.0; st->beta2_pow = 1.0;
  st->pose_enable = 0; st->pose_opt_steps = 0; st->pose_b1pow = 1.0; st->pose_b2pow = 1.0;
  st->grad_scale = c.f16 ? 64.f : 1.f; st->inv_grad_scale = 1.f / st->grad_scale;   // (fp16: adapted after every step, sched_post_wave)
  for (int i = 0; i < 64; ++i) st->dz_absmax_slots[i * 32] = 0u;
  if (c.schedule == SCHED_CONSTANT) st->lr = c.lr_min;
  else if (c.schedule == SCHED_1CYCLEPOLY) st->lr = c.lr_max * (c.warmup_lr / c.lr_max);  // LinearLR._initial_step
  else st->lr = onecycle_lr(c, 0);
  sched_prepare(st, c, 0.f);
}

// Executed by ONE full wavefront (all 64 lanes must call it): the lanes cooperate on the minimum of the cool-down
// criterion ring, lane 0 does the scalar bookkeeping.
// The schedule state lives in TWO slots (acez_trainer::st_slot): the kernels of a step read the slot the host passes them, the
// bookkeeping that closes the step reads that slot (`src`) and writes the OTHER one (`st`), which the host hands to every later
// launch. So the bookkeeping may run beside kernels that still read the old slot -- the optimiser's own launch (adamw_next_kernel) --
// without any ordering between them. Every path through this function leaves `st` a complete state (src == st: in place).
__device__ void sched_post_wave(const TrainState* src, TrainState* st, const SchedConfig& c, const float* grad_stats, float inv_global_batch,
                                float* log_loss, float* log_inl, int log_cap, const int* fault, const float* stat_partials, int n_loss_blocks,
                                const GradReduceArgs* tail = nullptr) {
  const int lane = threadIdx.x & 63;
  const uint32_t my_stripe = __builtin_nontemporal_load(&src->dz_absmax_slots[lane * 32]);
  uint32_t amax_bits = my_stripe;   // fp16: largest propagated gradient of the step (scaled units), over the 64 stripes
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) amax_bits = max(amax_bits, (uint32_t)__shfl_xor((int)amax_bits, off));
  // every load of the wave first: the scalar state (used by lane 0), the statistics, this lane's two ring entries
  SchedHot h = load_hot(src);
  const float ring0 = src->crit_buf[lane], ring1 = (lane + 64 < 100) ? src->crit_buf[lane + 64] : 0.f;
  // tail mode: the loss kernel's partials are requested with the state above, before anything is looked at (they sat behind the branch
  // below: a second dependent round trip of the wave that closes every step)
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (tail) {   // inside the optimiser's own launch (adamw_next_kernel / wgrad_opt_kernel): the reduced statistics are being written by sibling
                // workgroups of this launch, so the wave reduces the loss kernel's partials itself -- tail_output's arithmetic: the same bits
    int64_t d;
    const int64_t k0 = (int64_t)tail->n_layers * 512 + (tail->n_params - tail->n_wide);
    g0 = tail_output(*tail, k0, lane, d); g1 = tail_output(*tail, k0 + 1, lane, d); g2 = tail_output(*tail, k0 + 2, lane, d);
  }
  // (split flow: a rank's fault arrives all-reduced in statistics slot 3, and adamw_body raises the local fault word from it -- in
  // adamw_next_kernel inside the SAME launch as this wave, so the word itself would be a race here; the statistic is not)
  const bool fault_reduced = !tail && fmodf(grad_stats[3], 1024.f) != 0.f;
  if ((fault && *fault) || fault_reduced || !h.active) {
    // the step was abandoned (rowseq fault: no iteration is counted, nothing is logged) or the schedule has ended (state frozen,
    // ace_trainer.py:509-510): the other slot becomes a copy
    if (src != st) {
      st->crit_buf[lane] = ring0;
      if (lane + 64 < 100) st->crit_buf[lane + 64] = ring1;
      st->dz_absmax_slots[lane * 32] = my_stripe;
      if (lane == 0) store_hot(st, h);
    }
    return;
  }
  if (!tail) { g0 = grad_stats[0]; g1 = grad_stats[1]; g2 = grad_stats[2]; }
  const float loss = g0 * inv_global_batch;
  const float inl = g1 * inv_global_batch;
  // minimum of the ring as it will be after this step's push: the entries that stay, and the new value
  float crit_min = inl;
  {
    const int cnt = h.crit_count, pos = h.crit_pos;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = lane + 64 * k;
      const bool keep = i < cnt && !(cnt == 100 && i == pos);
      if (keep) crit_min = fminf(crit_min, k == 0 ? ring0 : ring1);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) crit_min = fminf(crit_min, __shfl_xor(crit_min, off));
  }
  {   // the ring as it is after this step's push (1cyclepoly: the new value replaces entry crit_pos), written by all lanes
    const bool push = c.schedule == SCHED_1CYCLEPOLY;
    const int pos = h.crit_pos;
    st->crit_buf[lane] = (push && lane == pos) ? inl : ring0;
    if (lane + 64 < 100) st->crit_buf[lane + 64] = (push && lane + 64 == pos) ? inl : ring1;
  }
  st->dz_absmax_slots[lane * 32] = 0u;   // the stripes of the state the NEXT step's kernels fold into
  if (lane != 0) return;
  h.last_loss = loss;
  h.last_inliers = inl;
  if (loss != loss) h.nan_flag = 1;   // ace_trainer.py:615-617
  const int it = h.iteration;
  if (it < log_cap) { log_loss[it] = loss; log_inl[it] = inl; }
  // fp16: an overflowing step's head update was skipped by adamw_body / wgrad_opt_kernel (the same test on the same words; split flow: the
  // all-reduced flag in statistics slot 3), as GradScaler.step skips optimizer.step(): AdamW's step count does not advance either
  // (ace_schedule.py:112-113). The pose / calibration optimisers are stepped outside the scaler (ace_trainer.py:634-640): they advance.
  const bool head_skipped = c.f16 && (f16_overflow(amax_bits) || (!tail && grad_stats[3] >= 1024.f));
  if (!head_skipped) {
    h.opt_steps += 1;
    h.beta1_pow *= c.beta1;
    h.beta2_pow *= c.beta2;
  }
  if (c.pose_refinement && h.pose_enable) {
    h.pose_opt_steps += 1;
    h.pose_b1pow *= c.beta1;
    h.pose_b2pow *= c.beta2;
  }
  // calibration refiner: its own AdamW on the scalar g (refine_calibration.py:21-26,58-59)
  if (c.refine_calibration) {
    const double g = (double)g2;
    const double lr = c.calib_lr;
    h.calib_steps += 1;
    double p = h.calib_g;
    p = p * (1.0 - lr * c.weight_decay);
    h.calib_m = h.calib_m + (g - h.calib_m) * (1.0 - c.beta1);
    h.calib_v = h.calib_v * c.beta2 + (1.0 - c.beta2) * g * g;
    const double bc1 = 1.0 - pow(c.beta1, (double)h.calib_steps);
    const double bc2 = 1.0 - pow(c.beta2, (double)h.calib_steps);
    const double denom = sqrt(h.calib_v) / sqrt(bc2) + c.eps;
    p = p - (lr / bc1) * (h.calib_m / denom);
    // fp32 parameter in the reference
    h.calib_g = (double)(float)p;
    h.calib_m = (double)(float)h.calib_m;
    h.calib_v = (double)(float)h.calib_v;
  }
  // scheduler.step()   ace_schedule.py:115-126
  if (c.schedule == SCHED_1CYCLEPOLY) {
    if (h.in_cooldown) {
      // LinearLR(start=1, end=lr_min/lr_max, total_iters=cooldown_iterations), chainable form
      const int e = ++h.cooldown_epoch;
      const double sf = 1.0, ef = c.lr_min / c.lr_max;
      if (e <= c.cooldown_iterations)
        h.lr = h.lr * (1.0 + (ef - sf) / ((double)c.cooldown_iterations * sf + (double)(e - 1) * (ef - sf)));
    } else {
      const int e = ++h.warmup_epoch;
      const double sf = c.warmup_lr / c.lr_max, ef = 1.0;
      if (e <= c.warmup_iterations)
        h.lr = h.lr * (1.0 + (ef - sf) / ((double)c.warmup_iterations * sf + (double)(e - 1) * (ef - sf)));
    }
    // rolling buffer of the last 100 batch_inliers (the entry itself was written above)
    const int pos = h.crit_pos;
    h.crit_pos = (pos + 1 == 100) ? 0 : pos + 1;
    if (h.crit_count < 100) h.crit_count += 1;
  } else if (c.schedule == SCHED_CIRCLE) {
    const int e = ++h.warmup_epoch;
    h.lr = onecycle_lr(c, e);
  }
  if (c.f16) {
    // Gradient scale of the NEXT step (the role of torch.cuda.amp.GradScaler.update, ace_schedule.py:107-113): the power of two that
    // would have put this step's largest propagated gradient near 4096 -- fp16 tops out at 65504: a factor of 16 for the change from
    // one step to the next. After an overflow (an inf was stored: this step's update was skipped by adamw_body, as GradScaler.step
    // skips it) the scale drops by 2^8. Powers of two only: scaling then commutes with fp16 rounding for every value in range.
    int e = ilogbf(h.grad_scale);
    if (f16_overflow(amax_bits)) e -= 8;
    else if (amax_bits != 0u) e = ilogbf(4096.f / (__uint_as_float(amax_bits) * h.inv_grad_scale));
    e = min(16, max(-8, e));
    h.grad_scale = ldexpf(1.f, e);
    h.inv_grad_scale = ldexpf(1.f, -e);
  }
  h.iteration = it + 1;   // ace_trainer.py:495
  sched_prepare_hot(h, c, crit_min);   // bookkeeping of the NEXT iteration, so that a step needs a single schedule launch
  store_hot(st, h);
}

__global__ __launch_bounds__(64) void sched_post_kernel(const TrainState* src, TrainState* st, SchedConfig c, const float* grad_stats, float inv_global_batch,
                                                        float* log_loss, float* log_inl, int log_cap, const int* fault, const float* stat_partials,
                                                        int n_loss_blocks) {
  sched_post_wave(src, st, c, grad_stats, inv_global_batch, log_loss, log_inl, log_cap, fault, stat_partials, n_loss_blocks);
}

// step_begin: the batch gather of iteration i + 1 and, in ONE extra single-wave workgroup, the schedule bookkeeping that
// closes iteration i (the two are independent: the gather does not look at the schedule state; a gather into the scratch
// rows of an inactive step is harmless). Saves a dependent single-thread launch per step.
struct PostArgs {
  const TrainState* src;   // the slot the closing step's kernels read
  TrainState* st;          // the slot the bookkeeping writes (the other one; == src: in place)
  SchedConfig c;
  const float* grad_stats;
  float inv_global_batch;
  float* log_loss;
  float* log_inl;
  int log_cap;
  const int* fault;
  const float* stat_partials;   // the loss kernel's per-workgroup statistics of the step being closed (slot 3: largest |ds|) and their count
  int n_loss_blocks;
};
__global__ __launch_bounds__(256) void step_begin_kernel(const uint16_t* __restrict__ feat, const int64_t* __restrict__ idx,
                                                         uint16_t* __restrict__ out, int n, PostArgs p, GatherMeta meta) {
  if (blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x < 64) sched_post_wave(p.src, p.st, p.c, p.grad_stats, p.inv_global_batch, p.log_loss, p.log_inl, p.log_cap, p.fault, p.stat_partials, p.n_loss_blocks);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = ((gridDim.x - 1) * blockDim.x) >> 6;
  gather_rows(feat, idx, out, n, wave, nwaves, lane, meta);
}

// The optimiser of step k, the batch gather of step k + 1 and the schedule bookkeeping that closes step k in ONE launch (single-GPU
// step with the next batch's indices known: acez_train_step_next). The three do not depend on each other: the bookkeeping reads the
// loss kernel's statistic partials and the state slot of step k and writes the OTHER slot (sched_post_wave), which only later launches
// read. Saves the step_begin launch (5 us + a kernel boundary) of every step: the gather hides under the HBM-bound optimiser.
// (First version: one slot, bookkeeping in the workgroup that finished last, found with a ticket counter -- 1700 atomics on one word
// cost 20 us.)   Grid: the gather blocks, the schedule wave, then adamw_kernel's n_adam blocks (order: below)
__global__ __launch_bounds__(256, 4) void adamw_next_kernel(AdamArgs a, int n_adam, const uint16_t* __restrict__ feat, const int64_t* __restrict__ idx_next,
                                                         uint16_t* __restrict__ out, int n_next, PostArgs p, GatherMeta meta) {
  __shared__ uint16_t tileT[64][66];
  // Grid order (round 6): the gather workgroups FIRST, then the schedule wave, then the optimiser's. The gather is a chain of three dependent
  // load levels (~7 us) and used to sit at the END of the grid, where its workgroups started when optimiser workgroups retired: the launch
  // was adamw_kernel's 12.2 us + 5.2. In front, the chain runs under the optimiser's HBM traffic from the first cycle. <= 128 VGPRs
  // (launch bounds): all 1024 workgroups of the launch are resident at once.
  const int b = (int)blockIdx.x;
  const int gblocks = (int)gridDim.x - 1 - n_adam;
  if (b < gblocks) {
    const int lane = threadIdx.x & 63;
    const int wave = (b * (int)blockDim.x + (int)threadIdx.x) >> 6;
    const int nwaves = (gblocks * (int)blockDim.x) >> 6;
    gather_rows(feat, idx_next, out, n_next, wave, nwaves, lane, meta);
    return;
  }
  if (b == gblocks) {
    if (threadIdx.x < 64) {
      // fused flow (a.slabs): the statistics are being reduced by sibling workgroups of this launch, the wave reduces the loss kernel's
      // partials itself; split flow (acez_train_update_next): they are in the bucket, reduced by grad_reduce_kernel and all-reduced since
      // (two call sites: a run-time choice between &a.tail and null makes the kernel arguments addressable -- the compiler copies them to
      // scratch, 680 bytes per lane, and the launch takes 51 us instead of 22)
      if (a.slabs) {
        sched_post_wave(p.src, p.st, p.c, p.grad_stats, p.inv_global_batch, p.log_loss, p.log_inl, p.log_cap, p.fault, p.stat_partials, p.n_loss_blocks, &a.tail);
      } else {
        sched_post_wave(p.src, p.st, p.c, p.grad_stats, p.inv_global_batch, p.log_loss, p.log_inl, p.log_cap, p.fault, p.stat_partials, p.n_loss_blocks);
      }
    }
    return;
  }
  adamw_body(a, b - gblocks - 1, tileT);
}

// cos(pi x) for x in [0,1] with basic operations only (Taylor around the nearest multiple of 1/2).
__device__ double det_cos_pi(double x) {
  const double PI = 3.14159265358979323846;
  // reduce: cos(pi x) = -sin(pi (x - 1/2))
  const double y = (x - 0.5) * PI;  // in [-pi/2, pi/2]
  const double y2 = y * y;
  // sin(y), Taylor to y^21 (|y| <= pi/2: remainder < 1e-17)
  double term = y, sum = y;
  for (int k = 1; k <= 11; ++k) {
    term = -term * y2 / (double)((2 * k) * (2 * k + 1));
    sum += term;
  }
  return -sum;
}

// ---------------------------------------------------------------------------------------------------
// wgrad_opt: wgrad_kernel's product with the optimiser step of the wide layers as its epilogue (single-GPU fused step). wgrad + adamw
// as two launches round-trip both split-K slabs through HBM (16.8 MB written, 16.8 MB read), start the optimiser's 25 MB of state
// reads only when the gradients are complete, and pay a kernel boundary; here
//  * both row slabs of a 128 x 128 tile are placed on ONE XCD (layer = XCD, 32 workgroups each for the default head). A workgroup
//    finalises the 64 x 128 half tile of its slab index: its two multiplier waves of the OTHER half store their accumulators to the
//    partner's exchange tile (this XCD's L2), wait for the acknowledgement and bump the partner's counter -- rowseq_kernel's hand-off:
//    L2-local atomic, bounded sc1 poll on the other side, no fence, nothing read before it is produced;
//  * the eight loader waves, idle once the last stage is requested, fetch p, m, v of the half tile a dozen stages before the K loop
//    ends, then poll, add own (through LDS) + partner partial in slab order (slab 0 + slab 1: the additions of grad_reduce_kernel /
//    adamw_body, so the parameters are bitwise those of backward + update), apply adamw_one and store p, m, v, W and -- through an
//    LDS transpose -- W^T.
// The guards are adamw_body's (schedule inactive, chain fault, NaN loss, fp16 overflow: nothing is stored); the exchange itself is
// unconditional, so the two partners never disagree about it. A poll that expires raises the trainer's fault word exactly like a
// rowseq hand-off (the host falls back to wgrad_kernel + adamw_kernel at its next state read); the stores of that wave are skipped.
// Host: only when the grid fits the CUs (every workgroup resident) and the placement probe has passed (head_api.hip).
// The four multiplier waves of a workgroup have nothing to do while its loader waves run the optimiser arithmetic (~3.5 us of VALU work
// per CU): in the first `nsmall` workgroups they are the small-parameter workgroups of the optimiser launch (adamw_small_columns: biases,
// fc3, statistics), and wave 0 of the last workgroup is the schedule wave that closes the step (do_post) -- with the next batch
// gathered beside the loss kernel (loss_gather_kernel), a step without pose refinement has no optimiser launch left at all.
// ---------------------------------------------------------------------------------------------------
constexpr int WGO_PF = 17;   // prefetch loads per loader lane: 4 x (p, m, v) float4 + the first five rows of the loss partials
// spin until an LDS counter of this workgroup reaches `target` (the waves of a workgroup meet through these after the K loop instead of
// s_barrier: a barrier would make the loader waves' arithmetic wait for the multiplier waves' small-parameter work and vice versa)
__device__ __forceinline__ void lds_wait_ge(uint32_t* f, uint32_t target) {
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_bump(uint32_t* f) {   // after this wave's earlier LDS operations (the LDS executes a wave's operations in order)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#ifdef ACEZ_DIAG   // timeline of the epilogue (tools/wgo_trace.py): stamp i of this wave
#define WGO_STAMP(i) do { if (o.trace && (threadIdx.x & 63) == 0) o.trace[((size_t)blockIdx.x * 12 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WGO_STAMP(i) do { } while (0)
#endif
template <class E = EltBf16>
__global__ __launch_bounds__(WGRAD_THREADS) void wgrad_opt_kernel(WgradArgs a, WgradOptArgs o, PostArgs post) {
  constexpr int RING = ACEZ_WGRAD_RING;
  static_assert(RING >= 3, "two free ring slots are used as staging areas after the K loop");
  __shared__ __attribute__((aligned(16))) uint16_t smem[RING][2][64 * 128];
  __shared__ uint32_t wsync[2];   // [0]: multiplier waves that have staged their accumulators, [1]: loader waves that have written their W^T rows
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  // workgroup b runs on XCD b % 8: both slabs of all 16 tiles of a layer on one XCD (placement affects speed AND the hand-off, hence the probe)
  const int b = blockIdx.x;
  const int xcd = b & 7, jx = b >> 3;
  const int layer = xcd + 8 * (jx >> 5), slab = (jx >> 4) & 1, tile = jx & 15;
  if (layer >= a.n_layers) return;
  if (t < 2) wsync[t] = 0;   // (ordered before its first use by the barriers of the K loop; a slab without rows has none)
  WGO_STAMP(0);   // kernel entry
  const int n0 = (tile >> 2) * 128, c0 = (tile & 3) * 128;
  const AdamArgs& ad = o.ad;
  // this lane's share of the workgroup's 64 x 128 half tile in the epilogue (loader waves): rows rb + 16 i, columns col4 .. col4 + 3
  const int e = t - 256, rb = e >> 5, col4 = (e & 31) * 4;
  const int64_t woff = ad.w_off[layer];
  const int nrow0 = n0 + 64 * slab;   // first row of the half tile this workgroup finalises
  float4 p4[4], m4[4], v4[4];
  float lp[5];   // the loss partials tail_output reads first (NaN guard)
  uint32_t amx = 0u, amxm = 0u;   // fp16: this lane's stripe of the step's largest propagated gradient (final before this launch), prefetched with p / m / v
  const int nlb = ad.tail.n_loss_blocks;
  auto prefetch = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t off = woff + (int64_t)(nrow0 + rb + 16 * i) * 512 + c0 + col4;
      p4[i] = *reinterpret_cast<const float4*>(ad.params + off);
      m4[i] = *reinterpret_cast<const float4*>(ad.m + off);
      v4[i] = *reinterpret_cast<const float4*>(ad.v + off);
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) lp[u] = ad.tail.stat_partials[(size_t)min(l + 64 * u, max(nlb - 1, 0)) * 4];
    if (E::is_f16) amx = __builtin_nontemporal_load(&ad.st->dz_absmax_slots[l * 32]);
  };
  // the multiplier waves' small-parameter share (workgroups b < nsmall): requested once the accumulators are on their way. (In front of
  // the send it delayed every exchange by the 2.5 us the requests take to issue -- tools/wgo_trace.py; from inside the K loop, two or
  // eight stages before its end, it bought nothing or stalled the loop on the loaded registers one iteration later: measured, round 4.)
  SmallCols<8> sm;
  float lpm[5];
  const bool do_small = b < o.nsmall && !(ACEZ_DBG(a.dbg) & 64);
  auto mprefetch = [&]() {
    if (do_small) {
      sm.issue(ad.tail, small_opt(ad), b);
#pragma unroll
      for (int u = 0; u < 5; ++u) lpm[u] = ad.tail.stat_partials[(size_t)min(l + 64 * u, max(nlb - 1, 0)) * 4];
      if (E::is_f16) amxm = __builtin_nontemporal_load(&ad.st->dz_absmax_slots[l * 32]);
    }
  };
  // the sum of the loss partials from five prefetched rows per lane: ONLY its NaN-ness is used (the guard tail_output applies to the step;
  // the logged loss is tail_output's own fixed-order sum), so the cross-lane part is a DPP sum + one v_readlane instead of six ds_bpermute
  // rounds on every wave's path from the K loop to its epilogue (a NaN stays a NaN in any order; the partials are >= 0)
  auto loss_sum = [&](const float (&v5)[5]) {
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 5; ++u)
      if (l + 64 * u < nlb) sum += v5[u];
    for (int r = l + 320; r < nlb; r += 64) sum += ad.tail.stat_partials[(size_t)r * 4];
    sum = wave_sum63(sum);
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sum), 63));
  };
  f32x16 acc[2][2];
  int KT;
  const bool loader = wgrad_kloop<E, WGO_PF + (E::is_f16 ? 1 : 0)>(a, smem, layer, slab, tile, acc, KT, prefetch);   // (fp16: + the absmax stripe)
  if (KT == 0) __builtin_amdgcn_s_barrier();   // (wsync)
  WGO_STAMP(1);   // K loop done
  // Slot KT % RING was last written for stage KT - RING and slot (KT + 1) % RING for stage KT - RING + 1: every wave is past the barrier
  // of stage KT - 1, i.e. done with both; the slot of stage KT - 1 itself may still be read by a slower multiplier wave.
  float* const stage = reinterpret_cast<float*>(&smem[KT % RING][0][0]);                                  // [64][128] fp32: own partial
  uint16_t (*const tileT)[66] = reinterpret_cast<uint16_t (*)[66]>(&smem[(KT + 1) % RING][0][0]);         // [128][66]: W^T staging
  const int pair = layer * 16 + tile;
  const TrainState* st = ad.st;
  if (!loader) {
    const int cw = w & 3, wn = cw >> 1, wc = cw & 1, h = l >> 5;
    if (wn != slab) {
      if (!(ACEZ_DBG(a.dbg) & 32)) {
        float* __restrict__ X = o.xch + (size_t)(pair * 2 + (1 - slab)) * 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              X[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 128 + wc * 64 + j * 32 + (l & 31)] = acc[i][j][r];
        ACEZ_VMCNT(0);   // acknowledged by the L2 both workgroups share
        if (l == 0) {
          const uint32_t one = 1;
          asm volatile("global_atomic_add %0, %1, off" ::"v"(o.flags + (size_t)(pair * 2 + (1 - slab)) * 32), "v"(one) : "memory");
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 128 + wc * 64 + j * 32 + (l & 31)] = acc[i][j][r];
      lds_bump(&wsync[0]);   // the own half is in LDS (two waves)
    }
    WGO_STAMP(2);   // multipliers: accumulators sent (acknowledged) / staged
    mprefetch();
    // from here on the multiplier waves are free: the small parameters and the schedule wave run beside the loader waves' arithmetic
    if (do_small) {   // this workgroup's share of the small parameters, under adamw_body's guards
      const int active = st->active;
      const AdamScalars sc = st->adam;
      const int fault_now = *ad.fault;
      sm.finish(ad.tail, small_opt(ad), b, sc, [&] {
        const float lossv = loss_sum(lpm);
        bool skip = !active || fault_now || lossv != lossv;
        if (E::is_f16) skip = skip || f16_overflow(absmax_reduce(amxm));
        return skip;
      });
    }
    WGO_STAMP(3);   // multipliers: small parameters done
    if (o.do_post && b == (int)gridDim.x - 1 && w == 0 && !(ACEZ_DBG(a.dbg) & 128))
      sched_post_wave(post.src, post.st, post.c, post.grad_stats, post.inv_global_batch, post.log_loss, post.log_inl, post.log_cap, post.fault,
                      post.stat_partials, post.n_loss_blocks, &ad.tail);
    WGO_STAMP(4);   // multipliers: (schedule wave) done
    return;
  }
  // ------------------------------------------------------------------ loader waves: the optimiser step of the half tile
  const int active = st->active;
  const AdamScalars s = st->adam;
  const float inv_scale = st->inv_grad_scale;
  const int fault_now = *ad.fault;
  const float lossv = loss_sum(lp);
  bool skip = !active || fault_now || lossv != lossv || (ACEZ_DBG(a.dbg) & 8);   // adamw_body's guards
  if (E::is_f16) skip = skip || f16_overflow(absmax_reduce(amx));
  const bool skip_by_guard = skip;   // (uniform over the launch: every wave evaluates the same words)
  if (!(ACEZ_DBG(a.dbg) & 48)) {   // (ablation bits of the diagnostics build, timing only: 8 = no final stores, 16 = no poll, 32 = no send, 64 = no small parameters, 128 = no schedule wave)
    uint32_t vseen, sseen, spins, timed;
    const uint32_t* flag = o.flags + (size_t)(pair * 2 + slab) * 32;
    uint32_t wgo_target = o.target;
#ifdef ACEZ_DIAG   // fault injection in SOME workgroups (ACEZ_WGO_FAULT_MOD): a partially applied step
    if (o.fault_mod > 0 && b % o.fault_mod == 1) wgo_target += 1u << 20;
#endif
    asm volatile(
        "s_mov_b32 %[spins], 0\n\t"
        "s_mov_b32 %[timed], 0\n"
        "1:\n\t"
        "global_load_dword %[vseen], %[flag], off sc1\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_readfirstlane_b32 %[sseen], %[vseen]\n\t"
        "s_sub_i32 %[sseen], %[sseen], %[target]\n\t"
        "s_cmp_ge_i32 %[sseen], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_sleep " ACEZ_SEQ_SLEEP "\n\t"
        "s_add_u32 %[spins], %[spins], 1\n\t"
        "s_cmp_lt_u32 %[spins], %[limit]\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b32 %[timed], 1\n"
        "2:"
        : [vseen] "=&v"(vseen), [sseen] "=&s"(sseen), [spins] "=&s"(spins), [timed] "=&s"(timed)
        : [flag] "v"(flag), [target] "s"(wgo_target), [limit] "s"(o.spin_limit)
        : "memory", "scc");
    if (timed) {   // the partner never arrived (the two slabs of a tile are not on one XCD after all): fault word, step switched off
      skip = true;
      if (l == 0) {
        // What the host needs to finish this step (wgo_recover, head_api.hip): which rows were not updated, and with what they would have
        // been. The small parameters and the schedule wave of this launch read the fault word long before it is raised and DO apply the
        // step; tiles whose exchange completed are updated too -- so the fall-back completes the step instead of undoing it. Every kernel
        // of the trainer that writes a step's buffers returns at entry while the word is set: the operands are still there.
        if (!skip_by_guard && o.status) {
          o.status[(size_t)b * WGRAD_LOADERS + (w - 4)] = o.epoch * 4u + 3u;
          o.rec->s = s; o.rec->inv_scale = inv_scale; o.rec->M = a.M; o.rec->in0 = a.In[0];
          __hip_atomic_store(&o.rec->epoch, o.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(ad.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(const_cast<int*>(&st->active), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  WGO_STAMP(2);   // loaders: guards evaluated, partner's counter seen
  float4 oth[4];
  {
    const float* __restrict__ Xin = o.xch + (size_t)(pair * 2 + slab) * 8192;
#pragma unroll
    for (int i = 0; i < 4; ++i) oth[i] = *reinterpret_cast<const float4*>(Xin + (rb + 16 * i) * 128 + col4);
  }
  lds_wait_ge(&wsync[0], 2);   // the own half is in LDS
  WGO_STAMP(3);   // loaders: own half staged (the partner's half is still in flight)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rb + 16 * i;
    const float4 own = *reinterpret_cast<const float4*>(stage + row * 128 + col4);
    // slab 0 + slab 1 (fp32 addition commutes: the same bits whichever of the two is `own`), then the un-scaling of the fp16 chain
    float4 g;
    g.x = (own.x + oth[i].x) * inv_scale; g.y = (own.y + oth[i].y) * inv_scale;
    g.z = (own.z + oth[i].z) * inv_scale; g.w = (own.w + oth[i].w) * inv_scale;
    float4 p = p4[i], m = m4[i], v = v4[i];
    p.x = adamw_one(p.x, g.x, m.x, v.x, s);
    p.y = adamw_one(p.y, g.y, m.y, v.y, s);
    p.z = adamw_one(p.z, g.z, m.z, v.z, s);
    p.w = adamw_one(p.w, g.w, m.w, v.w, s);
    const uint2 pk = E::pk4(p.x, p.y, p.z, p.w);
    if (!skip) {
      const int64_t off = woff + (int64_t)(nrow0 + row) * 512 + c0 + col4;
      *reinterpret_cast<float4*>(ad.params + off) = p;
      *reinterpret_cast<float4*>(ad.m + off) = m;
      *reinterpret_cast<float4*>(ad.v + off) = v;
      *reinterpret_cast<uint2*>(ad.Wb + (size_t)layer * 262144 + (size_t)(nrow0 + row) * 512 + c0 + col4) = pk;
    }
    tileT[col4 + 0][row] = (uint16_t)(pk.x & 0xffff);
    tileT[col4 + 1][row] = (uint16_t)(pk.x >> 16);
    tileT[col4 + 2][row] = (uint16_t)(pk.y & 0xffff);
    tileT[col4 + 3][row] = (uint16_t)(pk.y >> 16);
  }
  WGO_STAMP(4);   // loaders: optimiser arithmetic done, p / m / v / W stores issued
  lds_bump(&wsync[1]);
  lds_wait_ge(&wsync[1], WGRAD_LOADERS);   // every loader wave's rows of the transpose tile
  WGO_STAMP(5);   // loaders: all eight waves' transpose rows written
  {
    // W^T: row c0 + cl holds the 64 values n = nrow0 .. nrow0 + 63 of column cl: 128 contiguous bytes, 32 per lane
    const int cl = e >> 2, part = e & 3;
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = (uint32_t)tileT[cl][part * 16 + 2 * k] | ((uint32_t)tileT[cl][part * 16 + 2 * k + 1] << 16);
    // skip_by_guard, not skip: a wave whose own poll timed out still stores its W^T COLUMNS -- they hold the rows of the waves of this
    // workgroup that did complete (their W / parameter rows were stored above); its own rows of every column are rewritten by
    // wgo_recover_kernel. (With `skip` here a partial time-out inside one workgroup left W^T old where W was new.)
    if (!skip_by_guard) {
      uint16_t* dst = ad.WbT + (size_t)layer * 262144 + (size_t)(c0 + cl) * 512 + nrow0 + part * 16;
      *reinterpret_cast<uint4*>(dst) = make_uint4(q[0], q[1], q[2], q[3]);
      *reinterpret_cast<uint4*>(dst + 8) = make_uint4(q[4], q[5], q[6], q[7]);
    }
  }
#ifdef ACEZ_DIAG
  if (o.trace) { ACEZ_VMCNT(0); WGO_STAMP(6); }   // loaders: every store of this wave acknowledged
#endif
}

// The rows of wgrad_opt_kernel's half tiles whose loader wave timed out in launch rec->epoch (WgradOptArgs::status), finished from the
// slabs wgrad_kernel has just recomputed on the faulted step's untouched buffers: slab 0 + slab 1, the step's own AdamW scalars and
// gradient scale, adamw_one -- the arithmetic of the wave that gave up, so the parameters end up bitwise those of the two-launch flow.
// Same grid decode and the same lane -> (row, column) map as wgrad_opt_kernel's loader waves; one 512-thread workgroup per half tile.
template <class E = EltBf16>
__global__ __launch_bounds__(512) void wgo_recover_kernel(AdamArgs ad, const float* __restrict__ slabs, int64_t slab_stride, const uint32_t* __restrict__ status,
                                                          const WgoFaultRec* __restrict__ rec, int n_layers) {
  const int t = threadIdx.x, l = t & 63, lw = t >> 6;
  const int b = blockIdx.x;
  const int xcd = b & 7, jx = b >> 3;
  const int layer = xcd + 8 * (jx >> 5), slab = (jx >> 4) & 1, tile = jx & 15;
  if (layer >= n_layers) return;
  if (status[(size_t)b * WGRAD_LOADERS + lw] != rec->epoch * 4u + 3u) return;
  const AdamScalars s = rec->s;
  const float inv_scale = rec->inv_scale;
  const int n0 = (tile >> 2) * 128, c0 = (tile & 3) * 128;
  const int e = t, rb = e >> 5, col4 = (e & 31) * 4;
  const int64_t woff = ad.w_off[layer];
  const int nrow0 = n0 + 64 * slab;
  (void)l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rb + 16 * i;
    const int64_t off = woff + (int64_t)(nrow0 + row) * 512 + c0 + col4;
    const float4 s0 = *reinterpret_cast<const float4*>(slabs + off);
    const float4 s1 = *reinterpret_cast<const float4*>(slabs + slab_stride + off);
    float4 g;
    g.x = (s0.x + s1.x) * inv_scale; g.y = (s0.y + s1.y) * inv_scale; g.z = (s0.z + s1.z) * inv_scale; g.w = (s0.w + s1.w) * inv_scale;
    float4 p = *reinterpret_cast<const float4*>(ad.params + off), m = *reinterpret_cast<const float4*>(ad.m + off), v = *reinterpret_cast<const float4*>(ad.v + off);
    p.x = adamw_one(p.x, g.x, m.x, v.x, s);
    p.y = adamw_one(p.y, g.y, m.y, v.y, s);
    p.z = adamw_one(p.z, g.z, m.z, v.z, s);
    p.w = adamw_one(p.w, g.w, m.w, v.w, s);
    const uint2 pk = E::pk4(p.x, p.y, p.z, p.w);
    *reinterpret_cast<float4*>(ad.params + off) = p;
    *reinterpret_cast<float4*>(ad.m + off) = m;
    *reinterpret_cast<float4*>(ad.v + off) = v;
    *reinterpret_cast<uint2*>(ad.Wb + (size_t)layer * 262144 + (size_t)(nrow0 + row) * 512 + c0 + col4) = pk;
    uint16_t* wt = ad.WbT + (size_t)layer * 262144 + (size_t)(c0 + col4) * 512 + nrow0 + row;
    wt[0] = (uint16_t)(pk.x & 0xffff); wt[512] = (uint16_t)(pk.x >> 16); wt[1024] = (uint16_t)(pk.y & 0xffff); wt[1536] = (uint16_t)(pk.y >> 16);
  }
}

}  // namespace acez
