// pipe_proto.hip -- VERDICT r3 item 1: take the weight stream out of the per-layer fill. Timing + bit-equality experiment.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I acezero_amd/csrc tools/pipe_proto.hip -o tools/pipe_proto.bin && tools/pipe_proto.bin
//
// rowseq_kernel (the product's one-launch chain) makes every CU pull 208 KiB per layer through its L2 -> LDS path: 53 MB per layer
// over the chip = the L2's aggregate rate, 2.6 us for 1.07 us of matrix work, and the fill cannot start before the layer does.
// Here the chain is a LAYER PIPELINE with stationary weights:
//   * XCD g owns rows [640 g, 640 g + 640) of the batch for the whole chain; its 32 workgroups are (layer 0..7) x (column quarter 0..3);
//   * a workgroup keeps its 128 x 512 weight slice in REGISTERS for the whole launch (4 multiplier waves x 32 columns x 512 K = 128
//     VGPRs of MFMA A-fragments per lane), loaded once;
//   * rows stream through in micro-batches of MB rows: loader waves poll the progress words of the layer before (same XCD, L2-local,
//     no fence -- rowseq_kernel's hand-off), LDS-DMA the MB x 512 input rows into a ring slot, the multiplier waves run
//     16 K-steps of [ds_read_b128 B-fragments] x [register A-fragments], bias + ReLU, store, and bump their own progress word.
// L2 -> CU traffic per chain and CU: 128 KiB of weights once + 640 KiB of activations, against 8 x 208 KiB = 1664 KiB.
// Same MFMA (16x16x32), same K order per output element as rowgemm80 => bit-identical outputs (checked first).
#include "head_kernels.hip"
#include <cstdio>
#include <vector>
#include <cstring>
using namespace acez;

struct PipeArgs {
  const uint16_t* In0;      // layer 0 input [M][512]
  const uint16_t* W;        // [L][512][512]
  const float* bias;        // [L][512]
  uint16_t* out0;           // layer outputs: layer l at out0 + l * out_stride (a pointer table indexed by the layer would go through scratch and come back as a FLAT pointer)
  size_t out_stride;
  uint32_t* prog;           // [8 groups][8 layers][32 words]: word 4 c + w = micro-batches stored by multiplier wave w of quarter c (monotonic)
  uint32_t base;            // value of every word before this launch
  uint32_t limit;           // poll budget
  uint32_t* fault;
  unsigned long long* trace;   // null, or [8 waves][64 micro-batches][4] s_memtime stamps of workgroup (group 0, layer tr_layer, quarter 0)
  int tr_layer;
  int M, L, mode;           // mode bit 0: consumers do not wait (timing only: the floor without the hand-off latency)
};

__device__ __forceinline__ uint32_t row16_umin(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));
  return v;
}

// Knobs (template parameters) of the experiment:
//   MB     rows per micro-batch (16 / 32)              SIG    K-step of the NEXT micro-batch at which a wave publishes the one before
//   PF     K-steps the B-fragment ds_reads run ahead   POLLW  loader waves that poll the upstream words (4: all; 1: loader 0, the others
//   SLP    s_sleep argument between two polls                 read its result from an LDS word)
//   SPREAD progress words of different (group, layer) 4 KiB + 128 B apart instead of on adjacent 128-byte lines
//   ATOM   publish with an L2 atomic add (v1) instead of a plain store of the running count
template <int MB, int SIG, int PF, int POLLW, int SLP, bool SPREAD, bool ATOM, int DBG = 0>   // DBG (timing only): 1 = no B-fragment ds_reads, 2 = no epilogue / stores, 4 = no sched_barrier between K-steps, 8 = unswizzled reads
__global__ __launch_bounds__(512) void pipe_fwd_kernel(PipeArgs a) {
  constexpr int RF = MB / 16;               // row fragments per micro-batch
  constexpr int SLOT = MB * 512;            // elements per ring slot
  constexpr int NS = 128 / MB;              // ring slots: 128 KiB
  constexpr int IPM = MB / 4;               // DMA instructions per loader wave and micro-batch (one 1 KiB row each)
  constexpr int PSTRIDE = SPREAD ? 1024 + 32 : 32;
  __shared__ __attribute__((aligned(16))) uint16_t smem[NS * SLOT + 64];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int layer = jx >> 2, c = jx & 3;
  if (layer >= a.L) return;
  const int M = a.M;
  const int nmb_total = (M + MB - 1) / MB;
  const int per_g = (nmb_total + 7) >> 3;
  const int mb0 = g * per_g;
  const int nmb = max(0, min(per_g, nmb_total - mb0));
  uint32_t* const mine = a.prog + (g * 8 + layer) * PSTRIDE;
  volatile int* const rdy = reinterpret_cast<volatile int*>(smem + NS * SLOT);   // POLLW == 1: loader 0's latest count
  if (POLLW == 1) {
    if (t == 0) *rdy = 0;
    __syncthreads();
  }

  if (w >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - 4;
    const uint16_t* In = layer ? a.out0 + (size_t)(layer - 1) * a.out_stride : a.In0;
    const uint32_t* up = a.prog + (g * 8 + (layer > 0 ? layer - 1 : 0)) * PSTRIDE + (l & 15);
    int issued = 0;
    int ready = (layer == 0 || (a.mode & 1)) ? nmb : 0;
    uint32_t spins = 0;
    auto issue = [&](int mb) {
      uint16_t* slot = smem + (mb % NS) * SLOT;
      const int m0 = (mb0 + mb) * MB;
#pragma unroll
      for (int j = 0; j < IPM; ++j) {
        const int r = lw * IPM + j;
        const uint16_t* src = In + (size_t)min(m0 + r, M - 1) * 512 + ((l ^ (r & 15)) << 3);
        __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(slot + r * 512), 16, 0, 0);
      }
    };
    auto refresh = [&]() {
      if (POLLW == 4 || lw == 0) {
        uint32_t v;
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(up) : "memory");
        v = row16_umin(v);
        ready = (int)(__builtin_amdgcn_readfirstlane(v) - a.base);
        if (POLLW == 1 && l == 0) *rdy = ready;
      } else {
        ready = __builtin_amdgcn_readfirstlane(*rdy);
      }
    };
    const bool tr = a.trace && g == 0 && layer == a.tr_layer && c == 0 && l == 0;
    for (int i = 0; i < nmb; ++i) {
      if (tr && i < 64) a.trace[(w * 64 + i) * 4 + 0] = __builtin_amdgcn_s_memtime();
      const int cap = min(nmb, i + NS - 1);   // slots of micro-batches <= i - 2 have been read (barrier A_{i-1} is behind us)
      while (issued < cap) {
        if (ready <= issued) {
          refresh();
          if (ready <= issued) {
            if (issued > i) break;            // run-ahead only: do not hold up micro-batch i
            if (++spins > a.limit) { if (l == 0) *a.fault = 1; ready = nmb; } else __builtin_amdgcn_s_sleep(SLP);
            continue;
          }
        }
        issue(issued);
        ++issued;
      }
      const int ahead = issued - 1 - i;       // micro-batches requested after i: may stay in flight
      if (tr && i < 64) { a.trace[(w * 64 + i) * 4 + 1] = __builtin_amdgcn_s_memtime(); a.trace[(w * 64 + i) * 4 + 3] = (unsigned long long)issued; }
      switch (ahead) {
        case 0: ACEZ_VMCNT_C(0); break;
        case 1: ACEZ_VMCNT_C(IPM); break;
        case 2: ACEZ_VMCNT_C(2 * IPM); break;
        case 3: ACEZ_VMCNT_C(3 * IPM); break;
        case 4: ACEZ_VMCNT_C(4 * IPM); break;
        case 5: ACEZ_VMCNT_C(5 * IPM); break;
        default: ACEZ_VMCNT_C(6 * IPM); break;
      }
      if (tr && i < 64) a.trace[(w * 64 + i) * 4 + 2] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();           // A_i: micro-batch i has landed; the multipliers are done with i - 1
    }
  } else {
    // ------------------------------------------------------------------ multiplier waves
    const int fr = l & 15, fq = l >> 4;
    const int ncol = c * 128 + w * 32;        // first output column of this wave
    typedef EltBf16 E;
    E::frag fa[2][16];
    {
      const uint16_t* Wl = a.W + (size_t)layer * 512 * 512 + (size_t)ncol * 512;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 16; ++s) fa[i][s] = *reinterpret_cast<const E::frag*>(Wl + (i * 16 + fr) * 512 + s * 32 + fq * 8);
    }
    float4 bias[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bias[i] = *reinterpret_cast<const float4*>(a.bias + layer * 512 + ncol + i * 16 + 4 * fq);
    uint16_t* const out = a.out0 + (size_t)layer * a.out_stride;
    const int coff = ncol + (fq & 1) * 16 + (fq >> 1) * 8;   // after the permlane16 swaps this lane holds 8 consecutive columns
    uint32_t* const myword = mine + c * 4 + w;
    uint32_t done = a.base;                   // running count this wave publishes
    auto publish = [&]() {
      ACEZ_VMCNT(0);                          // this wave's stores are acknowledged by the XCD's L2 -> visible to the next layer's CUs
      ++done;
      if (l == 0) {
        if (ATOM) { const uint32_t one = 1; asm volatile("global_atomic_add %0, %1, off" ::"v"(myword), "v"(one) : "memory"); }
        else asm volatile("global_store_dword %0, %1, off" ::"v"(myword), "v"(done) : "memory");
      }
    };
    auto tile = [&](const f32x4& v4, int i) {   // bias + ReLU + pack of one accumulator tile
      const float b[4] = {bias[i].x, bias[i].y, bias[i].z, bias[i].w};
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(v4[r] + b[r], 0.f);
      return E::pk4(v[0], v[1], v[2], v[3]);
    };
    auto store_rows = [&](uint2 y0, uint2 y1, int row) {
      // odd 16-lane rows of y0 <-> even rows of y1: every lane ends with 16 contiguous bytes of one output row
      auto rx = __builtin_amdgcn_permlane16_swap(y0.x, y1.x, false, false);
      auto ry = __builtin_amdgcn_permlane16_swap(y0.y, y1.y, false, false);
      if (row < M) *reinterpret_cast<uint4*>(out + (size_t)row * 512 + coff) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
    };
    f32x4 prev[2][RF];
    int prev_m0 = 0;
    const bool tr = a.trace && g == 0 && layer == a.tr_layer && c == 0 && l == 0;
    for (int mb = 0; mb < nmb; ++mb) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (tr && mb < 64) a.trace[(w * 64 + mb) * 4 + 0] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();           // A_mb
      if (tr && mb < 64) a.trace[(w * 64 + mb) * 4 + 1] = __builtin_amdgcn_s_memtime();
      const uint16_t* slot = smem + (mb % NS) * SLOT;
      f32x4 acc[2][RF];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < RF; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
      E::frag fb[PF + 1][RF];
      auto rd = [&](int s) {
#pragma unroll
        for (int j = 0; j < RF; ++j) {
          if (DBG & 1) fb[s % (PF + 1)][j] = fa[j & 1][s];
          else if (DBG & 8) fb[s % (PF + 1)][j] = *reinterpret_cast<const E::frag*>(&slot[(j * 16 + fr) * 512 + ((4 * s + fq) << 3)]);
          else fb[s % (PF + 1)][j] = *reinterpret_cast<const E::frag*>(&slot[(j * 16 + fr) * 512 + (((4 * s + fq) ^ fr) << 3)]);
        }
      };
#pragma unroll
      for (int s = 0; s < PF; ++s) rd(s);
      uint2 y0{};
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s + PF < 16) rd(s + PF);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < RF; ++j) acc[i][j] = E::mfma16(fa[i][s], fb[s % (PF + 1)][j], acc[i][j]);
        // the epilogue of the micro-batch BEFORE rides in the shadow of this one's first MFMAs: K-steps 1, 2 -> row fragment 0, 3, 4 -> 1
        if (!(DBG & 2) && mb > 0 && s >= 1 && s <= 2 * RF) {
          const int j = (s - 1) >> 1;
          if ((s - 1) & 1) store_rows(y0, tile(prev[1][j], 1), prev_m0 + j * 16 + fr);
          else y0 = tile(prev[0][j], 0);
        }
        if (s == SIG && mb > 0) publish();
        if (!(DBG & 4)) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < RF; ++j) prev[i][j] = acc[i][j];
      if (tr && mb < 64) a.trace[(w * 64 + mb) * 4 + 2] = __builtin_amdgcn_s_memtime();
      prev_m0 = (mb0 + mb) * MB;
    }
    if (nmb > 0) {
#pragma unroll
      for (int j = 0; j < RF; ++j) store_rows(tile(prev[0][j], 0), tile(prev[1][j], 1), prev_m0 + j * 16 + fr);
      publish();
    }
  }
}

// v3: the epilogue moves to the loader waves. Measured on v2 (DBG variants): of the 2400 cycles a micro-batch takes per stage, the
// MFMAs need 1100, the deferred bias / ReLU / pack / permlane / store code in the multiplier waves adds 800 (VALU blocks and exec-masked
// store branches between the MFMAs) and the B-fragment reads 400. Here a multiplier wave writes its raw fp32 accumulators to one of two
// staging tiles (4 ds_write_b128) and goes on; after the next barrier the loader waves -- idle otherwise, and on the same SIMDs -- read
// the tile row-wise, add the bias, apply the ReLU, pack and store full 256-byte row segments, wait for the acknowledgement and publish.
//   LDS: ring of 3 x 32 KiB input micro-batches + 2 x 16.5 KiB staging tiles ([32][128 + 4] fp32: the 4-word pad spreads rows over banks).
template <int PF, int SLP, int DBG = 0>
__global__ __launch_bounds__(512) void pipe3_fwd_kernel(PipeArgs a) {
  constexpr int MB = 32, RF = 2, SLOT = MB * 512, NS = 3, IPM = 8, PSTRIDE = 32;
  constexpr int STG = 32 * 132;             // floats per staging tile
  __shared__ __attribute__((aligned(16))) uint16_t smem[NS * SLOT + 2 * STG * 2];
  float* const stg = reinterpret_cast<float*>(smem + NS * SLOT);
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int layer = jx >> 2, c = jx & 3;
  if (layer >= a.L) return;
  const int M = a.M;
  const int nmb_total = (M + MB - 1) / MB;
  const int per_g = (nmb_total + 7) >> 3;
  const int mb0 = g * per_g;
  const int nmb = max(0, min(per_g, nmb_total - mb0));
  uint32_t* const mine = a.prog + (g * 8 + layer) * PSTRIDE;

  if (w >= 4) {
    // ------------------------------------------------------------------ loader waves: DMA in, epilogue out
    const int lw = w - 4;
    const uint16_t* In = layer ? a.out0 + (size_t)(layer - 1) * a.out_stride : a.In0;
    uint16_t* const out = a.out0 + (size_t)layer * a.out_stride;
    const uint32_t* up = a.prog + (g * 8 + (layer > 0 ? layer - 1 : 0)) * PSTRIDE + (l & 15);
    uint32_t* const myword = mine + c * 4 + lw;
    uint32_t done = a.base;
    int issued = 0;
    int ready = (layer == 0 || (a.mode & 1)) ? nmb : 0;
    uint32_t spins = 0;
    // epilogue share of this lane: staging row 8 lw + (l >> 3), columns 16 (l & 7) .. + 15
    const int erow = lw * 8 + (l >> 3), ecol = (l & 7) * 16;
    float eb[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = *reinterpret_cast<const float4*>(a.bias + layer * 512 + c * 128 + ecol + 4 * q);
      eb[4 * q] = b4.x; eb[4 * q + 1] = b4.y; eb[4 * q + 2] = b4.z; eb[4 * q + 3] = b4.w;
    }
    auto issue = [&](int mb) {
      uint16_t* slot = smem + (mb % NS) * SLOT;
      const int m0 = (mb0 + mb) * MB;
#pragma unroll
      for (int j = 0; j < IPM; ++j) {
        const int r = lw * IPM + j;
        const uint16_t* src = In + (size_t)min(m0 + r, M - 1) * 512 + ((l ^ (r & 15)) << 3);
        __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(slot + r * 512), 16, 0, 0);
      }
    };
    auto epilogue = [&](int mb) {           // micro-batch mb's accumulators are in staging tile mb & 1
      if (DBG & 2) { ++done; if (l == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(myword), "v"(done) : "memory"); return; }
      const float* sp = stg + (mb & 1) * STG + erow * 132 + ecol;
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(sp + 4 * q);
      uint32_t pk[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pk[2 * q] = pack2(fmaxf(v[q].x + eb[4 * q], 0.f), fmaxf(v[q].y + eb[4 * q + 1], 0.f));
        pk[2 * q + 1] = pack2(fmaxf(v[q].z + eb[4 * q + 2], 0.f), fmaxf(v[q].w + eb[4 * q + 3], 0.f));
      }
      const int row = (mb0 + mb) * MB + erow;
      if (row < M) {
        uint16_t* o = out + (size_t)row * 512 + c * 128 + ecol;
        *reinterpret_cast<uint4*>(o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4*>(o + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      }
      ACEZ_VMCNT(0);                          // stores acknowledged by this XCD's L2 (and every earlier DMA of this wave has landed)
      ++done;
      if (l == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(myword), "v"(done) : "memory");
    };
    for (int i = 0; i < nmb; ++i) {
      const int cap = min(nmb, i + NS - 1);
      while (issued < cap) {
        if (ready <= issued) {
          uint32_t v;
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(up) : "memory");
          v = row16_umin(v);
          ready = (int)(__builtin_amdgcn_readfirstlane(v) - a.base);
          if (ready <= issued) {
            if (issued > i) break;
            if (++spins > a.limit) { if (l == 0) *a.fault = 1; ready = nmb; } else __builtin_amdgcn_s_sleep(SLP);
            continue;
          }
        }
        issue(issued);
        ++issued;
      }
      const int ahead = issued - 1 - i;
      switch (ahead) {
        case 0: ACEZ_VMCNT_C(0); break;
        case 1: ACEZ_VMCNT_C(IPM); break;
        default: ACEZ_VMCNT_C(2 * IPM); break;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();           // A_i: micro-batch i has landed; the multipliers have staged micro-batch i - 1
      if (i > 0) epilogue(i - 1);
    }
    __builtin_amdgcn_s_barrier();             // closing barrier: the last micro-batch is staged
    if (nmb > 0) epilogue(nmb - 1);
  } else {
    // ------------------------------------------------------------------ multiplier waves
    const int fr = l & 15, fq = l >> 4;
    const int ncol = c * 128 + w * 32;
    typedef EltBf16 E;
    E::frag fa[2][16];
    {
      const uint16_t* Wl = a.W + (size_t)layer * 512 * 512 + (size_t)ncol * 512;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 16; ++s) fa[i][s] = *reinterpret_cast<const E::frag*>(Wl + (i * 16 + fr) * 512 + s * 32 + fq * 8);
    }
    for (int mb = 0; mb < nmb; ++mb) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();           // A_mb
      const uint16_t* slot = smem + (mb % NS) * SLOT;
      f32x4 acc[2][RF];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < RF; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
      E::frag fb[PF + 1][RF];
      auto rd = [&](int s) {
#pragma unroll
        for (int j = 0; j < RF; ++j) {
          if (DBG & 1) fb[s % (PF + 1)][j] = fa[j & 1][s];
          else fb[s % (PF + 1)][j] = *reinterpret_cast<const E::frag*>(&slot[(j * 16 + fr) * 512 + (((4 * s + fq) ^ fr) << 3)]);
        }
      };
#pragma unroll
      for (int s = 0; s < PF; ++s) rd(s);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s + PF < 16) rd(s + PF);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < RF; ++j) acc[i][j] = E::mfma16(fa[i][s], fb[s % (PF + 1)][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // raw accumulators -> staging tile mb & 1: lane (fr, fq) of tile (i, j) holds row 16 j + fr, columns 32 w + 16 i + 4 fq .. + 3
      float* sp = stg + (mb & 1) * STG;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < RF; ++j) *reinterpret_cast<f32x4*>(sp + (j * 16 + fr) * 132 + w * 32 + i * 16 + 4 * fq) = acc[i][j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // closing barrier: the last micro-batch is staged
  }
}

// v4: v3 with the loader waves split by role -- waves 4, 5 only DMA (their queue stays deep: a store acknowledgement can only be
// waited for with vmcnt(0), which would drain it), waves 6, 7 only run the epilogue (stage tile -> bias / ReLU / pack -> 256-byte row
// segments -> acknowledgement -> publish). DMA addresses: per-lane byte offsets computed once, a wave-uniform base per micro-batch.
template <int PF, int SLP, int DBG = 0>
__global__ __launch_bounds__(512) void pipe4_fwd_kernel(PipeArgs a) {
  constexpr int MB = 32, RF = 2, SLOT = MB * 512, NS = 3, IPM = 16, PSTRIDE = 32;
  constexpr int STG = 32 * 132;             // floats per staging tile
  __shared__ __attribute__((aligned(16))) uint16_t smem[NS * SLOT + 2 * STG * 2];
  float* const stg = reinterpret_cast<float*>(smem + NS * SLOT);
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int layer = jx >> 2, c = jx & 3;
  if (layer >= a.L) return;
  const int M = a.M;
  const int nmb_total = (M + MB - 1) / MB;
  const int per_g = (nmb_total + 7) >> 3;
  const int mb0 = g * per_g;
  const int nmb = max(0, min(per_g, nmb_total - mb0));
  uint32_t* const mine = a.prog + (g * 8 + layer) * PSTRIDE;
  const bool tr = a.trace && g == 0 && layer == a.tr_layer && c == 0 && l == 0;

  if (w >= 6) {
    // ------------------------------------------------------------------ epilogue waves
    const int ew = w - 6;
    uint16_t* const out = a.out0 + (size_t)layer * a.out_stride;
    uint32_t* const myword = mine + c * 4 + 2 * ew;        // words 4 c + 2 ew, + 1 of this (group, layer): the consumers take the minimum of all 16
    uint32_t done = a.base;
    const int cg = l & 15, er0 = ew * 16 + (l >> 4);        // 8 columns cg * 8 .. + 7 of rows er0 + 4 q
    float eb[8];
    {
      const float4 b0 = *reinterpret_cast<const float4*>(a.bias + layer * 512 + c * 128 + cg * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(a.bias + layer * 512 + c * 128 + cg * 8 + 4);
      eb[0] = b0.x; eb[1] = b0.y; eb[2] = b0.z; eb[3] = b0.w; eb[4] = b1.x; eb[5] = b1.y; eb[6] = b1.z; eb[7] = b1.w;
    }
    auto epilogue = [&](int mb) {
      const float* sp = stg + (mb & 1) * STG + er0 * 132 + cg * 8;
      float4 v[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q][0] = *reinterpret_cast<const float4*>(sp + q * 4 * 132);
        v[q][1] = *reinterpret_cast<const float4*>(sp + q * 4 * 132 + 4);
      }
      const int row0 = (mb0 + mb) * MB + er0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 y = make_uint4(pack2(fmaxf(v[q][0].x + eb[0], 0.f), fmaxf(v[q][0].y + eb[1], 0.f)), pack2(fmaxf(v[q][0].z + eb[2], 0.f), fmaxf(v[q][0].w + eb[3], 0.f)),
                                   pack2(fmaxf(v[q][1].x + eb[4], 0.f), fmaxf(v[q][1].y + eb[5], 0.f)), pack2(fmaxf(v[q][1].z + eb[6], 0.f), fmaxf(v[q][1].w + eb[7], 0.f)));
        if (row0 + 4 * q < M && !(DBG & 2)) *reinterpret_cast<uint4*>(out + (size_t)(row0 + 4 * q) * 512 + c * 128 + cg * 8) = y;
      }
      if (tr && mb < 64) a.trace[(w * 64 + mb) * 4 + 1] = __builtin_amdgcn_s_memtime();
      ACEZ_VMCNT(0);                          // stores acknowledged by this XCD's L2
      ++done;
      if (l == 0) { const uint2 d2 = make_uint2(done, done); asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(myword), "v"(d2) : "memory"); }
      if (tr && mb < 64) a.trace[(w * 64 + mb) * 4 + 2] = __builtin_amdgcn_s_memtime();
    };
    for (int i = 0; i < nmb; ++i) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();           // A_i: the multipliers have staged micro-batch i - 1
      if (tr && i < 64) a.trace[(w * 64 + i) * 4 + 0] = __builtin_amdgcn_s_memtime();
      if (i > 0) epilogue(i - 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // closing barrier
    if (nmb > 0) epilogue(nmb - 1);
  } else if (w >= 4) {
    // ------------------------------------------------------------------ DMA waves
    const int lw = w - 4;
    const uint16_t* In = layer ? a.out0 + (size_t)(layer - 1) * a.out_stride : a.In0;
    const uint32_t* up = a.prog + (g * 8 + (layer > 0 ? layer - 1 : 0)) * PSTRIDE + (l & 15);
    int issued = 0;
    int ready = (layer == 0 || (a.mode & 1)) ? nmb : 0;
    uint32_t spins = 0;
    uint32_t off[IPM];                        // byte offset of this lane's 16 bytes inside a micro-batch, per DMA instruction
#pragma unroll
    for (int j = 0; j < IPM; ++j) { const int r = lw * IPM + j; off[j] = (uint32_t)(r * 1024 + ((l ^ (r & 15)) << 4)); }
    auto issue = [&](int mb) {
      uint16_t* slot = smem + (mb % NS) * SLOT;
      const char* base = reinterpret_cast<const char*>(In) + (size_t)(mb0 + mb) * (MB * 1024);   // wave-uniform
#pragma unroll
      for (int j = 0; j < IPM; ++j)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(base + off[j]), (lvoid_t*)(slot + (lw * IPM + j) * 512), 16, 0, 0);
    };
    for (int i = 0; i < nmb; ++i) {
      if (tr && i < 64) a.trace[(w * 64 + i) * 4 + 0] = __builtin_amdgcn_s_memtime();
      const int cap = min(nmb, i + NS - 1);
      while (issued < cap) {
        if (ready <= issued) {
          uint32_t v;
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(up) : "memory");
          v = row16_umin(v);
          ready = (int)(__builtin_amdgcn_readfirstlane(v) - a.base);
          if (ready <= issued) {
            if (issued > i) break;
            if (++spins > a.limit) { if (l == 0) *a.fault = 1; ready = nmb; } else __builtin_amdgcn_s_sleep(SLP);
            continue;
          }
        }
        issue(issued);
        ++issued;
      }
      if (tr && i < 64) { a.trace[(w * 64 + i) * 4 + 1] = __builtin_amdgcn_s_memtime(); a.trace[(w * 64 + i) * 4 + 3] = (unsigned long long)issued; }
      const int ahead = issued - 1 - i;
      switch (ahead) {
        case 0: ACEZ_VMCNT_C(0); break;
        case 1: ACEZ_VMCNT_C(IPM); break;
        default: ACEZ_VMCNT_C(2 * IPM); break;
      }
      if (tr && i < 64) a.trace[(w * 64 + i) * 4 + 2] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();           // A_i
    }
    __builtin_amdgcn_s_barrier();             // closing barrier
  } else {
    // ------------------------------------------------------------------ multiplier waves
    const int fr = l & 15, fq = l >> 4;
    const int ncol = c * 128 + w * 32;
    typedef EltBf16 E;
    E::frag fa[2][16];
    {
      const uint16_t* Wl = a.W + (size_t)layer * 512 * 512 + (size_t)ncol * 512;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 16; ++s) fa[i][s] = *reinterpret_cast<const E::frag*>(Wl + (i * 16 + fr) * 512 + s * 32 + fq * 8);
    }
    for (int mb = 0; mb < nmb; ++mb) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (tr && mb < 64) a.trace[(w * 64 + mb) * 4 + 0] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();           // A_mb
      if (tr && mb < 64) a.trace[(w * 64 + mb) * 4 + 1] = __builtin_amdgcn_s_memtime();
      const uint16_t* slot = smem + (mb % NS) * SLOT;
      f32x4 acc[2][RF];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < RF; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
      E::frag fb[PF + 1][RF];
      auto rd = [&](int s) {
#pragma unroll
        for (int j = 0; j < RF; ++j) {
          if (DBG & 1) fb[s % (PF + 1)][j] = fa[j & 1][s];
          else fb[s % (PF + 1)][j] = *reinterpret_cast<const E::frag*>(&slot[(j * 16 + fr) * 512 + (((4 * s + fq) ^ fr) << 3)]);
        }
      };
#pragma unroll
      for (int s = 0; s < PF; ++s) rd(s);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s + PF < 16) rd(s + PF);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < RF; ++j) acc[i][j] = E::mfma16(fa[i][s], fb[s % (PF + 1)][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
      float* sp = stg + (mb & 1) * STG;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < RF; ++j) *reinterpret_cast<f32x4*>(sp + (j * 16 + fr) * 132 + w * 32 + i * 16 + 4 * fq) = acc[i][j];
      if (tr && mb < 64) a.trace[(w * 64 + mb) * 4 + 2] = __builtin_amdgcn_s_memtime();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // closing barrier
  }
}

static uint16_t f2bf_host(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 5120, LMAX = 8;
  const size_t PROGW = 64 * (1024 + 32) + 64;
  setvbuf(stdout, nullptr, _IONBF, 0);
  uint16_t *In, *W; float* bias; uint32_t *prog, *fault;
  uint16_t *outA[LMAX], *outB[LMAX];
  CK(hipMalloc(&In, (size_t)M * 512 * 2)); CK(hipMalloc(&W, (size_t)LMAX * 512 * 512 * 2)); CK(hipMalloc(&bias, LMAX * 512 * 4));
  CK(hipMalloc(&prog, PROGW * 4)); CK(hipMemset(prog, 0, PROGW * 4));
  CK(hipMalloc(&fault, 4)); CK(hipMemset(fault, 0, 4));
  for (int i = 0; i < LMAX; ++i) CK(hipMalloc(&outA[i], (size_t)M * 512 * 2));
  CK(hipMalloc(&outB[0], (size_t)LMAX * M * 512 * 2));
  for (int i = 1; i < LMAX; ++i) outB[i] = outB[0] + (size_t)i * M * 512;
  std::vector<uint16_t> h((size_t)M * 512), hw((size_t)LMAX * 512 * 512);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& x : h) x = f2bf_host(rnd());
  for (auto& x : hw) x = f2bf_host(rnd() * 0.12f);
  std::vector<float> hb(LMAX * 512);
  for (auto& x : hb) x = rnd() * 0.1f + 0.02f;
  CK(hipMemcpy(In, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t base16 = 0, base32 = 0;
  uint32_t* prog32 = prog;
  uint32_t* prog16; CK(hipMalloc(&prog16, PROGW * 4)); CK(hipMemset(prog16, 0, PROGW * 4));
  // variants: {MB, SIG, PF, POLLW, SLP, SPREAD, ATOM}
  struct Variant { const char* name; int mb; void (*k)(PipeArgs); };
  const Variant variants[] = {
    {"MB32 sig10 pf3 poll4 slp1              ", 32, pipe_fwd_kernel<32, 10, 3, 4, 1, false, false>},
    {"MB32 sig10 pf3 poll1 slp1              ", 32, pipe_fwd_kernel<32, 10, 3, 1, 1, false, false>},
    {"MB32 sig10 pf3 poll4 slp4              ", 32, pipe_fwd_kernel<32, 10, 3, 4, 4, false, false>},
    {"MB32 sig10 pf3 poll4 slp1 spread       ", 32, pipe_fwd_kernel<32, 10, 3, 4, 1, true, false>},
    {"MB32 sig6  pf3 poll4 slp1              ", 32, pipe_fwd_kernel<32, 6, 3, 4, 1, false, false>},
    {"MB32 sig10 pf2 poll4 slp1              ", 32, pipe_fwd_kernel<32, 10, 2, 4, 1, false, false>},
    {"MB16 sig8  pf3 poll4 slp1              ", 16, pipe_fwd_kernel<16, 8, 3, 4, 1, false, false>},
    {"MB16 sig8  pf3 poll1 slp1 spread       ", 16, pipe_fwd_kernel<16, 8, 3, 1, 1, true, false>},
    {"v3 MB32 pf3 slp1 (loader-side epilogue)", 32, pipe3_fwd_kernel<3, 1>},
    {"v3 MB32 pf4 slp1                       ", 32, pipe3_fwd_kernel<4, 1>},
    {"v4 MB32 pf3 slp1 (DMA / epilogue waves)", 32, pipe4_fwd_kernel<3, 1>},
    {"v3 MB32 pf3 slp1 DBG2 no epilogue      ", 32, pipe3_fwd_kernel<3, 1, 2>},
    {"v3 MB32 pf3 slp1 DBG3 MFMA only        ", 32, pipe3_fwd_kernel<3, 1, 3>},
    {"MB32 DBG1 no ds_read                   ", 32, pipe_fwd_kernel<32, 10, 3, 4, 1, false, false, 1>},
    {"MB32 DBG2 no epilogue                  ", 32, pipe_fwd_kernel<32, 10, 3, 4, 1, false, false, 2>},
    {"MB32 DBG3 no ds_read no epilogue       ", 32, pipe_fwd_kernel<32, 10, 3, 4, 1, false, false, 3>},
    {"MB32 DBG4 no sched_barrier             ", 32, pipe_fwd_kernel<32, 10, 3, 4, 1, false, false, 4>},
    {"MB32 DBG8 unswizzled reads             ", 32, pipe_fwd_kernel<32, 10, 3, 4, 1, false, false, 8>},
    {"MB32 DBG7 MFMA only                    ", 32, pipe_fwd_kernel<32, 10, 3, 4, 1, false, false, 7>},
  };
  const int NVALID = 11;   // the variants after these are timing-only
  const int NV = (int)(sizeof(variants) / sizeof(variants[0]));
  auto run_pipe = [&](int L, int variant, int mode) {
    PipeArgs a{};
    a.In0 = In; a.W = W; a.bias = bias; a.M = M; a.L = L; a.mode = mode; a.limit = 20000; a.fault = fault;
    a.out0 = outB[0]; a.out_stride = (size_t)M * 512;
    const bool mb16 = variants[variant].mb == 16;
    a.prog = mb16 ? prog16 : prog32; a.base = mb16 ? base16 : base32;
    const int MB = variants[variant].mb;
    const int per_g = (((M + MB - 1) / MB) + 7) / 8;
    hipLaunchKernelGGL(variants[variant].k, dim3(32 * L), dim3(512), 0, 0, a);
    (mb16 ? base16 : base32) += (uint32_t)per_g;
  };
  // the progress words of a layer only advance in launches that run that layer: a change of L restarts the count
  auto reset = [&]() {
    (void)hipDeviceSynchronize();
    (void)hipMemset(prog32, 0, PROGW * 4); (void)hipMemset(prog16, 0, PROGW * 4);
    base16 = base32 = 0;
  };
  auto run_ref = [&](int L) {
    for (int i = 0; i < L; ++i) {
      RowGemmArgs g{};
      g.In = i ? outA[i - 1] : In; g.W = W + (size_t)i * 512 * 512; g.bias = bias + i * 512; g.out_main = outA[i]; g.M = M; g.N = 512; g.K = 512;
      g.relu = 1; g.aux_mode = AUX_NONE;
      launch_rowgemm(g, 0);
    }
  };
  if (M % 256) { printf("this experiment wants M %% 256 == 0 (uniform micro-batch count per XCD)\n"); return 1; }
  // bit equality of every layer's output
  std::vector<uint16_t> ra((size_t)M * 512), rb((size_t)M * 512);
  for (int L : {1, 2, 8}) {
    reset();
    for (int variant = 0; variant < NVALID; ++variant) {
      for (int i = 0; i < LMAX; ++i) CK(hipMemset(outB[i], 0xff, (size_t)M * 512 * 2));
      run_ref(L); run_pipe(L, variant, 0);
      CK(hipDeviceSynchronize());
      size_t bad = 0, nz = 0;
      for (int li = 0; li < L; ++li) {
        CK(hipMemcpy(ra.data(), outA[li], ra.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), outB[li], rb.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ra.size(); ++i) { bad += ra[i] != rb[i]; nz += ra[i] != 0; }
      }
      uint32_t f = 0; CK(hipMemcpy(&f, fault, 4, hipMemcpyDeviceToHost));
      printf("L=%d variant=%d: %zu mismatching of %zu (non-zero %zu) fault=%u\n", L, variant, bad, ra.size() * L, nz, f);
      if (f) { CK(hipMemset(fault, 0, 4)); }
    }
  }
  for (int rep = 0; rep < 2; ++rep)
    for (int L : {1, 8}) {
      reset();
      const int n = 100;
      float ms_ref;
      for (int i = 0; i < 10; ++i) run_ref(L);
      CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_ref(L); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1));
      if (rep) printf("L=%d  per-layer launches %7.2f us (%.2f per layer)\n", L, ms_ref * 1e3 / n, ms_ref * 1e3 / n / L);
      for (int variant = 0; variant < NV; ++variant)
        for (int mode : {0, 1}) {
          float ms;
          for (int i = 0; i < 10; ++i) run_pipe(L, variant, mode);
          CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_pipe(L, variant, mode); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
          { uint32_t ff = 0; CK(hipMemcpy(&ff, fault, 4, hipMemcpyDeviceToHost)); if (ff) { printf("FAULT (poll budget expired) at L=%d variant=%d mode=%d\n", L, variant, mode); return 2; } }
          if (rep) printf("L=%d  pipeline %s %s: %7.2f us (%.2f per layer)\n", L, variants[variant].name,
                          mode ? "NO WAIT (timing only)" : "                     ", ms * 1e3 / n, ms * 1e3 / n / L);
        }
    }
  // in-kernel s_memtime trace of one workgroup (variant 0), for L = 1 (free running) and L = 8 (layers 1 and 7)
  {
    unsigned long long* trace; CK(hipMalloc(&trace, 8 * 64 * 4 * 8));
    std::vector<unsigned long long> ht(8 * 64 * 4);
    const int TRV = 10;   // v4
    const int cases[3][2] = {{1, 0}, {8, 1}, {8, 7}};
    for (auto& cs : cases) {
      reset();
      for (int rep = 0; rep < 3; ++rep) {   // the third launch is read
        CK(hipMemset(trace, 0, 8 * 64 * 4 * 8));
        PipeArgs a{};
        a.In0 = In; a.W = W; a.bias = bias; a.M = M; a.L = cs[0]; a.mode = 0; a.limit = 20000; a.fault = fault;
        a.out0 = outB[0]; a.out_stride = (size_t)M * 512; a.prog = prog32; a.base = base32; a.trace = trace; a.tr_layer = cs[1];
        hipLaunchKernelGGL(variants[TRV].k, dim3(32 * cs[0]), dim3(512), 0, 0, a);
        base32 += (uint32_t)((((M + 31) / 32) + 7) / 8);
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(ht.data(), trace, ht.size() * 8, hipMemcpyDeviceToHost));
      const unsigned long long t0 = ht[(4 * 64 + 0) * 4 + 0];
      printf("trace L=%d layer %d (cycles since loader wave 4's first stamp; s_memtime ticks at 100 MHz x ? -- see deltas)\n", cs[0], cs[1]);
      printf("  mb | loader4: top  issued(n)  landed | mult0: at-barrier  past-barrier  staged | epilogue wave 6: past-barrier  stores-issued  published (of mb - 1)\n");
      for (int i = 0; i < 20; ++i)
        printf("  %2d | %8lld %8lld (%2lld) %8lld | %8lld %8lld %8lld | %8lld %8lld %8lld\n", i, (long long)(ht[(4 * 64 + i) * 4 + 0] - t0), (long long)(ht[(4 * 64 + i) * 4 + 1] - t0),
               (long long)ht[(4 * 64 + i) * 4 + 3], (long long)(ht[(4 * 64 + i) * 4 + 2] - t0), (long long)(ht[(0 * 64 + i) * 4 + 0] - t0),
               (long long)(ht[(0 * 64 + i) * 4 + 1] - t0), (long long)(ht[(0 * 64 + i) * 4 + 2] - t0), (long long)(ht[(6 * 64 + i) * 4 + 0] - t0),
               i ? (long long)(ht[(6 * 64 + i - 1) * 4 + 1] - t0) : 0ll, i ? (long long)(ht[(6 * 64 + i - 1) * 4 + 2] - t0) : 0ll);
    }
  }
  uint32_t f = 0; CK(hipMemcpy(&f, fault, 4, hipMemcpyDeviceToHost));
  printf("fault word at exit: %u\n", f);
  return 0;
}
