// seam_proto.hip -- VERDICT r1 item 1(d): "build and measure one 2-layer fused kernel with a same-XCD 4-WG hand-off".
// Timing + bit-equality experiment, not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I acezero_amd/csrc tools/seam_proto.hip -o /tmp/seam && /tmp/seam
//
// rowseq_kernel runs L forward layers (bias + ReLU, the cheapest epilogue of the step) of the 5120 x 512 x 512 GEMM in ONE
// launch with rowgemm80's tiling, roles, ring and K order. Layer l + 1 of row tile mt needs layer l's 80 x 512 output = the
// four column tiles of mt, which rowgemm80's grid decode already places on one XCD: each wave bumps a per-row-tile counter
// in that XCD's L2 after its stores have been acknowledged, the loader waves of the next layer poll it. What the seam can
// hide that a kernel boundary cannot: the next layer's first four W stages (64 KiB of the 208 KiB a workgroup pulls per
// layer) are requested BEFORE the wait, while the epilogue runs.
// MODE bit 0: release AND acquire with agent-scope fences (buffer_wbl2 / buffer_inv sc1); bit 1: only the acquire fence;
// 0: no fence at all -- the stores' acknowledgements and an L1-bypassing poll, valid because producer and consumers share one L2.
#include "head_kernels.hip"
#include <cstdio>
#include <vector>
#include <cstring>
using namespace acez;

// s_waitcnt vmcnt(n) for a value known after unrolling (the K loop is fully unrolled: n folds to a constant and one case survives)
#define ACEZ_VMCNT_DYN(n) do { switch (n) { case 0: ACEZ_VMCNT(0); break; case 2: ACEZ_VMCNT(2); break; case 4: ACEZ_VMCNT(4); break; case 8: ACEZ_VMCNT(8); break; case 3: ACEZ_VMCNT(3); break; case 6: ACEZ_VMCNT(6); break; case 7: ACEZ_VMCNT(7); break; \
  case 9: ACEZ_VMCNT(9); break; case 10: ACEZ_VMCNT(10); break; case 12: ACEZ_VMCNT(12); break; case 13: ACEZ_VMCNT(13); break; case 14: ACEZ_VMCNT(14); break; \
  case 15: ACEZ_VMCNT(15); break; case 18: ACEZ_VMCNT(18); break; case 17: ACEZ_VMCNT(17); break; case 21: ACEZ_VMCNT(21); break; default: ACEZ_VMCNT(0); break; } } while (0)

struct SeqArgs {
  const uint16_t* In;       // layer 0 input
  const uint16_t* W;        // [L][512][512]
  const float* bias;        // [L][512]
  uint16_t* out[8];         // layer outputs
  uint32_t* flags;          // [row tiles], monotonically increasing
  uint32_t base;            // seams completed by earlier launches
  int M, L, mode;
};

// R: ring depth: 4 = the product's; 5 = one more 28 KiB stage in flight (fits beside ONE staging tile: 160 KiB exactly).
// AH: the stage barrier of iteration kt certifies stage kt + 1 (one AHEAD): the multiplier waves request the fragments of stage kt + 1
// before they run the MFMAs of stage kt (two fragment sets in registers), so that neither the LDS latency nor the barrier skew sits
// between a stage's barrier and its first MFMA. Same MFMAs in the same order: bit-identical.
template <int R, bool AH>
__global__ __launch_bounds__(512) void rowseq_kernel(SeqArgs a) {
  constexpr int STAGE = (128 + 96) * 64;
  __shared__ __attribute__((aligned(16))) uint16_t smem[R * STAGE + 80 * 128];
  uint16_t* const stB = smem + R * STAGE;
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int mtiles = (a.M + 79) / 80;
  const int per_xcd = (mtiles + 7) >> 3;
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + (jx >> 2);
  if (mt >= mtiles) return;
  const int n0 = (jx & 3) * 128, m0 = mt * 80;
  const int M = a.M;
  constexpr int K = 512, KT = 8, N = 512;

  for (int layer = 0; layer < a.L; ++layer) {
    const uint16_t* In = layer ? a.out[layer - 1] : a.In;
    const uint16_t* Wl = a.W + (size_t)layer * 512 * 512;
    if (w >= 4) {
      const int lw = w - 4;
      const uint16_t* gW[4];
      const uint16_t* gI[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (lw * 4 + j) * 8 + (l >> 3);
        gW[j] = Wl + (size_t)(n0 + row) * K + ((l & 7) ^ ((row >> 1) & 7)) * 8;
      }
      // mode bit 6 (64): the In part of a stage carries the tile's 80 rows instead of 96 (10 groups of 8 rows: loader waves 0 / 1 request
      // three, waves 2 / 3 two): 7.7 % less fill per layer
      const bool rows80 = (a.mode & 64) != 0;
      const int NI = (rows80 && lw >= 2) ? 2 : 3;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int g = rows80 ? lw + 4 * j : lw * 3 + j;
        const int row = g * 8 + (l >> 3);
        gI[j] = In + (size_t)min(m0 + row, M - 1) * K + ((l & 7) ^ ((row >> 1) & 7)) * 8;
      }
      auto issueW = [&](int kt) {
        uint16_t* slot = smem + (kt % R) * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_global_load_lds((gvoid_t*)(gW[j] + kt * 64), (lvoid_t*)(slot + (lw * 4 + j) * 8 * 64), 16, 0, 0);
      };
      auto issueI = [&](int kt) {
        uint16_t* slot = smem + (kt % R) * STAGE;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < NI) __builtin_amdgcn_global_load_lds((gvoid_t*)(gI[j] + kt * 64), (lvoid_t*)(slot + 128 * 64 + (rows80 ? lw + 4 * j : lw * 3 + j) * 8 * 64), 16, 0, 0);
      };
      if (layer == 0) {   // later layers: requested at the end of the layer before
#pragma unroll
        for (int k = 0; k < R; ++k) issueW(k);
      }
      if (layer > 0 && !(a.mode & 4)) {   // mode bit 2 (4): NO wait at all -- timing only (results are wrong): the ceiling of what hiding the seam can give
        const uint32_t target = (a.base + (uint32_t)layer) * 32u;       // 4 workgroups x 8 waves per seam
        if (a.mode & 1) {
          while ((int32_t)(__hip_atomic_load(&a.flags[mt * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
          // same-XCD hand-off: the counter and the tiles live in THIS XCD's L2; read it past the L1 (sc1), no L2 invalidate
          uint32_t seen;
          do {
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(a.flags + mt * 32) : "memory");
            if ((int32_t)(seen - target) < 0) __builtin_amdgcn_s_sleep(1);
          } while ((int32_t)(seen - target) < 0);
          if (a.mode & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
      }
#pragma unroll
      for (int k = 0; k < R; ++k) issueI(k);
      if (AH) {
        ACEZ_VMCNT_DYN(NI * (R - 1));         // stage 0 has landed
        __builtin_amdgcn_s_barrier();         // #P
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          if (kt + 1 < KT) {
            // stage kt + 1 must have landed. Issued so far, in order: In 0 .. R-1 (3 instructions each), then the refills R, R+1, ... (7 each),
            // one after each of the barriers #1 .. #(kt-1) while stages remain
            const int issued = (kt - 1 < KT - R) ? (kt - 1 > 0 ? kt - 1 : 0) : KT - R;
            const int s1 = kt + 1;
            const int allowed = (s1 < R) ? NI * (R - 1 - s1) + (4 + NI) * issued : (4 + NI) * (issued - (s1 - R + 1));
            ACEZ_VMCNT_DYN(allowed);
          }
          __builtin_amdgcn_s_barrier();       // #kt: stage kt + 1 has landed; the multipliers have read stage kt - 1
          if (kt >= 1 && kt + R - 1 < KT) { issueW(kt + R - 1); issueI(kt + R - 1); }
        }
      } else {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int init_after = (kt < R) ? NI * (R - 1 - kt) : 0;
        const int last_refill = (kt - 1 + R - 1 < KT - 1) ? kt - 1 + R - 1 : KT - 1;   // highest stage requested before this wait
        const int refills = (last_refill >= R && last_refill > kt) ? (last_refill - (kt > R - 1 ? kt : R - 1)) : 0;
        ACEZ_VMCNT_DYN(init_after + (4 + NI) * refills);
        __builtin_amdgcn_s_barrier();
        if (kt >= 1 && kt + R - 1 < KT) { issueW(kt + R - 1); issueI(kt + R - 1); }
      }
      }
      __builtin_amdgcn_s_barrier();       // K loop over: the ring is free
      if (layer + 1 < a.L) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gW[j] += 512 * 512;
#pragma unroll
        for (int k = 0; k < R; ++k) issueW(k);
      }
      __builtin_amdgcn_s_barrier();       // output tile complete
    } else {
      f32x4 acc[2][5];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
      const int fr = l & 15, fq = l >> 4;
      float4 bias[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) bias[i] = *reinterpret_cast<const float4*>(a.bias + layer * 512 + n0 + w * 32 + i * 16 + 4 * fq);
      if (AH) {
        bf16x8 fa[2][2][2], fb[2][2][5];   // [set][kk][fragment]
        auto fetch = [&](int kt, int set) {
          const uint16_t* sW = smem + (kt % R) * STAGE;
          const uint16_t* sI = sW + 128 * 64;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int c = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[set][kk][i] = *reinterpret_cast<const bf16x8*>(&sW[swz(w * 32 + i * 16 + fr, c)]);
#pragma unroll
            for (int j = 0; j < 5; ++j) fb[set][kk][j] = *reinterpret_cast<const bf16x8*>(&sI[swz(j * 16 + fr, c)]);
          }
        };
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();         // #P: stage 0 has landed
        fetch(0, 0);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          // the fragments of stage kt (requested one iteration ago) must be in registers before this wave lets the loaders refill
          // that slot (after barrier #kt + 1); they are: the MFMAs below consume them, and the wait here keeps the scheduler honest
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();       // #kt: stage kt + 1 has landed
          if (kt + 1 < KT) fetch(kt + 1, (kt + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kt & 1][kk][i], fb[kt & 1][kk][j], acc[i][j], 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        // (straight-line code: without this the scheduler leaves the previous stage's last ds_reads in flight across the barrier
        //  that releases the refill of their slot -- see rowgemm80_body)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint16_t* sW = smem + (kt % R) * STAGE;
        const uint16_t* sI = sW + 128 * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int c = kk * 4 + fq;
          bf16x8 fa[2], fb[5];
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(&sW[swz(w * 32 + i * 16 + fr, c)]);
#pragma unroll
          for (int j = 0; j < 5; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(&sI[swz(j * 16 + fr, c)]);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int ml = j * 16 + fr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int nl = w * 32 + i * 16 + 4 * fq;
          float v[4] = {acc[i][j][0] + bias[i].x, acc[i][j][1] + bias[i].y, acc[i][j][2] + bias[i].z, acc[i][j][3] + bias[i].w};
          *reinterpret_cast<uint2*>(&stB[st_off(ml, nl)]) = pack4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    {
      uint16_t* out = a.out[layer];
      const int q0 = t, q1 = t + 512, q2 = t + 1024;
      const int so0 = (q0 >> 4) * 128 + (((q0 & 15) ^ ((q0 >> 4) & 15)) << 3);
      const int so1 = (q1 >> 4) * 128 + (((q1 & 15) ^ ((q1 >> 4) & 15)) << 3);
      const int so2 = (q2 < 1280) ? (q2 >> 4) * 128 + (((q2 & 15) ^ ((q2 >> 4) & 15)) << 3) : 0;
      const uint4 m0v = *reinterpret_cast<const uint4*>(&stB[so0]);
      const uint4 m1v = *reinterpret_cast<const uint4*>(&stB[so1]);
      const uint4 m2v = *reinterpret_cast<const uint4*>(&stB[so2]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int r0 = m0 + (q0 >> 4), r1 = m0 + (q1 >> 4), r2 = m0 + (q2 >> 4);
      const size_t o0 = (size_t)r0 * N + n0 + (q0 & 15) * 8, o1 = (size_t)r1 * N + n0 + (q1 & 15) * 8, o2 = (size_t)r2 * N + n0 + (q2 & 15) * 8;
      if (r0 < M) *reinterpret_cast<uint4*>(out + o0) = m0v;
      if (r1 < M) *reinterpret_cast<uint4*>(out + o1) = m1v;
      if (q2 < 1280 && r2 < M) *reinterpret_cast<uint4*>(out + o2) = m2v;
    }
    if (layer + 1 < a.L) {
      if (a.mode & 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (l == 0) __hip_atomic_fetch_add(&a.flags[mt * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (!(a.mode & 8)) ACEZ_VMCNT(0);  // stores acknowledged by this XCD's L2 = visible to the three sibling workgroups (mode bit 3 (8): not waited for -- timing only)
        if (l == 0) {
          const uint32_t one = 1;
          asm volatile("global_atomic_add %0, %1, off" ::"v"(a.flags + mt * 32), "v"(one) : "memory");
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4: the weight stream out of the LDS ring. rowseq_wreg_kernel keeps rowseq's tiling (80 x 128, one workgroup
// per CU, same K order => bit-identical) but a multiplier wave holds its 32-column x 512 weight panel in REGISTERS (2 x 16 A-fragments
// = 128 VGPRs), loaded with plain global loads right after the wave has signalled the hand-off of the layer before -- the registers are
// dead from the end of the K loop, and the ~1.5 us until the next K loop starts (sibling stores, counter, poll, first In stage) hide the
// 32 KiB per wave. The ring then carries In only: all eight [96 x 64] stages of a layer (96 KiB) are requested at once behind the
// hand-off, no refills, and a K-step reads 5 fragments from LDS instead of 7.
// FRAGW: W is read from a fragment-ordered copy (per workgroup column tile and wave: [K-step 16][column fragment 2][lane 64] x 16 B, every
// load instruction 1 KiB contiguous) instead of row-major (sixteen 64-byte pieces per instruction).
template <bool FRAGW>
__global__ __launch_bounds__(512) void rowseq_wreg_kernel(SeqArgs a) {
  constexpr int STAGE = 96 * 64;
  __shared__ __attribute__((aligned(16))) uint16_t smem[8 * STAGE + 80 * 128];
  uint16_t* const stB = smem + 8 * STAGE;
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int mtiles = (a.M + 79) / 80;
  const int per_xcd = (mtiles + 7) >> 3;
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + (jx >> 2);
  if (mt >= mtiles) return;
  const int nt = jx & 3, n0 = nt * 128, m0 = mt * 80;
  const int M = a.M;
  constexpr int K = 512, KT = 8, N = 512;
  const int fr = l & 15, fq = l >> 4;
  bf16x8 fa[2][16];
  auto load_w = [&](int layer) {
    if (FRAGW) {
      const uint16_t* p = a.W + (size_t)layer * 512 * 512 + (size_t)(nt * 4 + w) * (32 * 512) + l * 8;
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i][s] = *reinterpret_cast<const bf16x8*>(p + (s * 2 + i) * 512);
    } else {
      const uint16_t* p = a.W + (size_t)layer * 512 * 512 + (size_t)(n0 + w * 32) * 512;
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i][s] = *reinterpret_cast<const bf16x8*>(p + (i * 16 + fr) * 512 + s * 32 + fq * 8);
    }
  };
  if (w < 4) load_w(0);

  for (int layer = 0; layer < a.L; ++layer) {
    const uint16_t* In = layer ? a.out[layer - 1] : a.In;
    if (w >= 4) {
      const int lw = w - 4;
      const uint16_t* gI[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int row = (lw * 3 + j) * 8 + (l >> 3);
        gI[j] = In + (size_t)min(m0 + row, M - 1) * K + ((l & 7) ^ ((row >> 1) & 7)) * 8;
      }
      if (layer > 0 && !(a.mode & 4)) {
        const uint32_t target = (a.base + (uint32_t)layer) * 32u;
        uint32_t seen;
        do {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(a.flags + mt * 32) : "memory");
          if ((int32_t)(seen - target) < 0) __builtin_amdgcn_s_sleep(1);
        } while ((int32_t)(seen - target) < 0);
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          __builtin_amdgcn_global_load_lds((gvoid_t*)(gI[j] + kt * 64), (lvoid_t*)(smem + kt * STAGE + (lw * 3 + j) * 8 * 64), 16, 0, 0);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        ACEZ_VMCNT_DYN(3 * (KT - 1 - kt));
        __builtin_amdgcn_s_barrier();       // stage kt has landed
      }
      __builtin_amdgcn_s_barrier();         // K loop over
      __builtin_amdgcn_s_barrier();         // output tile complete
    } else {
      f32x4 acc[2][5];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
      float4 bias[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) bias[i] = *reinterpret_cast<const float4*>(a.bias + layer * 512 + n0 + w * 32 + i * 16 + 4 * fq);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        const uint16_t* sI = smem + kt * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int c = kk * 4 + fq;
          bf16x8 fb[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(&sI[swz(j * 16 + fr, c)]);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][kt * 2 + kk], fb[j], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();         // K loop over (no stage is refilled in this kernel: only the staging tile is reused)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int ml = j * 16 + fr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int nl = w * 32 + i * 16 + 4 * fq;
          float v[4] = {acc[i][j][0] + bias[i].x, acc[i][j][1] + bias[i].y, acc[i][j][2] + bias[i].z, acc[i][j][3] + bias[i].w};
          *reinterpret_cast<uint2*>(&stB[st_off(ml, nl)]) = pack4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    {
      uint16_t* out = a.out[layer];
      const int q0 = t, q1 = t + 512, q2 = t + 1024;
      const int so0 = (q0 >> 4) * 128 + (((q0 & 15) ^ ((q0 >> 4) & 15)) << 3);
      const int so1 = (q1 >> 4) * 128 + (((q1 & 15) ^ ((q1 >> 4) & 15)) << 3);
      const int so2 = (q2 < 1280) ? (q2 >> 4) * 128 + (((q2 & 15) ^ ((q2 >> 4) & 15)) << 3) : 0;
      const uint4 m0v = *reinterpret_cast<const uint4*>(&stB[so0]);
      const uint4 m1v = *reinterpret_cast<const uint4*>(&stB[so1]);
      const uint4 m2v = *reinterpret_cast<const uint4*>(&stB[so2]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int r0 = m0 + (q0 >> 4), r1 = m0 + (q1 >> 4), r2 = m0 + (q2 >> 4);
      const size_t o0 = (size_t)r0 * N + n0 + (q0 & 15) * 8, o1 = (size_t)r1 * N + n0 + (q1 & 15) * 8, o2 = (size_t)r2 * N + n0 + (q2 & 15) * 8;
      if (r0 < M) *reinterpret_cast<uint4*>(out + o0) = m0v;
      if (r1 < M) *reinterpret_cast<uint4*>(out + o1) = m1v;
      if (q2 < 1280 && r2 < M) *reinterpret_cast<uint4*>(out + o2) = m2v;
    }
    if (layer + 1 < a.L) {
      if (!(a.mode & 8)) ACEZ_VMCNT(0);
      if (l == 0) {
        const uint32_t one = 1;
        asm volatile("global_atomic_add %0, %1, off" ::"v"(a.flags + mt * 32), "v"(one) : "memory");
      }
      if (w < 4) load_w(layer + 1);         // behind the signal: the vmcnt(0) above must cover the stores only
      // the next layer's loaders write stages 0.. while a slow multiplier may still... no: every multiplier passed the "K loop over"
      // barrier before any wave got here, and the staging tile is rewritten only after the next K loop
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4, third variant: the WHOLE weight tile of the next layer in flight before its K loop starts. rowseq requests 64 KiB of the next
// layer's 128 KiB behind the K loop and streams the rest with the In tile through the 4-slot ring -- at the L2's delivered rate (53 MB
// per layer over the chip, 2.6 us) while the chip's L2 -> CU paths sit idle during the epilogue / hand-off phase every layer goes
// through in near lock step. Here the LDS holds EIGHT weight slots (the layer's whole 128 x 512 tile, 128 KiB) and three 80-row In
// slots (30 KiB); the staging tile of the epilogue overlays the two weight slots the K loop frees first, and a layer's stage s lives in
// slot (2 layer + s) & 7, so that at the end of a K loop six stages of the next layer can be requested at once into the six slots
// that are not the staging tile, and the last two once the copy-out has drained it. The K loop then only streams In (80 KiB per layer).
// Wave roles: 0..3 multiply, 4..5 load In, copy out and signal, 6..7 load In and prefetch weights (they never store, so nobody's
// vmcnt(0) in front of the hand-off signal has to drain a weight prefetch).
template <bool SPLIT>   // SPLIT: waves 6, 7 prefetch and never store (first form); else all four loader waves prefetch and all eight waves copy out
__global__ __launch_bounds__(512) void rowseq_fullw_kernel(SeqArgs a) {
  constexpr int WSLOT = 128 * 64, ISLOT = 80 * 64, NI = 3;
  __shared__ __attribute__((aligned(16))) uint16_t smem[8 * WSLOT + NI * ISLOT];
  uint16_t* const sIn = smem + 8 * WSLOT;
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int mtiles = (a.M + 79) / 80;
  const int per_xcd = (mtiles + 7) >> 3;
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + (jx >> 2);
  if (mt >= mtiles) return;
  const int n0 = (jx & 3) * 128, m0 = mt * 80;
  const int M = a.M;
  constexpr int K = 512, KT = 8, N = 512;
  const int fr = l & 15, fq = l >> 4;

  for (int layer = 0; layer < a.L; ++layer) {
    const uint16_t* In = layer ? a.out[layer - 1] : a.In;
    const int base = (2 * layer) & 7;                       // W stage s of this layer: slot (base + s) & 7; staging = slots base, base + 1
    uint16_t* const stB = smem + ((base)&7) * WSLOT;        // 20 KiB of the 32 KiB of two adjacent slots (base is even: never wraps)
    if (w >= 4) {
      const int lw = w - 4;
      const bool prefetcher = SPLIT ? lw >= 2 : true;
      const int NIw = lw < 2 ? 3 : 2;                       // In DMA instructions per stage of this wave (10 groups of 8 rows over 4 waves)
      const uint16_t* gI[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int row = (lw + 4 * j) * 8 + (l >> 3);
        gI[j] = In + (size_t)min(m0 + row, M - 1) * K + ((l & 7) ^ ((row >> 1) & 7)) * 8;
      }
      auto issueI = [&](int kt) {
        uint16_t* slot = sIn + (kt % NI) * ISLOT;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < NIw) __builtin_amdgcn_global_load_lds((gvoid_t*)(gI[j] + kt * 64), (lvoid_t*)(slot + (lw + 4 * j) * 8 * 64), 16, 0, 0);
      };
      // weight stage s of layer ly, this prefetcher's half: 8 instructions of 8 rows each
      auto issueW = [&](int ly, int s) {
        const uint16_t* Wl = a.W + (size_t)ly * 512 * 512;
        uint16_t* slot = smem + ((2 * ly + s) & 7) * WSLOT;
        constexpr int PJ = SPLIT ? 8 : 4;
        const int pw = SPLIT ? lw - 2 : lw;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
          const int row = (pw * PJ + j) * 8 + (l >> 3);
          __builtin_amdgcn_global_load_lds((gvoid_t*)(Wl + (size_t)(n0 + row) * K + ((l & 7) ^ ((row >> 1) & 7)) * 8 + s * 64),
                                           (lvoid_t*)(slot + (pw * PJ + j) * 8 * 64), 16, 0, 0);
        }
      };
      if (layer == 0 && prefetcher) {
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) issueW(0, s2);
      }
      if (layer > 0 && !(a.mode & 4)) {
        const uint32_t target = (a.base + (uint32_t)layer) * (SPLIT ? 24u : 32u);       // 4 workgroups x 6 (8) storing waves per seam
        uint32_t seen;
        do {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(a.flags + mt * 32) : "memory");
          if ((int32_t)(seen - target) < 0) __builtin_amdgcn_s_sleep(1);
        } while ((int32_t)(seen - target) < 0);
      }
      issueI(0); issueI(1); issueI(2);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        // in order: [all W stages of this layer] I0 I1 I2 | I3 | I4 ... : before barrier kt, stage kt must have landed
        const int after = (kt <= 5) ? 2 : (7 - kt);         // In stages requested after stage kt at this point
        if (NIw == 3) ACEZ_VMCNT_DYN(3 * after); else ACEZ_VMCNT_DYN(2 * after);
        __builtin_amdgcn_s_barrier();                       // stage kt has landed; the multipliers are done with stage kt - 1
        if (kt >= 1 && kt + 2 < KT) issueI(kt + 2);
      }
      __builtin_amdgcn_s_barrier();                         // K loop over: every weight slot is free
      if (prefetcher && layer + 1 < a.L) {
#pragma unroll
        for (int s2 = 0; s2 < 6; ++s2) issueW(layer + 1, s2);
      }
      __builtin_amdgcn_s_barrier();                         // output tile complete
    } else {
      f32x4 acc[2][5];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
      float4 bias[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) bias[i] = *reinterpret_cast<const float4*>(a.bias + layer * 512 + n0 + w * 32 + i * 16 + 4 * fq);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint16_t* sW = smem + ((base + kt) & 7) * WSLOT;
        const uint16_t* sI = sIn + (kt % NI) * ISLOT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int c = kk * 4 + fq;
          bf16x8 fa[2], fb[5];
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(&sW[swz(w * 32 + i * 16 + fr, c)]);
#pragma unroll
          for (int j = 0; j < 5; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(&sI[swz(j * 16 + fr, c)]);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // K loop over
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int ml = j * 16 + fr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int nl = w * 32 + i * 16 + 4 * fq;
          float v[4] = {acc[i][j][0] + bias[i].x, acc[i][j][1] + bias[i].y, acc[i][j][2] + bias[i].z, acc[i][j][3] + bias[i].w};
          *reinterpret_cast<uint2*>(&stB[st_off(ml, nl)]) = pack4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // output tile complete
    }
    // copy-out by waves 0..5 (384 lanes, 1280 16-byte chunks of the 80 x 128 tile)
    constexpr int NSTW = SPLIT ? 6 : 8;                     // storing waves
    if (w < NSTW) {
      uint16_t* out = a.out[layer];
      const int tt = w * 64 + l;
      uint4 v[4];
      int q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        q[u] = (SPLIT || u < 3) ? tt + NSTW * 64 * u : 1280;
        const int qq = q[u] < 1280 ? q[u] : 0;
        v[u] = *reinterpret_cast<const uint4*>(&stB[(qq >> 4) * 128 + (((qq & 15) ^ ((qq >> 4) & 15)) << 3)]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = m0 + (q[u] >> 4);
        if (q[u] < 1280 && r < M) *reinterpret_cast<uint4*>(out + (size_t)r * N + n0 + (q[u] & 15) * 8) = v[u];
      }
    }
    __builtin_amdgcn_s_barrier();                           // staging drained: its two weight slots may be refilled
    if (w >= (SPLIT ? 6 : 4) && layer + 1 < a.L) {
      // (the same lambda is not visible here: re-stated) weight stages 6, 7 of the next layer into this layer's staging slots
      constexpr int PJ = SPLIT ? 8 : 4;
      const int pw = SPLIT ? w - 6 : w - 4;
      const uint16_t* Wl = a.W + (size_t)(layer + 1) * 512 * 512;
#pragma unroll
      for (int s2 = 6; s2 < 8; ++s2) {
        uint16_t* slot = smem + ((2 * (layer + 1) + s2) & 7) * WSLOT;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
          const int row = (pw * PJ + j) * 8 + (l >> 3);
          __builtin_amdgcn_global_load_lds((gvoid_t*)(Wl + (size_t)(n0 + row) * K + ((l & 7) ^ ((row >> 1) & 7)) * 8 + s2 * 64),
                                           (lvoid_t*)(slot + (pw * PJ + j) * 8 * 64), 16, 0, 0);
        }
      }
    }
    if (w < NSTW && layer + 1 < a.L) {
      if (!(a.mode & 8)) ACEZ_VMCNT(0);
      if (l == 0) {
        const uint32_t one = 1;
        asm volatile("global_atomic_add %0, %1, off" ::"v"(a.flags + mt * 32), "v"(one) : "memory");
      }
    }
  }
}

static uint16_t f2bf_host(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const int M = 5120, LMAX = 8;
  uint16_t *In, *W; float* bias; uint32_t* flags;
  uint16_t *outA[LMAX], *outB[LMAX];
  CK(hipMalloc(&In, (size_t)M * 512 * 2)); CK(hipMalloc(&W, (size_t)LMAX * 512 * 512 * 2)); CK(hipMalloc(&bias, LMAX * 512 * 4));
  CK(hipMalloc(&flags, 64 * 32 * 4)); CK(hipMemset(flags, 0, 64 * 32 * 4));
  for (int i = 0; i < LMAX; ++i) { CK(hipMalloc(&outA[i], (size_t)M * 512 * 2)); CK(hipMalloc(&outB[i], (size_t)M * 512 * 2)); }
  std::vector<uint16_t> h((size_t)M * 512), hw((size_t)LMAX * 512 * 512);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& x : h) x = f2bf_host(rnd());
  for (auto& x : hw) x = f2bf_host(rnd() * 0.12f);
  std::vector<float> hb(LMAX * 512);
  for (auto& x : hb) x = rnd() * 0.1f + 0.02f;
  CK(hipMemcpy(In, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  // fragment-ordered copy of W for rowseq_wreg_kernel<true>: element (n, k) of a layer with n = nt*128 + w*32 + i*16 + fr, k = s*32 + fq*8 + e
  // at ((nt*4 + w) * 32 + s*2 + i) * 512 + (fq*16 + fr) * 8 + e
  uint16_t* Wf; CK(hipMalloc(&Wf, (size_t)LMAX * 512 * 512 * 2));
  {
    std::vector<uint16_t> hf(hw.size());
    for (int L = 0; L < LMAX; ++L)
      for (int n = 0; n < 512; ++n)
        for (int k = 0; k < 512; ++k) {
          const int nt = n >> 7, w = (n >> 5) & 3, i = (n >> 4) & 1, fr = n & 15, s2 = k >> 5, fq = (k >> 3) & 3, e = k & 7;
          hf[(size_t)L * 262144 + (size_t)((nt * 4 + w) * 32 + s2 * 2 + i) * 512 + (fq * 16 + fr) * 8 + e] = hw[(size_t)L * 262144 + (size_t)n * 512 + k];
        }
    CK(hipMemcpy(Wf, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t base = 0;
  auto run_wreg = [&](int L, bool fragw, int mode) {
    SeqArgs a{};
    a.In = In; a.W = fragw ? Wf : W; a.bias = bias; a.flags = flags; a.base = base; a.M = M; a.L = L; a.mode = mode;
    for (int i = 0; i < LMAX; ++i) a.out[i] = outB[i];
    if (fragw) hipLaunchKernelGGL((rowseq_wreg_kernel<true>), dim3(256), dim3(512), 0, 0, a);
    else hipLaunchKernelGGL((rowseq_wreg_kernel<false>), dim3(256), dim3(512), 0, 0, a);
    base += (uint32_t)(L - 1);
  };
  auto run_seq = [&](int L, int mode) {
    SeqArgs a{};
    a.In = In; a.W = W; a.bias = bias; a.flags = flags; a.base = base; a.M = M; a.L = L; a.mode = mode;
    for (int i = 0; i < LMAX; ++i) a.out[i] = outB[i];
    // mode bit 4 (16): five ring slots; bit 5 (32): stage barriers one stage ahead + fragment double buffering
    if ((a.mode & 48) == 48) hipLaunchKernelGGL((rowseq_kernel<5, true>), dim3(256), dim3(512), 0, 0, a);
    else if (a.mode & 32) hipLaunchKernelGGL((rowseq_kernel<4, true>), dim3(256), dim3(512), 0, 0, a);
    else if (a.mode & 16) hipLaunchKernelGGL((rowseq_kernel<5, false>), dim3(256), dim3(512), 0, 0, a);
    else hipLaunchKernelGGL((rowseq_kernel<4, false>), dim3(256), dim3(512), 0, 0, a);
    base += (uint32_t)(L - 1);
  };
  auto run_ref = [&](int L) {
    for (int i = 0; i < L; ++i) {
      RowGemmArgs g{};
      g.In = i ? outA[i - 1] : In; g.W = W + (size_t)i * 512 * 512; g.bias = bias + i * 512; g.out_main = outA[i]; g.M = M; g.N = 512; g.K = 512;
      g.relu = 1; g.aux_mode = AUX_NONE;
      launch_rowgemm(g, 0);
    }
  };
  // bit equality of the last layer's output
  std::vector<uint16_t> ra((size_t)M * 512), rb((size_t)M * 512);
  for (int L : {2, 4, 8}) {
    for (int mode : {0, 1, 2, 16, 32, 48, 64}) {
      for (int i = 0; i < LMAX; ++i) CK(hipMemset(outB[i], 0xff, (size_t)M * 512 * 2));
      run_ref(L); run_seq(L, mode);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(ra.data(), outA[L - 1], ra.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), outB[L - 1], rb.size() * 2, hipMemcpyDeviceToHost));
      size_t bad = 0, nz = 0;
      for (size_t i = 0; i < ra.size(); ++i) { bad += ra[i] != rb[i]; nz += ra[i] != 0; }
      printf("L=%d mode=%d: %zu mismatching of %zu (non-zero %zu)\n", L, mode, bad, ra.size(), nz);
    }
  }
  uint32_t* flags2; CK(hipMalloc(&flags2, 64 * 32 * 4)); CK(hipMemset(flags2, 0, 64 * 32 * 4));
  uint32_t base2 = 0;
  uint32_t* flags3; CK(hipMalloc(&flags3, 64 * 32 * 4)); CK(hipMemset(flags3, 0, 64 * 32 * 4));
  uint32_t base3 = 0;
  int fullw_split = 0;
  auto run_fullw = [&](int L, int mode) {
    SeqArgs a{};
    a.In = In; a.W = W; a.bias = bias; a.M = M; a.L = L; a.mode = mode;
    for (int i = 0; i < LMAX; ++i) a.out[i] = outB[i];
    if (fullw_split) { a.flags = flags2; a.base = base2; hipLaunchKernelGGL(rowseq_fullw_kernel<true>, dim3(256), dim3(512), 0, 0, a); base2 += (uint32_t)(L - 1); }
    else { a.flags = flags3; a.base = base3; hipLaunchKernelGGL(rowseq_fullw_kernel<false>, dim3(256), dim3(512), 0, 0, a); base3 += (uint32_t)(L - 1); }
  };
  for (fullw_split = 0; fullw_split < 2; ++fullw_split)
  for (int L : {1, 2, 8}) {
    for (int i = 0; i < LMAX; ++i) CK(hipMemset(outB[i], 0xff, (size_t)M * 512 * 2));
    run_ref(L); run_fullw(L, 0);
    CK(hipDeviceSynchronize());
    size_t bad = 0;
    for (int li = 0; li < L; ++li) {
      CK(hipMemcpy(ra.data(), outA[li], ra.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), outB[li], rb.size() * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ra.size(); ++i) bad += ra[i] != rb[i];
    }
    printf("full-W prefetch (%s) L=%d: %zu mismatching of %zu (all layers)\n", fullw_split ? "2 prefetch waves" : "4 prefetch waves", L, bad, ra.size() * L);
  }
  for (int rep = 0; rep < 2; ++rep)
    for (fullw_split = 0; fullw_split < 2; ++fullw_split)
    for (int L : {1, 2, 8})
      for (int mode : {0, 12}) {
        float ms; const int n = 200;
        for (int i = 0; i < 10; ++i) run_fullw(L, mode);
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_fullw(L, mode); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("L=%d  full-W prefetch (%s) mode %2d%s: %7.2f us (%.2f per layer)\n", L, fullw_split ? "2 prefetch waves" : "4 prefetch waves", mode, mode ? " (timing only)" : "              ", ms * 1e3 / n, ms * 1e3 / n / L);
      }
  for (int L : {2, 8})
    for (int fragw = 0; fragw < 2; ++fragw) {
      for (int i = 0; i < LMAX; ++i) CK(hipMemset(outB[i], 0xff, (size_t)M * 512 * 2));
      run_ref(L); run_wreg(L, fragw != 0, 0);
      CK(hipDeviceSynchronize());
      size_t bad = 0;
      for (int li = 0; li < L; ++li) {
        CK(hipMemcpy(ra.data(), outA[li], ra.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), outB[li], rb.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ra.size(); ++i) bad += ra[i] != rb[i];
      }
      printf("W in registers L=%d %s: %zu mismatching of %zu (all layers)\n", L, fragw ? "fragment-ordered W" : "row-major W", bad, ra.size() * L);
    }
  for (int rep = 0; rep < 2; ++rep)
    for (int L : {1, 2, 4, 8}) {
      for (int fragw = 0; fragw < 2; ++fragw)
        for (int mode : {0, 4, 12}) {
          float ms; const int n = 200;
          for (int i = 0; i < 10; ++i) run_wreg(L, fragw != 0, mode);
          CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_wreg(L, fragw != 0, mode); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep) printf("L=%d  W in registers (%s) mode %2d%s: %7.2f us (%.2f per layer)\n", L, fragw ? "fragment-ordered" : "row-major       ", mode,
                          mode ? " (timing only)" : "              ", ms * 1e3 / n, ms * 1e3 / n / L);
        }
      float ms_ref, ms0, ms1, ms2;
      const int n = 200;
      for (int i = 0; i < 10; ++i) run_ref(L);
      CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_ref(L); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1));
      for (int i = 0; i < 10; ++i) run_seq(L, 0);
      CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_seq(L, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms0, e0, e1));
      for (int i = 0; i < 10; ++i) run_seq(L, 1);
      CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_seq(L, 1); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
      for (int i = 0; i < 10; ++i) run_seq(L, 2);
      CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_seq(L, 2); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms2, e0, e1));
      if (rep) {
        for (int mode : {64, 16, 32, 48, 4, 8, 12, 28, 44, 76}) {   // 16 / 32 / 48: five slots / one stage ahead / both (valid results); timing only: no consumer wait / no store-acknowledgement wait / neither
          float ms;
          for (int i = 0; i < 10; ++i) run_seq(L, mode);
          CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_seq(L, mode); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
          printf("L=%d  one launch, timing only, mode %2d (4: consumers do not wait, 8: producers do not wait for store acknowledgements): %7.2f us (%.2f per layer)\n", L, mode, ms * 1e3 / n, ms * 1e3 / n / L);
        }
      }
      if (rep) printf("L=%d  per-layer launches %7.2f us | one launch: no fences %7.2f us, acquire fence only %7.2f us, release+acquire fences %7.2f us   (per layer %.2f | %.2f / %.2f / %.2f)\n",
                      L, ms_ref * 1e3 / n, ms0 * 1e3 / n, ms2 * 1e3 / n, ms1 * 1e3 / n, ms_ref * 1e3 / n / L, ms0 * 1e3 / n / L, ms2 * 1e3 / n / L, ms1 * 1e3 / n / L);
    }
  return 0;
}
