// split_proto.hip -- VERDICT r4 item 2: "two independent half-tile pipelines per CU", measured in its zero-scheduling form first:
// TWO co-resident workgroups per CU (<= 80 KiB of LDS and <= 128 VGPRs each), each a complete rowseq pipeline on a smaller tile, so
// that one workgroup's exposed In latency, epilogue and hand-off elapse under the other's K loop -- the hardware interleaves them.
// Timing + bit-equality experiment, not part of the library (the product's 80 x 128 rowseq_kernel is the baseline in the same process).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I acezero_amd/csrc tools/split_proto.hip -o /tmp/split && /tmp/split
//
// rowsplit_kernel<TR, NC, R>: tile = TR rows x NC columns, 3-slot ring of [W NC x 64 | In IR x 64] stages (IR = TR rounded up to 16),
// one [TR][NC] staging tile; 4 multiplier + 4 loader waves as in rowgemm80; same MFMA, same K order per output element => the outputs
// are bit-identical to the per-layer launches. Two shapes:
//   A  40 x 128 (row halves):    168 KiB per workgroup and layer through the L2 -> LDS path, 336 KiB per CU (product: 208 KiB per CU);
//                                every 40-row half is its own pipeline of 4 sibling workgroups (128 counters)
//   B  80 x  64 (column halves): 144 KiB per workgroup and layer, 288 KiB per CU; a row tile has 8 siblings
// Forward chain only (bias + ReLU: the cheapest epilogue of the step), L layers in one launch, same-XCD hand-off as rowseq_kernel.
#define main seam_proto_main
#include "seam_proto.hip"
#undef main

template <int NC>
__device__ __forceinline__ int st_offn(int row, int col) { return row * NC + ((((col >> 3) ^ (row & (NC / 8 - 1))) << 3) | (col & 7)); }

template <int TR, int NC, int R>
__global__ __launch_bounds__(512, 2) void rowsplit_kernel(SeqArgs a) {
  constexpr int RF = (TR + 15) / 16, IR = RF * 16, CF = NC / 64;
  constexpr int IG = IR / 8;             // 8-row DMA groups of the In part of a stage
  constexpr int NWI = NC / 32;           // W DMA instructions per loader wave and stage
  constexpr int NTILES = 512 / NC;
  constexpr int STAGE = (NC + IR) * 64;
  __shared__ __attribute__((aligned(16))) uint16_t smem[R * STAGE + TR * NC];
  uint16_t* const stB = smem + R * STAGE;
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int mtiles = (a.M + TR - 1) / TR;
  const int per_xcd = (mtiles + 7) >> 3;
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + jx / NTILES;
  if (mt >= mtiles) return;
  const int n0 = (jx % NTILES) * NC, m0 = mt * TR;
  const int M = a.M;
  constexpr int K = 512, KT = 8, N = 512;

  for (int layer = 0; layer < a.L; ++layer) {
    const uint16_t* In = layer ? a.out[layer - 1] : a.In;
    const uint16_t* Wl = a.W + (size_t)layer * 512 * 512;
    if (w >= 4) {
      const int lw = w - 4;
      const uint16_t* gW[NWI];
      const uint16_t* gI[3];
      const int NI = (IG - lw + 3) / 4;   // In groups lw, lw + 4, lw + 8 below IG
#pragma unroll
      for (int j = 0; j < NWI; ++j) {
        const int row = (lw + 4 * j) * 8 + (l >> 3);
        gW[j] = Wl + (size_t)(n0 + row) * K + ((l & 7) ^ ((row >> 1) & 7)) * 8;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int row = (lw + 4 * j) * 8 + (l >> 3);
        gI[j] = In + (size_t)min(m0 + row, M - 1) * K + ((l & 7) ^ ((row >> 1) & 7)) * 8;
      }
      auto issueW = [&](int kt) {
        uint16_t* slot = smem + (kt % R) * STAGE;
#pragma unroll
        for (int j = 0; j < NWI; ++j)
          __builtin_amdgcn_global_load_lds((gvoid_t*)(gW[j] + kt * 64), (lvoid_t*)(slot + (lw + 4 * j) * 8 * 64), 16, 0, 0);
      };
      auto issueI = [&](int kt) {
        uint16_t* slot = smem + (kt % R) * STAGE;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < NI) __builtin_amdgcn_global_load_lds((gvoid_t*)(gI[j] + kt * 64), (lvoid_t*)(slot + NC * 64 + (lw + 4 * j) * 8 * 64), 16, 0, 0);
      };
      if (layer == 0) {
#pragma unroll
        for (int k = 0; k < R; ++k) issueW(k);
      }
      if (layer > 0 && !(a.mode & 4)) {
        const uint32_t target = (a.base + (uint32_t)layer) * (uint32_t)(NTILES * 8);   // NTILES workgroups x 8 waves per seam
        uint32_t seen;
        do {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(a.flags + mt * 32) : "memory");
          if ((int32_t)(seen - target) < 0) __builtin_amdgcn_s_sleep(1);
        } while ((int32_t)(seen - target) < 0);
      }
#pragma unroll
      for (int k = 0; k < R; ++k) issueI(k);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int init_after = (kt < R) ? NI * (R - 1 - kt) : 0;
        const int last_refill = (kt - 1 + R - 1 < KT - 1) ? kt - 1 + R - 1 : KT - 1;   // highest stage requested before this wait
        const int refills = (last_refill >= R && last_refill > kt) ? (last_refill - (kt > R - 1 ? kt : R - 1)) : 0;
        wait_vmcnt_dyn(init_after + (NWI + NI) * refills);
        __builtin_amdgcn_s_barrier();
        if (kt >= 1 && kt + R - 1 < KT) { issueW(kt + R - 1); issueI(kt + R - 1); }
      }
      __builtin_amdgcn_s_barrier();       // K loop over: the ring is free
      if (layer + 1 < a.L) {
#pragma unroll
        for (int j = 0; j < NWI; ++j) gW[j] += 512 * 512;
#pragma unroll
        for (int k = 0; k < R; ++k) issueW(k);
      }
      __builtin_amdgcn_s_barrier();       // output tile complete
    } else {
      f32x4 acc[CF][RF];
#pragma unroll
      for (int i = 0; i < CF; ++i)
#pragma unroll
        for (int j = 0; j < RF; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
      const int fr = l & 15, fq = l >> 4;
      float4 bias[CF];
#pragma unroll
      for (int i = 0; i < CF; ++i) bias[i] = *reinterpret_cast<const float4*>(a.bias + layer * 512 + n0 + w * (NC / 4) + i * 16 + 4 * fq);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint16_t* sW = smem + (kt % R) * STAGE;
        const uint16_t* sI = sW + NC * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int c = kk * 4 + fq;
          bf16x8 fa[CF], fb[RF];
#pragma unroll
          for (int i = 0; i < CF; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(&sW[swz(w * (NC / 4) + i * 16 + fr, c)]);
#pragma unroll
          for (int j = 0; j < RF; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(&sI[swz(j * 16 + fr, c)]);
#pragma unroll
          for (int i = 0; i < CF; ++i)
#pragma unroll
            for (int j = 0; j < RF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < RF; ++j) {
        const int ml = j * 16 + fr;
#pragma unroll
        for (int i = 0; i < CF; ++i) {
          const int nl = w * (NC / 4) + i * 16 + 4 * fq;
          float v[4] = {acc[i][j][0] + bias[i].x, acc[i][j][1] + bias[i].y, acc[i][j][2] + bias[i].z, acc[i][j][3] + bias[i].w};
          if (ml < TR) *reinterpret_cast<uint2*>(&stB[st_offn<NC>(ml, nl)]) = pack4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    {
      uint16_t* out = a.out[layer];
      constexpr int CPR = NC / 8;               // 16-byte chunks per tile row
      constexpr int CH = TR * CPR;              // chunks of the tile (640 for both shapes)
      static_assert(CH <= 1024, "two chunks per thread");
      const int q0 = t, q1 = t + 512;
      const int rr0 = q0 / CPR, c0 = q0 % CPR, rr1 = (q1 < CH) ? q1 / CPR : 0, c1 = q1 % CPR;
      const uint4 m0v = *reinterpret_cast<const uint4*>(&stB[rr0 * NC + ((c0 ^ (rr0 & (CPR - 1))) << 3)]);
      const uint4 m1v = *reinterpret_cast<const uint4*>(&stB[rr1 * NC + ((c1 ^ (rr1 & (CPR - 1))) << 3)]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int r0 = m0 + rr0, r1 = m0 + rr1;
      if (q0 < CH && r0 < M) *reinterpret_cast<uint4*>(out + (size_t)r0 * N + n0 + c0 * 8) = m0v;
      if (q1 < CH && r1 < M) *reinterpret_cast<uint4*>(out + (size_t)r1 * N + n0 + c1 * 8) = m1v;
    }
    if (layer + 1 < a.L) {
      if (!(a.mode & 8)) ACEZ_VMCNT(0);  // stores acknowledged by this XCD's L2 = visible to the sibling workgroups
      if (l == 0) {
        const uint32_t one = 1;
        asm volatile("global_atomic_add %0, %1, off" ::"v"(a.flags + mt * 32), "v"(one) : "memory");
      }
    }
  }
}

template <class KERN>
static int occupancy(KERN k, const char* name) {
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 512, 0);
  hipFuncAttributes fa{};
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k));
  printf("%s: %d workgroups per CU (%s), %d VGPRs, %zu B LDS\n", name, nb, hipGetErrorString(e), fa.numRegs, fa.sharedSizeBytes);
  return nb;
}

int main() {
  const int M = 5120, LMAX = 8;
  uint16_t *In, *W; float* bias;
  uint16_t *outA[LMAX], *outB[LMAX];
  CK(hipMalloc(&In, (size_t)M * 512 * 2)); CK(hipMalloc(&W, (size_t)LMAX * 512 * 512 * 2)); CK(hipMalloc(&bias, LMAX * 512 * 4));
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
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // one counter array + base per kernel shape (the counters are monotonic per row tile)
  uint32_t* flags[3]; uint32_t base[3] = {0, 0, 0};
  for (int i = 0; i < 3; ++i) { CK(hipMalloc(&flags[i], 128 * 32 * 4)); CK(hipMemset(flags[i], 0, 128 * 32 * 4)); }
  occupancy(rowseq_kernel<4, false>, "rowseq 80x128 (product shape)");
  const int occA = occupancy(rowsplit_kernel<40, 128, 3>, "rowsplit A 40x128");
  const int occB = occupancy(rowsplit_kernel<80, 64, 3>, "rowsplit B 80x64");
  if (occA < 2 || occB < 2) printf("WARNING: a split shape is not co-resident twice per CU: its workgroups would deadlock waiting for siblings; skipped\n");
  auto run = [&](int shape, int L, int mode) {
    SeqArgs a{};
    a.In = In; a.W = W; a.bias = bias; a.flags = flags[shape]; a.base = base[shape]; a.M = M; a.L = L; a.mode = mode;
    for (int i = 0; i < LMAX; ++i) a.out[i] = outB[i];
    if (shape == 0) hipLaunchKernelGGL((rowseq_kernel<4, false>), dim3(256), dim3(512), 0, 0, a);
    else if (shape == 1) hipLaunchKernelGGL((rowsplit_kernel<40, 128, 3>), dim3(512), dim3(512), 0, 0, a);
    else hipLaunchKernelGGL((rowsplit_kernel<80, 64, 3>), dim3(512), dim3(512), 0, 0, a);
    base[shape] += (uint32_t)(L - 1);
  };
  auto run_ref = [&](int L) {
    for (int i = 0; i < L; ++i) {
      RowGemmArgs g{};
      g.In = i ? outA[i - 1] : In; g.W = W + (size_t)i * 512 * 512; g.bias = bias + i * 512; g.out_main = outA[i]; g.M = M; g.N = 512; g.K = 512;
      g.relu = 1; g.aux_mode = AUX_NONE;
      launch_rowgemm(g, 0);
    }
  };
  const char* names[3] = {"rowseq 80x128, 1 per CU", "split A 40x128, 2 per CU", "split B 80x64,  2 per CU"};
  std::vector<uint16_t> ra((size_t)M * 512), rb((size_t)M * 512);
  for (int shape = 0; shape < 3; ++shape) {
    if ((shape == 1 && occA < 2) || (shape == 2 && occB < 2)) continue;
    for (int L : {1, 2, 8}) {
      for (int i = 0; i < LMAX; ++i) CK(hipMemset(outB[i], 0xff, (size_t)M * 512 * 2));
      run_ref(L); run(shape, L, 0);
      CK(hipDeviceSynchronize());
      size_t bad = 0;
      for (int li = 0; li < L; ++li) {
        CK(hipMemcpy(ra.data(), outA[li], ra.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), outB[li], rb.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ra.size(); ++i) bad += ra[i] != rb[i];
      }
      printf("%s L=%d: %zu mismatching of %zu (all layers)\n", names[shape], L, bad, ra.size() * L);
    }
  }
  for (int rep = 0; rep < 3; ++rep)
    for (int L : {1, 2, 8}) {
      float ms; const int n = 200;
      for (int i = 0; i < 10; ++i) run_ref(L);
      CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_ref(L); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("L=%d  per-layer launches             : %7.2f us (%.2f per layer)\n", L, ms * 1e3 / n, ms * 1e3 / n / L);
      for (int shape = 0; shape < 3; ++shape) {
        if ((shape == 1 && occA < 2) || (shape == 2 && occB < 2)) continue;
        for (int mode : {0, 4, 12}) {
          for (int i = 0; i < 10; ++i) run(shape, L, mode);
          CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run(shape, L, mode); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep) printf("L=%d  %s mode %2d%s: %7.2f us (%.2f per layer)\n", L, names[shape], mode, mode ? " (timing only)" : "              ", ms * 1e3 / n, ms * 1e3 / n / L);
        }
      }
    }
  return 0;
}
