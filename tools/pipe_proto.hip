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
  uint16_t* out[8];         // layer outputs
  uint32_t* prog;           // [8 groups][8 layers][32 words]: word 4 c + w = micro-batches stored by multiplier wave w of quarter c (monotonic)
  uint32_t base;            // value of every word before this launch
  uint32_t limit;           // poll budget
  uint32_t* fault;
  int M, L, mode;           // mode bit 0: consumers do not wait (timing only: the floor without the hand-off latency)
};

__device__ __forceinline__ uint32_t row16_umin(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));
  return v;
}

template <int MB, int SIG>
__global__ __launch_bounds__(512) void pipe_fwd_kernel(PipeArgs a) {
  constexpr int RF = MB / 16;               // row fragments per micro-batch
  constexpr int SLOT = MB * 512;            // elements per ring slot
  constexpr int NS = 128 / MB;              // ring slots: 128 KiB
  constexpr int IPM = MB / 4;               // DMA instructions per loader wave and micro-batch (one 1 KiB row each)
  __shared__ __attribute__((aligned(16))) uint16_t smem[NS * SLOT];
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
  uint32_t* const mine = a.prog + (g * 8 + layer) * 32;

  if (w >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - 4;
    const uint16_t* In = layer ? a.out[layer - 1] : a.In0;
    const uint32_t* up = a.prog + (g * 8 + (layer > 0 ? layer - 1 : 0)) * 32 + (l & 15);
    int issued = 0;
    int ready = (layer == 0 || (a.mode & 1)) ? nmb : 0;
    uint32_t spins = 0;
    // DMA instruction j of this wave covers slot row r = lw * IPM + j (1 KiB); lane l writes physical chunk l, which must hold
    // the logical chunk l ^ (r & 15)
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
    for (int i = 0; i < nmb; ++i) {
      const int cap = min(nmb, i + NS - 1);   // slots of micro-batches <= i - 2 have been read (barrier A_{i-1} is behind us)
      while (issued < cap) {
        if (ready <= issued) {
          uint32_t v;
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(up) : "memory");
          v = row16_umin(v);
          ready = (int)(__builtin_amdgcn_readfirstlane(v) - a.base);
          if (ready <= issued) {
            if (issued > i) break;            // run-ahead only: do not hold up micro-batch i
            if (++spins > a.limit) { if (l == 0) *a.fault = 1; ready = nmb; } else __builtin_amdgcn_s_sleep(1);
            continue;
          }
        }
        issue(issued);
        ++issued;
      }
      const int ahead = issued - 1 - i;       // micro-batches requested after i: may stay in flight
      switch (ahead) {
        case 0: ACEZ_VMCNT_C(0); break;
        case 1: ACEZ_VMCNT_C(IPM); break;
        case 2: ACEZ_VMCNT_C(2 * IPM); break;
        case 3: ACEZ_VMCNT_C(3 * IPM); break;
        case 4: ACEZ_VMCNT_C(4 * IPM); break;
        case 5: ACEZ_VMCNT_C(5 * IPM); break;
        default: ACEZ_VMCNT_C(6 * IPM); break;
      }
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
    uint16_t* const out = a.out[layer];
    const int coff = ncol + (fq & 1) * 16 + (fq >> 1) * 8;   // after the permlane16 swaps this lane holds 8 consecutive columns
    uint32_t* const myword = mine + c * 4 + w;
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
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        E::frag fb[RF];
#pragma unroll
        for (int j = 0; j < RF; ++j) fb[j] = *reinterpret_cast<const E::frag*>(&slot[(j * 16 + fr) * 512 + (((4 * s + fq) ^ fr) << 3)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < RF; ++j) acc[i][j] = E::mfma16(fa[i][s], fb[j], acc[i][j]);
        if (s == SIG && mb > 0) {             // the stores of the micro-batch before: acknowledged by this XCD's L2 -> visible to the next layer
          ACEZ_VMCNT(0);
          if (l == 0) { const uint32_t one = 1; asm volatile("global_atomic_add %0, %1, off" ::"v"(myword), "v"(one) : "memory"); }
        }
      }
      const int m0 = (mb0 + mb) * MB;
#pragma unroll
      for (int j = 0; j < RF; ++j) {
        uint2 y[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float b[4] = {bias[i].x, bias[i].y, bias[i].z, bias[i].w};
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[i][j][r] + b[r], 0.f);
          y[i] = E::pk4(v[0], v[1], v[2], v[3]);
        }
        // odd 16-lane rows of y[0] <-> even rows of y[1]: every lane ends with 16 contiguous bytes of one output row
        auto rx = __builtin_amdgcn_permlane16_swap(y[0].x, y[1].x, false, false);
        auto ry = __builtin_amdgcn_permlane16_swap(y[0].y, y[1].y, false, false);
        const int row = m0 + j * 16 + fr;
        if (row < M) *reinterpret_cast<uint4*>(out + (size_t)row * 512 + coff) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
      }
    }
    if (nmb > 0) {
      ACEZ_VMCNT(0);
      if (l == 0) { const uint32_t one = 1; asm volatile("global_atomic_add %0, %1, off" ::"v"(myword), "v"(one) : "memory"); }
    }
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
  setvbuf(stdout, nullptr, _IONBF, 0);
  uint16_t *In, *W; float* bias; uint32_t *prog, *fault;
  uint16_t *outA[LMAX], *outB[LMAX];
  CK(hipMalloc(&In, (size_t)M * 512 * 2)); CK(hipMalloc(&W, (size_t)LMAX * 512 * 512 * 2)); CK(hipMalloc(&bias, LMAX * 512 * 4));
  CK(hipMalloc(&prog, 8 * 8 * 32 * 4)); CK(hipMemset(prog, 0, 8 * 8 * 32 * 4));
  CK(hipMalloc(&fault, 4)); CK(hipMemset(fault, 0, 4));
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
  uint32_t base16 = 0, base32 = 0;
  uint32_t* prog32 = prog;
  uint32_t* prog16; CK(hipMalloc(&prog16, 8 * 8 * 32 * 4)); CK(hipMemset(prog16, 0, 8 * 8 * 32 * 4));
  // variant: 0 = MB 32 signal at K-step 6, 1 = MB 16 signal at 6, 2 = MB 32 signal at 12, 3 = MB 16 signal at 12
  auto run_pipe = [&](int L, int variant, int mode) {
    PipeArgs a{};
    a.In0 = In; a.W = W; a.bias = bias; a.M = M; a.L = L; a.mode = mode; a.limit = 20000; a.fault = fault;
    for (int i = 0; i < LMAX; ++i) a.out[i] = outB[i];
    const bool mb16 = variant & 1;
    a.prog = mb16 ? prog16 : prog32; a.base = mb16 ? base16 : base32;
    const int MB = mb16 ? 16 : 32;
    const int per_g = (((M + MB - 1) / MB) + 7) / 8;
    if (variant == 0) hipLaunchKernelGGL((pipe_fwd_kernel<32, 6>), dim3(32 * L), dim3(512), 0, 0, a);
    else if (variant == 1) hipLaunchKernelGGL((pipe_fwd_kernel<16, 6>), dim3(32 * L), dim3(512), 0, 0, a);
    else if (variant == 2) hipLaunchKernelGGL((pipe_fwd_kernel<32, 12>), dim3(32 * L), dim3(512), 0, 0, a);
    else hipLaunchKernelGGL((pipe_fwd_kernel<16, 12>), dim3(32 * L), dim3(512), 0, 0, a);
    (mb16 ? base16 : base32) += (uint32_t)per_g;
  };
  // the progress words of a layer only advance in launches that run that layer: a change of L restarts the count
  auto reset = [&]() {
    (void)hipDeviceSynchronize();
    (void)hipMemset(prog32, 0, 8 * 8 * 32 * 4); (void)hipMemset(prog16, 0, 8 * 8 * 32 * 4);
    base16 = base32 = 0;
  };
  auto run_ref = [&](int L) {
    for (int i = 0; i < L; ++i) {
      RowGemmArgs g{};
      g.In = i ? outA[i - 1] : In; g.W = W + (size_t)i * 512 * 512; g.bias = bias + i * 512; g.out_main = outA[i]; g.M = M; g.N = 512; g.K = 512;
      g.relu = 1; g.aux_mode = AUX_NONE;
      launch_rowgemm(g, 80, 0);
    }
  };
  if (M % 256) { printf("this experiment wants M %% 256 == 0 (uniform micro-batch count per XCD)\n"); return 1; }
  // bit equality of every layer's output
  std::vector<uint16_t> ra((size_t)M * 512), rb((size_t)M * 512);
  for (int L : {1, 2, 8}) {
    reset();
    for (int variant : {0, 1, 2, 3}) {
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
    for (int L : {1, 2, 4, 8}) {
      reset();
      const int n = 200;
      float ms_ref;
      for (int i = 0; i < 10; ++i) run_ref(L);
      CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_ref(L); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1));
      if (rep) printf("L=%d  per-layer launches %7.2f us (%.2f per layer)\n", L, ms_ref * 1e3 / n, ms_ref * 1e3 / n / L);
      for (int variant : {0, 1, 2, 3})
        for (int mode : {0, 1}) {
          float ms;
          for (int i = 0; i < 10; ++i) run_pipe(L, variant, mode);
          CK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) run_pipe(L, variant, mode); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
          { uint32_t ff = 0; CK(hipMemcpy(&ff, fault, 4, hipMemcpyDeviceToHost)); if (ff) { printf("FAULT (poll budget expired) at L=%d variant=%d mode=%d\n", L, variant, mode); return 2; } }
          if (rep) printf("L=%d  pipeline MB=%d signal@%d %s: %7.2f us (%.2f per layer)\n", L, (variant & 1) ? 16 : 32, (variant & 2) ? 12 : 6,
                          mode ? "NO WAIT (timing only)" : "                     ", ms * 1e3 / n, ms * 1e3 / n / L);
        }
    }
  uint32_t f = 0; CK(hipMemcpy(&f, fault, 4, hipMemcpyDeviceToHost));
  printf("fault word at exit: %u\n", f);
  return 0;
}
