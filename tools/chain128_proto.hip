// chain128_proto.hip -- timing prototype (round 6): the head's eight pointwise 512 -> 512 layers on whole frames as ONE launch with a
// 128-row activation tile resident in LDS (updated in place: the accumulators hold a layer's output until every wave has read its input) and
// the weights pulled straight into registers in MFMA operand order (fragment-ordered copy of W: one wave-instruction = 1 KiB contiguous).
// Round 5's head_infer kept 64 rows + an 80 KiB weight ring (18 us per layer and tile, slower than the per-layer launches); here the LDS
// holds 128 rows and nothing else, so a weight byte is used twice as often and the ring's "bytes in flight" are registers.
//   hipcc --offload-arch=gfx950 -O3 -o tools/chain128_proto.bin tools/chain128_proto.hip && tools/chain128_proto.bin [rows] [PF]
// Prints us per pass and per layer-tile; checks the result of a small case against a host reference (bf16 inputs, fp32 accumulation).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) const bf16x8 lds_frag;
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

static inline uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float h_bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

#ifndef PF
#define PF 4
#endif
constexpr int L = 8;
// timing-only ablations: ABL & 1 = every weight fragment from the same 1 KiB (L1 hits: no weight stream), ABL & 2 = B fragments read once per
// layer (no LDS stream), ABL & 4 = every layer reads layer 0's weights (512 KiB working set instead of 4 MiB)
#ifndef ABL
#define ABL 0
#endif
constexpr int WSTEP = (ABL & 1) ? 0 : 512;
// ROT: the K steps of a layer start at a workgroup-dependent step (the 32 CUs of an XCD read 32 different weight lines at any moment instead of one)
#ifndef ROT
#define ROT 0
#endif

// Wp: [layer][co block of 32 (16)][K step of 16 (32)][lane (64)][8]: lane (fr = l & 31, fh = l >> 5) = W[cb*32 + fr][kk*16 + fh*8 ..]
__global__ __launch_bounds__(512) void chain128_kernel(const uint16_t* __restrict__ In, const uint16_t* __restrict__ Wp, const float* __restrict__ bias,
                                                       uint16_t* __restrict__ Out, int M, int layers) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[128 * 512];
  lds_byte* const lds = (lds_byte*)tile;
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = l & 31, fh = l >> 5;
  const int ntiles = (M + 127) >> 7;
  unsigned baddr[4], bx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned r = j * 32 + fr;
    baddr[j] = r * 1024;
    bx[j] = (r & 31) ^ (unsigned)fh;
  }
  const int rot = ROT ? (int)((blockIdx.x >> 3) & 31) : 0;
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int m0 = tl << 7;
    // ---- input tile -> LDS (LDS-DMA: wave w rows 16 w .. 16 w + 15, 16 instructions of one row each; lane = 16-byte chunk, swizzled)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = w * 16 + i;
      const int g = min(m0 + row, M - 1);
      __builtin_amdgcn_global_load_lds((gvoid_t*)(In + (size_t)g * 512 + ((l ^ (row & 31)) << 3)), (lvoid_t*)(lds + row * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int layer = 0; layer < layers; ++layer) {
      f32x16 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      const uint16_t* wp[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wp[i] = Wp + ((size_t)(((ABL & 4) ? 0 : layer) * 16 + w * 2 + i) * 32) * 512 + l * 8;
      bf16x8 wa[PF][2];
#pragma unroll
      for (int k = 0; k < PF; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) wa[k][i] = *reinterpret_cast<const bf16x8*>(wp[i] + WSTEP * ((k + rot) & 31));
      bf16x8 fb[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[0][j] = *(lds_frag*)(lds + baddr[j] + ((((unsigned)(2 * rot)) ^ bx[j]) << 4));
      // K loop in groups of PF steps (PF even): the loop is NOT unrolled beyond a group, step addresses are computed as they are needed
      // (the fully unrolled 32-step form kept ~190 address registers alive: 94 dwords of spills)
#pragma clang loop unroll(disable)
      for (int k0 = 0; k0 < 32; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int kk = k0 + u;
          if (!(ABL & 2)) {
            const unsigned c2 = 2u * (unsigned)((kk + 1 + rot) & 31);     // (the read past the last step wraps to step `rot`: harmless)
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[(u + 1) & 1][j] = *(lds_frag*)(lds + baddr[j] + ((c2 ^ bx[j]) << 4));
          }
          bf16x8 cur[2] = {wa[u][0], wa[u][1]};
          {
            const int kn = (kk + PF + rot) & 31;                          // (past the end: re-reads early steps, results unused)
#pragma unroll
            for (int i = 0; i < 2; ++i) wa[u][i] = *reinterpret_cast<const bf16x8*>(wp[i] + WSTEP * kn);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[i], fb[(ABL & 2) ? 0 : (u & 1)][j], acc[i][j], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();             // every wave has read the layer's input
      float4 bv[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const float4*>(bias + layer * 512 + w * 64 + i * 32 + 8 * q + 4 * fh);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ml = j * 32 + fr;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = w * 64 + i * 32 + 8 * q + 4 * fh;
            const float4 b = bv[i][q];
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            bf2 lo, hi;
            lo[0] = (__bf16)fmaxf(acc[i][j][4 * q + 0] + b.x, 0.f); lo[1] = (__bf16)fmaxf(acc[i][j][4 * q + 1] + b.y, 0.f);
            hi[0] = (__bf16)fmaxf(acc[i][j][4 * q + 2] + b.z, 0.f); hi[1] = (__bf16)fmaxf(acc[i][j][4 * q + 3] + b.w, 0.f);
            uint2 pk = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
            *reinterpret_cast<uint2*>(&tile[ml * 512 + ((((nl >> 3) ^ (ml & 31)) << 3) | (nl & 7))]) = pk;
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    for (int q = t; q < 128 * 64; q += 512) {
      const int row = q >> 6, ch = q & 63, m = m0 + row;
      if (m < M) *reinterpret_cast<uint4*>(Out + (size_t)m * 512 + ch * 8) = *reinterpret_cast<const uint4*>(&tile[row * 512 + ((ch ^ (row & 31)) << 3)]);
    }
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 307200;
  std::vector<uint16_t> hW((size_t)L * 512 * 512), hWp(hW.size()), hIn((size_t)M * 512);
  std::vector<float> hb(L * 512);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hW) v = h_f2bf(rnd() * 0.08f);
  for (auto& v : hIn) v = h_f2bf(rnd());
  for (auto& v : hb) v = rnd() * 0.1f;
  for (int ly = 0; ly < L; ++ly)
    for (int cb = 0; cb < 16; ++cb)
      for (int kk = 0; kk < 32; ++kk)
        for (int ln = 0; ln < 64; ++ln)
          for (int e = 0; e < 8; ++e)
            hWp[((((size_t)ly * 16 + cb) * 32 + kk) * 64 + ln) * 8 + e] = hW[((size_t)ly * 512 + cb * 32 + (ln & 31)) * 512 + kk * 16 + (ln >> 5) * 8 + e];
  uint16_t *dIn, *dWp, *dOut; float* db;
  hipMalloc(&dIn, hIn.size() * 2); hipMalloc(&dWp, hWp.size() * 2); hipMalloc(&dOut, hIn.size() * 2); hipMalloc(&db, hb.size() * 4);
  hipMemcpy(dIn, hIn.data(), hIn.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dWp, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  const int ntiles = (M + 127) / 128;
  const int grid = ntiles < 256 ? ntiles : 256;
  // correctness on the first 256 rows, 2 layers
  hipLaunchKernelGGL(chain128_kernel, dim3(2), dim3(512), 0, 0, dIn, dWp, db, dOut, 256, 2);
  std::vector<uint16_t> hOut(256 * 512);
  hipMemcpy(hOut.data(), dOut, hOut.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int r = 0; r < 256; r += 37) {
    std::vector<float> x(512), y(512);
    for (int c = 0; c < 512; ++c) x[c] = h_bf2f(hIn[(size_t)r * 512 + c]);
    for (int ly = 0; ly < 2; ++ly) {
      for (int o = 0; o < 512; ++o) {
        float a = 0;
        for (int c = 0; c < 512; ++c) a += x[c] * h_bf2f(hW[((size_t)ly * 512 + o) * 512 + c]);
        y[o] = h_bf2f(h_f2bf(fmaxf(a + hb[ly * 512 + o], 0.f)));
      }
      x = y;
    }
    for (int o = 0; o < 512; ++o) { maxerr = fmax(maxerr, fabs(x[o] - h_bf2f(hOut[(size_t)r * 512 + o]))); maxref = fmax(maxref, fabs(x[o])); }
  }
  printf("check: max |err| %.4g of max |ref| %.4g\n", maxerr, maxref);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int layers : {8, 1}) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(chain128_kernel, dim3(grid), dim3(512), 0, 0, dIn, dWp, db, dOut, M, layers);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(chain128_kernel, dim3(grid), dim3(512), 0, 0, dIn, dWp, db, dOut, M, layers);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / reps * 1e3;
    printf("PF %d rows %d layers %d: %.1f us per pass, %.2f us per layer and 128-row tile, %.1f TFLOP/s\n", PF, M, layers, us,
           us / layers / ((double)ntiles / grid), 2.0 * M * 512 * 512 * layers / us / 1e6);
  }
  return 0;
}
