// mfma_dtype_rate.hip -- does the chip sustain the same MFMA rate on fp16 operands as on bf16 operands? (round 6: every MFMA-dense kernel of
// this repo runs 4-10 % slower in its fp16 instantiation although the instruction counts and the documented MFMA rates are the same.)
// Whole chip, register operands only, ~100 ms per run so that the power management settles; operands are random bit patterns of "normal"
// magnitude (|x| in [0.5, 2)), or zeros (no switching activity in the multipliers).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_dtype_rate.hip -o tools/mfma_dtype_rate.bin && tools/mfma_dtype_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool F16>
__global__ __launch_bounds__(256) void k(float* out, int iters, int zero) {
  s16x8 a[4], b[2];
  uint32_t x = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
  for (int i = 0; i < 6; ++i)
    for (int e = 0; e < 8; ++e) {
      // sign random, exponent of [0.5, 2), mantissa random
      const uint32_t r = rnd();
      short v = F16 ? (short)(((r & 1) << 15) | ((14 + ((r >> 1) & 1)) << 10) | ((r >> 8) & 0x3ff))
                    : (short)(((r & 1) << 15) | ((126 + ((r >> 1) & 1)) << 7) | ((r >> 8) & 0x7f));
      if (zero) v = 0;
      (i < 4 ? a[i] : b[i - 4])[e] = v;
    }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (F16) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[u & 3]), __builtin_bit_cast(f16x8, b[(u >> 2) & 1]), acc[u & 3], 0, 0, 0);
      else acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[u & 3]), __builtin_bit_cast(bf16x8, b[(u >> 2) & 1]), acc[u & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  float* out; CK(hipMalloc(&out, 2048 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 2048, iters = 40000;   // 8 waves per CU-SIMD pair; 2048 * 4 waves * 40000 * 32 MFMAs * 32768 flop = 343 PFLOP... ~150 ms
  for (int rep = 0; rep < 3; ++rep)
    for (int zero = 0; zero < 2; ++zero)
      for (int f16 = 0; f16 < 2; ++f16) {
        CK(hipEventRecord(e0, 0));
        if (f16) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, out, iters, zero);
        else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, out, iters, zero);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 16;
        printf("rep %d  %-5s %-7s %8.2f ms  %7.1f TFLOP/s\n", rep, f16 ? "fp16" : "bf16", zero ? "zeros" : "random", ms, flop / (ms * 1e-3) / 1e12);
      }
  return 0;
}
