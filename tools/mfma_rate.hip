// mfma_rate.hip -- issue rate of the bf16 MFMAs the head kernels use, by accumulator pattern and waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_rate.hip -o tools/mfma_rate.bin && tools/mfma_rate.bin
// Question behind it (tools/pipe_proto.hip): a K loop of v_mfma_f32_16x16x32_bf16 with register-resident A fragments ran at 31
// cycles per instruction where 16 were expected. Patterns: NACC accumulators written round-robin, operands from registers only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int NA>   // NA distinct A fragments (register reuse pattern)
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* cyc, int iters) {
  bf16x8 a[NA], b[2];
  for (int i = 0; i < NA; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(0.001f * (threadIdx.x + i + e));
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)(0.002f * (threadIdx.x + 3 * i + e));
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u % NA], b[(u / NACC) & 1], acc[u % NACC], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* cyc, int iters) {
  bf16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(0.001f * (threadIdx.x + i + e));
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)(0.002f * (threadIdx.x + 3 * i + e));
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u / NACC) & 1], acc[u % NACC], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class K>
static int run(const char* name, K kern, int blocks, int threads, double flop_per_mfma, int per_iter, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 100);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double n = (double)iters * per_iter;
  const double waves = (double)blocks * threads / 64;
  printf("%-46s blocks %4d x %3d thr: %7.1f us  %6.2f memtime ticks / MFMA  %6.2f ns / MFMA / wave  %8.1f TFLOP/s\n", name, blocks, threads, ms * 1e3,
         (double)c / n, ms * 1e6 / n, waves * n * flop_per_mfma / (ms * 1e-3) / 1e12);
  return 0;
}
int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&cyc, 8));
  const double f16 = 16.0 * 16 * 32 * 2, f32 = 32.0 * 32 * 16 * 2;
  for (int blocks : {32, 256}) {
    for (int threads : {256, 512}) {
      run("16x16x32  1 accumulator (dependent chain)", k16<1, 4>, blocks, threads, f16, 64, out, cyc);
      run("16x16x32  2 accumulators", k16<2, 4>, blocks, threads, f16, 64, out, cyc);
      run("16x16x32  4 accumulators", k16<4, 4>, blocks, threads, f16, 64, out, cyc);
      run("16x16x32  4 accumulators, 32 A fragments", k16<4, 32>, blocks, threads, f16, 64, out, cyc);
      run("16x16x32  8 accumulators", k16<8, 4>, blocks, threads, f16, 64, out, cyc);
      run("16x16x32 10 accumulators", k16<10, 4>, blocks, threads, f16, 64, out, cyc);
      run("32x32x16  1 accumulator (dependent chain)", k32<1>, blocks, threads, f32, 32, out, cyc);
      run("32x32x16  2 accumulators", k32<2>, blocks, threads, f32, 32, out, cyc);
      run("32x32x16  4 accumulators", k32<4>, blocks, threads, f32, 32, out, cyc);
    }
  }
  return 0;
}
