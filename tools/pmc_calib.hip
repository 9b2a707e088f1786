// pmc_calib.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for the access patterns of the head kernels?  
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pmc_calib.hip -o tools/pmc_calib.bin
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o pmc -- tools/pmc_calib.bin        (and again with WRITE_SIZE: tools/prof_r04.sh)
// The guide (MI355X_MICROARCH.md, HBM) says FETCH_SIZE counts 128-byte requests at 64 bytes for a wide coalesced stream -- "double it" --
// and asks for a calibration in one's own access pattern. Every kernel below moves a KNOWN number of bytes ONCE (buffers of 1 GiB: four
// times the Infinity Cache, so nothing is served on-die), in one of the patterns the step's kernels use:
//   calib_stream16      global_load_dwordx4, lane-contiguous (adamw's parameter / moment streams, the gathers' row copies)
//   calib_ldsdma_1k     global_load_lds_dwordx4, 1 KiB contiguous per wave-instruction
//   calib_ldsdma_rows   global_load_lds_dwordx4, 8 rows x 128 contiguous bytes per wave-instruction, rows 1 KiB apart: the K-stage
//                       pattern of rowgemm80 / rowseq / wgrad
//   calib_write16       global_store_dwordx4, lane-contiguous
//   calib_write_rows    16-byte stores forming 256-byte row segments 1 KiB apart: the GEMM epilogues' tile copy-out
// tools/prof_r04.sh divides the counter values by the byte counts printed here and stores the factors with the profile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void calib_stream16(const uint4* __restrict__ src, size_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void calib_ldsdma_1k(const uint16_t* __restrict__ src, size_t bytes, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[4][8][512];   // 4 waves x 8 KiB
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t per_wave = 8 * 1024, nwaves = (size_t)gridDim.x * 4, wid = (size_t)blockIdx.x * 4 + w;
  for (size_t off = wid * per_wave; off + per_wave <= bytes; off += nwaves * per_wave) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(reinterpret_cast<const char*>(src) + off + j * 1024 + l * 16), (lvoid_t*)&smem[w][j][0], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (smem[w][0][l] == 0x1234 && bytes == 1) *sink = 1;
}
__global__ __launch_bounds__(256) void calib_ldsdma_rows(const uint16_t* __restrict__ src, size_t rows, uint32_t* sink) {
  // src = [rows][512] bf16 (1 KiB rows); a wave walks 8-row groups, one K-stage (64 elements = 128 bytes per row) per instruction
  __shared__ __attribute__((aligned(16))) uint16_t smem[4][8][512];
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * 4, wid = (size_t)blockIdx.x * 4 + w;
  for (size_t r0 = wid * 8; r0 + 8 <= rows; r0 += nwaves * 8) {
    const uint16_t* g = src + (r0 + (l >> 3)) * 512 + (l & 7) * 8;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(g + kt * 64), (lvoid_t*)&smem[w][kt][0], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (smem[w][0][l] == 0x1234 && rows == 1) *sink = 1;
}
__global__ __launch_bounds__(256) void calib_write16(uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void calib_write_rows(uint16_t* __restrict__ dst, size_t rows) {
  // [rows][512] bf16: 16 lanes x 16 bytes = one 256-byte segment of a row (a 128-column tile), 4 rows per wave-instruction
  const size_t nthreads = (size_t)gridDim.x * 256, t = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (size_t q = t; q < rows * 16; q += nthreads) {
    const size_t row = q >> 4;
    *reinterpret_cast<uint4*>(dst + row * 512 + (q & 15) * 8) = make_uint4((uint32_t)q, 1, 2, 3);   // column tile 0 of every row: 256 of its 1024 bytes
  }
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  void *a, *b; uint32_t* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(calib_stream16, dim3(2048), dim3(256), 0, 0, (const uint4*)a, bytes / 16, sink);
  hipLaunchKernelGGL(calib_ldsdma_1k, dim3(1024), dim3(256), 0, 0, (const uint16_t*)b, bytes, sink);
  hipLaunchKernelGGL(calib_ldsdma_rows, dim3(1024), dim3(256), 0, 0, (const uint16_t*)a, bytes / 1024, sink);
  hipLaunchKernelGGL(calib_write16, dim3(2048), dim3(256), 0, 0, (uint4*)b, bytes / 16);
  hipLaunchKernelGGL(calib_write_rows, dim3(2048), dim3(256), 0, 0, (uint16_t*)a, bytes / 1024);
  CK(hipDeviceSynchronize());
  printf("{\"calib_stream16\": {\"read_bytes\": %zu}, \"calib_ldsdma_1k\": {\"read_bytes\": %zu}, \"calib_ldsdma_rows\": {\"read_bytes\": %zu}, "
         "\"calib_write16\": {\"write_bytes\": %zu}, \"calib_write_rows\": {\"write_bytes\": %zu}}\n",
         bytes, bytes, bytes, bytes, bytes / 4);
  return 0;
}
