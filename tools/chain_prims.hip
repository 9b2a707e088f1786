// Micro-benchmark of the synchronisation primitives the chain kernel can be built from (gfx950): cycles per operation,
// measured with s_memtime around 512 repetitions, for workgroups of 8 / 12 / 16 waves, one workgroup per CU (160 KiB LDS).
//   hipcc --offload-arch=gfx950 -O3 -o tools/chain_prims.bin tools/chain_prims.hip && tools/chain_prims.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) int lds_int;
#define REPS 512
__global__ void prims(long long* out, int mode) {
  extern __shared__ __attribute__((aligned(16))) int dyn[];
  lds_int* flags = (lds_int*)dyn;
  const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nw = blockDim.x >> 6;
  if (t < 64) flags[t] = 0;
  __syncthreads();
  long long t0 = 0, t1 = 0;
  if (mode == 0) {          // s_barrier, all waves
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; ++i) __builtin_amdgcn_s_barrier();
    t1 = __builtin_amdgcn_s_memtime();
  } else if (mode == 1) {   // poll hit: volatile ds_read_b32 + readfirstlane + compare (flag already set), every wave
    int acc = 0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; ++i) {
      while (__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(flags + (i & 7))) < 0) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      acc += i;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (acc == 12345) out[999] = acc;
  } else if (mode == 2) {   // flag_set: waitcnt lgkmcnt(0) + lane-0 ds_write
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; ++i) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (l == 0) *(volatile lds_int*)(flags + 8 + w) = i;
      asm volatile("" ::: "memory");
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else if (mode == 3) {   // ping-pong between wave 0 and wave 1 through two flags (round trip = 2 hand-offs); other waves idle-spin on a third flag
    if (w == 0) {
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 1; i <= REPS; ++i) {
        if (l == 0) *(volatile lds_int*)(flags + 16) = i;
        while (__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(flags + 17)) < i) __builtin_amdgcn_s_sleep(1);
      }
      t1 = __builtin_amdgcn_s_memtime();
      if (l == 0) *(volatile lds_int*)(flags + 18) = 1;
    } else if (w == 1) {
      for (int i = 1; i <= REPS; ++i) {
        while (__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(flags + 16)) < i) __builtin_amdgcn_s_sleep(1);
        if (l == 0) *(volatile lds_int*)(flags + 17) = i;
      }
    } else {
      while (__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(flags + 18)) < 1) __builtin_amdgcn_s_sleep(2);
    }
  } else if (mode == 4) {   // the same ping-pong without s_sleep in the spin
    if (w == 0) {
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 1; i <= REPS; ++i) {
        if (l == 0) *(volatile lds_int*)(flags + 16) = i;
        while (__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(flags + 17)) < i) {}
      }
      t1 = __builtin_amdgcn_s_memtime();
      if (l == 0) *(volatile lds_int*)(flags + 18) = 1;
    } else if (w == 1) {
      for (int i = 1; i <= REPS; ++i) {
        while (__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(flags + 16)) < i) {}
        if (l == 0) *(volatile lds_int*)(flags + 17) = i;
      }
    } else {
      while (__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(flags + 18)) < 1) __builtin_amdgcn_s_sleep(2);
    }
  } else if (mode == 5) {   // s_memtime + dependent use (the cost of a stamp without the store)
    long long s = 0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; ++i) s += __builtin_amdgcn_s_memtime();
    t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345) out[999] = s;
  } else if (mode == 6) {   // all-to-all: every wave sets its flag to i, then waits until all nw flags >= i (a flag barrier)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 1; i <= REPS; ++i) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (l == 0) *(volatile lds_int*)(flags + 32 + w) = i;
      for (;;) {
        int m = i;
        for (int k = 0; k < nw; ++k) m = min(m, *(volatile lds_int*)(flags + 32 + k));
        if (__builtin_amdgcn_readfirstlane(m) >= i) break;
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  if (l == 0) out[blockIdx.x * 16 + w] = t1 - t0;
}
int main() {
  long long* d;
  hipMalloc(&d, 1 << 16);
  hipFuncSetAttribute((const void*)prims, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const char* names[] = {"s_barrier", "poll hit (ds_read+readfirstlane)", "flag_set", "ping-pong round trip (s_sleep 1)", "ping-pong round trip (busy)",
                         "s_memtime", "flag barrier (all-to-all)"};
  for (int waves : {8, 12, 16})
    for (int mode = 0; mode < 7; ++mode) {
      hipMemset(d, 0, 1 << 16);
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(prims, dim3(160), dim3(64 * waves), 160 * 1024, 0, d, mode);
      hipDeviceSynchronize();
      long long h[16 * 160];
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      double s = 0; int c = 0;
      for (int b = 0; b < 160; ++b) { if (h[b * 16]) { s += (double)h[b * 16]; ++c; } }
      printf("waves %2d  %-36s %8.1f cycles/op (wave 0, mean of %d workgroups)\n", waves, names[mode], s / (c ? c : 1) / REPS, c);
    }
  return 0;
}
