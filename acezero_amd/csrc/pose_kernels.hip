// pose_kernels.hip -- per-image pose refinement inside the training step.
//
// Reference: refine_poses.py. `mlp` strategy: PoseNetwork(0, 128) (:15-72) maps each image's 3x4 world->cam pose to
// an additive update (:152-176, weight 0.1), the rotation part is re-orthonormalised with roma.special_gramschmidt
// (:136-150; roma 1.4.1 is not vendored -- Gram-Schmidt on the first two COLUMNS, third = cross product), the
// refined pose replaces the buffer pose in the projection (ace_trainer.py:527-533), gradients flow back through the
// Gram-Schmidt step into the network, which has its own AdamW (refine_poses.py:133) stepped only after
// `pose_refinement_wait` iterations (ace_trainer.py:634-636).
//
// MI355X design: the reference evaluates the network on B = 5120 per-patch copies of the poses; all patches of an
// image share its pose, so here the network runs once per IMAGE (n_images rows, typically 10^2..10^4) in fp32 on the
// vector ALU (the reference keeps this part outside autocast too), and the per-patch pose gradients are reduced per
// image in a fixed order (no atomics). The work is tiny (70 924 parameters) next to the head, so it is expressed
// with one generic LDS-tiled fp32 GEMM kernel instead of bespoke fused kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "head_kernels.h"

namespace acez {

// C[m][n] = epi( sum_k A(m,k) * B(k,n) ), generic strides, 64x64 tile, 256 threads x (4x4), BK = 16.
// grid.z splits K; slice z writes its partial to C + z * c_split (reduced by small_reduce_kernel).
struct SGemmArgs {
  const float* A; int64_t sa_m, sa_k;
  const float* B; int64_t sb_k, sb_n;
  float* C; int64_t sc_m, sc_n;
  int M, N, K;
  const float* bias;   // [N] or null
  const float* add;    // same strides as C, or null (added before relu)
  const float* mask;   // same strides as C, or null: C is zeroed where mask <= 0 (relu backward)
  int relu;
  float scale;         // multiplies the product (before bias/add)
  int ksplit; int64_t c_split;
  const int* active;
  float* c_lastcol;    // if set: column N-1 of B reads as 1.0 and column N-1 of the result goes to c_lastcol[m] (+ z * c_split)
};

__global__ __launch_bounds__(256) void sgemm_small_kernel(SGemmArgs a) {
  if (a.active && !*a.active) return;
  __shared__ float sA[16][65], sB[16][65];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int kper = (a.K + a.ksplit - 1) / a.ksplit;
  const int kb = blockIdx.z * kper, ke = min(a.K, kb + kper);
  float acc[4][4] = {};
  for (int k0 = kb; k0 < ke; k0 += 16) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int q = t + 256 * p, kk = q >> 6, i = q & 63;
      const int k = k0 + kk;
      sA[kk][i] = (k < ke && m0 + i < a.M) ? a.A[(int64_t)(m0 + i) * a.sa_m + (int64_t)k * a.sa_k] : 0.f;
      const bool ones = a.c_lastcol && (n0 + i == a.N - 1);
      sB[kk][i] = (k < ke && n0 + i < a.N) ? (ones ? 1.f : a.B[(int64_t)k * a.sb_k + (int64_t)(n0 + i) * a.sb_n]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = sA[kk][ty * 4 + i]; bv[i] = sB[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* C = a.C + (int64_t)blockIdx.z * a.c_split;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= a.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= a.N) continue;
      if (a.c_lastcol && n == a.N - 1) { a.c_lastcol[(int64_t)blockIdx.z * a.c_split + m] = acc[i][j] * a.scale; continue; }
      const int64_t o = (int64_t)m * a.sc_m + (int64_t)n * a.sc_n;
      float v = acc[i][j] * a.scale;
      if (a.bias) v += a.bias[n];
      if (a.add) v += a.add[o];
      if (a.relu) v = fmaxf(v, 0.f);
      if (a.mask && !(a.mask[o] > 0.f)) v = 0.f;
      C[o] = v;
    }
  }
}

// out[i] = sum_z part[z * stride + i]  (fixed order)
__global__ __launch_bounds__(256) void small_reduce_kernel(const float* part, int64_t stride, int nz, float* out, int64_t n, const int* active) {
  if (active && !*active) return;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int z = 0; z < nz; ++z) acc += part[(int64_t)z * stride + i];
  out[i] = acc;
}

// column sums: out[n] = sum_m X[m][n] for X [M][N] row-major, one wave per column, lane-strided + butterfly
__global__ __launch_bounds__(256) void colsum_kernel(const float* X, int M, int N, float* out, const int* active) {
  if (active && !*active) return;
  const int lane = threadIdx.x & 63;
  const int n = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (n >= N) return;
  float acc = 0.f;
  for (int m = lane; m < M; m += 64) acc += X[(int64_t)m * N + n];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) out[n] = acc;
}

// refined pose of image i: P = T0[:3] + w * delta (3x4), rotation columns re-orthonormalised (special_gramschmidt),
// written as a 4x4 (last row 0 0 0 1) for the loss kernel
__global__ __launch_bounds__(256) void pose_compose_kernel(const float* T0 /*[I][16]*/, const float* delta /*[I][12]*/, float w,
                                                           float* out /*[I][16]*/, int n_images, const int* active) {
  if (active && !*active) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_images) return;
  float P[12];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) P[r * 4 + c] = T0[(size_t)i * 16 + r * 4 + c] + w * delta[(size_t)i * 12 + r * 4 + c];
  float x[3] = {P[0], P[4], P[8]}, y[3] = {P[1], P[5], P[9]};
  const float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  x[0] /= nx; x[1] /= nx; x[2] /= nx;
  const float d = x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
  y[0] -= d * x[0]; y[1] -= d * x[1]; y[2] -= d * x[2];
  const float ny = sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
  y[0] /= ny; y[1] /= ny; y[2] /= ny;
  const float z[3] = {x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0]};
  float* o = out + (size_t)i * 16;
#pragma unroll
  for (int r = 0; r < 3; ++r) { o[r * 4 + 0] = x[r]; o[r * 4 + 1] = y[r]; o[r * 4 + 2] = z[r]; o[r * 4 + 3] = P[r * 4 + 3]; }
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
}

// per-image gradient of the refined pose -> gradient of the network output (delta), through the Gram-Schmidt step
__global__ __launch_bounds__(256) void pose_compose_bwd_kernel(const float* T0, const float* delta, float w, const float* dT /*[I][12]*/,
                                                               float* ddelta /*[I][12]*/, int n_images, const int* active) {
  if (active && !*active) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_images) return;
  float P[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) P[k] = T0[(size_t)i * 16 + k] + w * delta[(size_t)i * 12 + k];
  const float xr[3] = {P[0], P[4], P[8]}, yr[3] = {P[1], P[5], P[9]};
  const float nx = sqrtf(xr[0] * xr[0] + xr[1] * xr[1] + xr[2] * xr[2]);
  const float x[3] = {xr[0] / nx, xr[1] / nx, xr[2] / nx};
  const float d = x[0] * yr[0] + x[1] * yr[1] + x[2] * yr[2];
  const float yp[3] = {yr[0] - d * x[0], yr[1] - d * x[1], yr[2] - d * x[2]};
  const float ny = sqrtf(yp[0] * yp[0] + yp[1] * yp[1] + yp[2] * yp[2]);
  const float y[3] = {yp[0] / ny, yp[1] / ny, yp[2] / ny};
  const float* g = dT + (size_t)i * 12;
  float gx[3] = {g[0], g[4], g[8]}, gy[3] = {g[1], g[5], g[9]};
  const float gz[3] = {g[2], g[6], g[10]};
  // z = x cross y:  dx += y cross gz,  dy += gz cross x
  gx[0] += y[1] * gz[2] - y[2] * gz[1]; gx[1] += y[2] * gz[0] - y[0] * gz[2]; gx[2] += y[0] * gz[1] - y[1] * gz[0];
  gy[0] += gz[1] * x[2] - gz[2] * x[1]; gy[1] += gz[2] * x[0] - gz[0] * x[2]; gy[2] += gz[0] * x[1] - gz[1] * x[0];
  // y = yp / |yp|
  const float ydg = y[0] * gy[0] + y[1] * gy[1] + y[2] * gy[2];
  const float gyp[3] = {(gy[0] - y[0] * ydg) / ny, (gy[1] - y[1] * ydg) / ny, (gy[2] - y[2] * ydg) / ny};
  // yp = yr - (x . yr) x
  const float xdg = x[0] * gyp[0] + x[1] * gyp[1] + x[2] * gyp[2];
  const float gyr[3] = {gyp[0] - x[0] * xdg, gyp[1] - x[1] * xdg, gyp[2] - x[2] * xdg};
  gx[0] -= d * gyp[0] + xdg * yr[0]; gx[1] -= d * gyp[1] + xdg * yr[1]; gx[2] -= d * gyp[2] + xdg * yr[2];
  // x = xr / |xr|
  const float xdgx = x[0] * gx[0] + x[1] * gx[1] + x[2] * gx[2];
  const float gxr[3] = {(gx[0] - x[0] * xdgx) / nx, (gx[1] - x[1] * xdgx) / nx, (gx[2] - x[2] * xdgx) / nx};
  float* o = ddelta + (size_t)i * 12;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    o[r * 4 + 0] = w * gxr[r];
    o[r * 4 + 1] = w * gyr[r];
    o[r * 4 + 2] = 0.f;          // the third column of the raw matrix does not reach the output
    o[r * 4 + 3] = w * g[r * 4 + 3];
  }
}

// dT[i][:] = sum over the batch rows whose image is i of row_dT[row][:], rows visited in increasing order: one wave per
// image scans the row->image table 64 rows at a time (ballot), so the sum order is fixed (no atomics)
__global__ __launch_bounds__(256) void pose_grad_reduce_kernel(const float* row_dT /*[n][12]*/, const int* row_image /*[n]*/, int n,
                                                               float* dT /*[I][12]*/, int n_images, const int* active) {
  if (active && !*active) return;
  const int lane = threadIdx.x & 63;
  const int img = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (img >= n_images) return;
  float acc = 0.f;  // lane k < 12 accumulates component k
  for (int r0 = 0; r0 < n; r0 += 64) {
    const int r = r0 + lane;
    const bool hit = (r < n) && (row_image[r] == img);
    unsigned long long m = __ballot(hit);
    while (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      if (lane < 12) acc += row_dT[(size_t)(r0 + b) * 12 + lane];
    }
  }
  if (lane < 12) dT[(size_t)img * 12 + lane] = acc;
}

// torch.optim.AdamW on a small flat parameter vector with its own step counter; the gradient is the fixed-order
// sum of `nz` partial vectors. Applied only when *enable != 0 (ace_trainer.py:634-636).
__global__ __launch_bounds__(256) void adamw_small_kernel(float* p, float* m, float* v, const float* gpart, int64_t gstride, int nz, int64_t n,
                                                          const AdamScalars* sc, const int* enable, const int* active) {
  if (active && !*active) return;
  if (enable && !*enable) return;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const AdamScalars s = *sc;
  float gv = 0.f;
  for (int z = 0; z < nz; ++z) gv += gpart[(int64_t)z * gstride + i];
  float pv = p[i] * s.decay, mv = m[i], vv = v[i];
  mv = mv + (gv - mv) * s.one_minus_beta1;
  vv = vv * s.beta2 + s.one_minus_beta2 * gv * gv;
  const float denom = sqrtf(vv) / s.bc2_sqrt + s.eps;
  p[i] = pv - s.step_size * (mv / denom);
  m[i] = mv;
  v[i] = vv;
}

}  // namespace acez
