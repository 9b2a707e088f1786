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
// image in a fixed order (no atomics). First version: one generic LDS-tiled fp32 GEMM launch per layer (26 launches) -- at 1000
// images that cost more than the whole head step; the fused kernels further down replaced it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "head_kernels.h"

namespace acez {

// ---------------------------------------------------------------------------------------------------
// Orthonormalisation of the updated pose's rotation block (refine_poses.py:135-150, --refinement_ortho):
//   0  roma.special_gramschmidt: Gram-Schmidt on the first two COLUMNS, third = cross product
//   1  roma.special_procrustes : nearest rotation R = U diag(1, 1, det(U V^T)) V^T of the SVD M = U S V^T
// (roma 1.4.1 is not vendored in the reference; both are restated from its documentation.) Each comes with its backward.
// The 3x3 SVD is computed from the Jacobi eigen-decomposition of M^T M (M is close to a rotation: singular values ~1).
// ---------------------------------------------------------------------------------------------------
struct Procrustes {
  float U[9], V[9], sg[3];   // U' = U D (sign fix folded in), V, signed singular values: M = U' diag(sg) V^T, R = U' V^T
};
__device__ __forceinline__ void procrustes_decompose(const float m[9], Procrustes& q) {
  float a[3][3], v[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      a[i][j] = m[0 * 3 + i] * m[0 * 3 + j] + m[1 * 3 + i] * m[1 * 3 + j] + m[2 * 3 + i] * m[2 * 3 + j];   // (M^T M)_ij
      v[i][j] = (i == j) ? 1.f : 0.f;
    }
#pragma unroll
  for (int sweep = 0; sweep < 6; ++sweep) {
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      constexpr int PP[3] = {0, 0, 1}, QQ[3] = {1, 2, 2}, RR[3] = {2, 1, 0};
      const int p = PP[pq], qq = QQ[pq], r = RR[pq];
      const float apq = a[p][qq];
      if (fabsf(apq) > 1e-30f) {
        const float tau = (a[qq][qq] - a[p][p]) / (2.f * apq);
        const float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
        const float c = 1.f / sqrtf(1.f + t * t), sn = t * c;
        a[p][p] -= t * apq;
        a[qq][qq] += t * apq;
        a[p][qq] = a[qq][p] = 0.f;
        const float arp = a[r][p], arq = a[r][qq];
        a[r][p] = a[p][r] = c * arp - sn * arq;
        a[r][qq] = a[qq][r] = sn * arp + c * arq;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float vp = v[i][p], vq = v[i][qq];
          v[i][p] = c * vp - sn * vq;
          v[i][qq] = sn * vp + c * vq;
        }
      }
    }
  }
  float sg[3] = {sqrtf(fmaxf(a[0][0], 1e-30f)), sqrtf(fmaxf(a[1][1], 1e-30f)), sqrtf(fmaxf(a[2][2], 1e-30f))};
  const float det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  const int ks = (sg[0] <= sg[1] && sg[0] <= sg[2]) ? 0 : ((sg[1] <= sg[2]) ? 1 : 2);   // the sign fix sits on the smallest singular value
  const float d = det < 0.f ? -1.f : 1.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float di = (i == ks) ? d : 1.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      q.U[r * 3 + i] = (m[r * 3 + 0] * v[0][i] + m[r * 3 + 1] * v[1][i] + m[r * 3 + 2] * v[2][i]) / sg[i] * di;
    q.sg[i] = sg[i] * di;
#pragma unroll
    for (int r = 0; r < 3; ++r) q.V[r * 3 + i] = v[r][i];
  }
}
__device__ __forceinline__ void procrustes_rotation(const Procrustes& q, float R[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[r * 3 + c] = q.U[r * 3 + 0] * q.V[c * 3 + 0] + q.U[r * 3 + 1] * q.V[c * 3 + 1] + q.U[r * 3 + 2] * q.V[c * 3 + 2];
}
// dL/dM from dL/dR:  Y = U'^T G V,  Z_ij = (Y_ij - Y_ji) / (sg_i + sg_j),  dL/dM = U' Z V^T
__device__ __forceinline__ void procrustes_backward(const Procrustes& q, const float G[9], float GM[9]) {
  float Y[3][3], Z[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc += q.U[r * 3 + i] * G[r * 3 + c] * q.V[c * 3 + j];
      Y[i][j] = acc;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float den = q.sg[i] + q.sg[j];
      Z[i][j] = (i == j || fabsf(den) < 1e-12f) ? 0.f : (Y[i][j] - Y[j][i]) / den;
    }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc += q.U[r * 3 + i] * Z[i][j] * q.V[c * 3 + j];
      GM[r * 3 + c] = acc;
    }
}

// refined 4x4 pose from the raw updated 3x4 matrix Pm (row-major [r][4])
__device__ __forceinline__ void pose_orthonormalise(const float Pm[12], int ortho, float* o /*16*/) {
  if (ortho == 1) {
    const float m[9] = {Pm[0], Pm[1], Pm[2], Pm[4], Pm[5], Pm[6], Pm[8], Pm[9], Pm[10]};
    Procrustes q;
    procrustes_decompose(m, q);
    float R[9];
    procrustes_rotation(q, R);
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[r * 4 + 0] = R[r * 3 + 0]; o[r * 4 + 1] = R[r * 3 + 1]; o[r * 4 + 2] = R[r * 3 + 2]; o[r * 4 + 3] = Pm[r * 4 + 3]; }
  } else {
    float x[3] = {Pm[0], Pm[4], Pm[8]}, y[3] = {Pm[1], Pm[5], Pm[9]};
    const float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    x[0] /= nx; x[1] /= nx; x[2] /= nx;
    const float d = x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
    y[0] -= d * x[0]; y[1] -= d * x[1]; y[2] -= d * x[2];
    const float ny = sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    y[0] /= ny; y[1] /= ny; y[2] /= ny;
    const float z[3] = {x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0]};
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[r * 4 + 0] = x[r]; o[r * 4 + 1] = y[r]; o[r * 4 + 2] = z[r]; o[r * 4 + 3] = Pm[r * 4 + 3]; }
  }
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
}
// gradient wrt the raw updated matrix Pm (scaled by w -> gradient wrt the network output / the pose parameters) from the gradient g
// wrt the refined pose (both 3x4 row-major)
__device__ __forceinline__ void pose_orthonormalise_bwd(const float Pm[12], const float* g, int ortho, float w, float o[12]) {
  if (ortho == 1) {
    const float m[9] = {Pm[0], Pm[1], Pm[2], Pm[4], Pm[5], Pm[6], Pm[8], Pm[9], Pm[10]};
    Procrustes q;
    procrustes_decompose(m, q);
    const float G[9] = {g[0], g[1], g[2], g[4], g[5], g[6], g[8], g[9], g[10]};
    float GM[9];
    procrustes_backward(q, G, GM);
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[r * 4 + 0] = w * GM[r * 3 + 0]; o[r * 4 + 1] = w * GM[r * 3 + 1]; o[r * 4 + 2] = w * GM[r * 3 + 2]; o[r * 4 + 3] = w * g[r * 4 + 3]; }
    return;
  }
  const float xr[3] = {Pm[0], Pm[4], Pm[8]}, yr[3] = {Pm[1], Pm[5], Pm[9]};
  const float nx = sqrtf(xr[0] * xr[0] + xr[1] * xr[1] + xr[2] * xr[2]);
  const float x[3] = {xr[0] / nx, xr[1] / nx, xr[2] / nx};
  const float d = x[0] * yr[0] + x[1] * yr[1] + x[2] * yr[2];
  const float yp[3] = {yr[0] - d * x[0], yr[1] - d * x[1], yr[2] - d * x[2]};
  const float ny = sqrtf(yp[0] * yp[0] + yp[1] * yp[1] + yp[2] * yp[2]);
  const float y[3] = {yp[0] / ny, yp[1] / ny, yp[2] / ny};
  float gx[3] = {g[0], g[4], g[8]}, gy[3] = {g[1], g[5], g[9]};
  const float gz[3] = {g[2], g[6], g[10]};
  // z = x cross y:  dx += y cross gz,  dy += gz cross x
  gx[0] += y[1] * gz[2] - y[2] * gz[1]; gx[1] += y[2] * gz[0] - y[0] * gz[2]; gx[2] += y[0] * gz[1] - y[1] * gz[0];
  gy[0] += gz[1] * x[2] - gz[2] * x[1]; gy[1] += gz[2] * x[0] - gz[0] * x[2]; gy[2] += gz[0] * x[1] - gz[1] * x[0];
  const float ydg = y[0] * gy[0] + y[1] * gy[1] + y[2] * gy[2];
  const float gyp[3] = {(gy[0] - y[0] * ydg) / ny, (gy[1] - y[1] * ydg) / ny, (gy[2] - y[2] * ydg) / ny};
  const float xdg = x[0] * gyp[0] + x[1] * gyp[1] + x[2] * gyp[2];
  const float gyr[3] = {gyp[0] - x[0] * xdg, gyp[1] - x[1] * xdg, gyp[2] - x[2] * xdg};
  gx[0] -= d * gyp[0] + xdg * yr[0]; gx[1] -= d * gyp[1] + xdg * yr[1]; gx[2] -= d * gyp[2] + xdg * yr[2];
  const float xdgx = x[0] * gx[0] + x[1] * gx[1] + x[2] * gx[2];
  const float gxr[3] = {(gx[0] - x[0] * xdgx) / nx, (gx[1] - x[1] * xdgx) / nx, (gx[2] - x[2] * xdgx) / nx};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    o[r * 4 + 0] = w * gxr[r];
    o[r * 4 + 1] = w * gyr[r];
    o[r * 4 + 2] = 0.f;          // the third column of the raw matrix does not reach the output
    o[r * 4 + 3] = w * g[r * 4 + 3];
  }
}

// refined pose of image i: P = T0[:3] + w * delta (3x4), rotation columns re-orthonormalised (special_gramschmidt),
// written as a 4x4 (last row 0 0 0 1) for the loss kernel
__global__ __launch_bounds__(256) void pose_compose_kernel(const float* T0 /*[I][16]*/, const float* delta /*[I][12]*/, float w,
                                                           float* out /*[I][16]*/, int n_images, const int* active, int ortho) {
  if (active && !*active) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_images) return;
  float P[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) P[k] = T0[(size_t)i * 16 + k] + w * delta[(size_t)i * 12 + k];
  float o[16];
  pose_orthonormalise(P, ortho, o);
#pragma unroll
  for (int k = 0; k < 16; ++k) out[(size_t)i * 16 + k] = o[k];
}

// per-image gradient of the refined pose -> gradient of the network output (delta), through the orthonormalisation
__global__ __launch_bounds__(256) void pose_compose_bwd_kernel(const float* T0, const float* delta, float w, const float* dT /*[I][12]*/,
                                                               float* ddelta /*[I][12]*/, int n_images, const int* active, int ortho) {
  if (active && !*active) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_images) return;
  float P[12], o[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) P[k] = T0[(size_t)i * 16 + k] + w * delta[(size_t)i * 12 + k];
  pose_orthonormalise_bwd(P, dT + (size_t)i * 12, ortho, w, o);
#pragma unroll
  for (int k = 0; k < 12; ++k) ddelta[(size_t)i * 12 + k] = o[k];
}

// ---------------------------------------------------------------------------------------------------
// Fused pose network (refine_poses.py:15-72 PoseNetwork(0, 128) + :152-176): with one generic GEMM launch per layer the
// network cost 20 launches of ~23 us at 1000 images (32 workgroups each) -- more than the whole head. The three kernels
// below run it data-parallel over tiles of 16 images on the fp32 matrix cores, weights read straight from L2:
//   pose_mlp_fwd_kernel    all 7 layers + compose/Gram-Schmidt, keeps the activations the backward needs
//   pose_mlp_bwd_kernel    compose backward + the chain of input gradients, keeps every layer's output gradient
//   pose_mlp_wgrad_kernel  dW = dY^T X, db = column sums, for all layers in one launch: a workgroup owns a 16 x 16 tile, its four
//                          waves a quarter of the images each, combined in wave order
// fp32 throughout (the reference keeps this network outside autocast). Parameter order = named_parameters():
// head_skip, conv1, conv2, conv3, fc1, fc2, fc3 (weight, bias each).
// ---------------------------------------------------------------------------------------------------
struct PoseNetArgs {
  const float* P;          // flat parameters
  const float* Wt;         // [4][128][128] transposed copies of conv2, conv3, fc1, fc2 (pose_transpose_kernel), forward only
  const float* T0;         // [I][16] original world->cam poses
  int I;
  float w;                 // update weight (refine_poses.py:165)
  float *a1, *a2, *a3, *r, *f1, *f2;   // [I][128] activations
  float* delta;            // [I][12]
  float* pose_cur;         // [I][16] refined poses (forward) 
  const float* dT;         // [I][12] gradient wrt the refined pose (backward)
  float *ddelta, *dz2, *dz1, *dr, *dzc3, *dzc2, *dzc1;   // backward: gradients wrt each layer's pre-activation output
  const int* active;
  int ortho;               // 0 gram-schmidt, 1 procrustes (--refinement_ortho)
  unsigned long long* trace;   // diagnostics build (ACEZ_POSE_TRACE=1, tools/pose_trace.py): [tiles][16] s_memtime stamps of thread 0; else null
};
#ifdef ACEZ_DIAG
#define PN_STAMP(a, tile, i) do { __builtin_amdgcn_sched_barrier(0); if ((a).trace && threadIdx.x == 0) (a).trace[(size_t)(tile) * 16 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PN_STAMP(a, tile, i) do { } while (0)
#endif
constexpr int64_t PN_SKIP_W = 0, PN_SKIP_B = 1536, PN_C1_W = 1664, PN_C1_B = 3200, PN_C2_W = 3328, PN_C2_B = 19712, PN_C3_W = 19840,
                  PN_C3_B = 36224, PN_F1_W = 36352, PN_F1_B = 52736, PN_F2_W = 52864, PN_F2_B = 69248, PN_F3_W = 69376, PN_F3_B = 70912;
constexpr int PN_IMG = 16;     // images per workgroup

typedef __attribute__((ext_vector_type(4))) float pn_f4;

// One layer for a tile of 16 images on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate):
//   out[n][img] = epi( sum_k A(n, k) * in[k][img] ),  A(n, k) = Wa[n * si + k * sk]   (n < N, k < K, K % 4 == 0)
// The A operand is read straight from global memory (L2-resident, 284 KiB for the whole network): no weight staging, so a
// layer is ~32 dependent MFMA steps instead of a 128-iteration LDS-latency-bound loop. Wave w owns the 16-row blocks 2w and
// 2w + 1 of the output (lane: column = image l & 15, rows 4 * (l >> 4) + r). in: LDS [K][16]; out: LDS [N][16] + global [I][N].
//   forward : A = W (si = K, sk = 1) or its transposed copy, epi = (+ bias, + add, relu)
//   backward: A = W^T (si = 1, sk = row pitch of W), epi = (mask by the stored activation)
// The A operands of a layer do not depend on the previous layer's result, only the B operand (activations in LDS) does: they
// are fetched one layer AHEAD (pn_fetch) so that a layer costs its MFMA chain, not an L2 round trip (~1 us) plus the chain.
template <int K>
struct PnA {
  float a0[K / 4], a1[K / 4];
};
template <int K>
__device__ __forceinline__ void pn_fetch(PnA<K>& A, const float* __restrict__ Wa, int si, int sk, int N) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = l & 15, lk = l >> 4, n0 = w * 32;
  const int r0 = min(n0 + li, N - 1), r1 = min(n0 + 16 + li, N - 1);
  const float* p0 = Wa + (size_t)r0 * si + (size_t)lk * sk;
  const float* p1 = Wa + (size_t)r1 * si + (size_t)lk * sk;
  const bool v0 = n0 + li < N, v1 = n0 + 16 + li < N;
#pragma unroll
  for (int q = 0; q < K / 4; ++q) {
    const float x0 = p0[(size_t)(4 * q) * sk], x1 = p1[(size_t)(4 * q) * sk];
    A.a0[q] = v0 ? x0 : 0.f;
    A.a1[q] = v1 ? x1 : 0.f;
  }
}
template <int K>
__device__ __forceinline__ void pn_mfma_layer(const PnA<K>& A, int N, const float* __restrict__ bias, const float* sIn, bool relu,
                                              const float* sAdd, const float* __restrict__ gMask, float* sOut, float* __restrict__ gOut,
                                              int i0, int I) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = l & 15, lk = l >> 4;
  const int n0 = w * 32, img = i0 + li;
  // bias / relu-mask values of this lane's 8 output rows: independent of the MFMA chain, requested before it
  float bv[2][4], mv[2][4];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = min(n0 + 16 * blk + 4 * lk + r, N - 1);
      bv[blk][r] = bias ? bias[n] : 0.f;
      mv[blk][r] = gMask ? gMask[(size_t)min(img, I - 1) * N + n] : 1.f;
    }
  __syncthreads();   // sIn complete, previous readers of sOut done
  if (n0 < N) {
    const bool two = n0 + 16 < N;
    pn_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
      const float bq = sIn[(4 * q + lk) * PN_IMG + li];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A.a0[q], bq, acc0, 0, 0, 0);
      if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A.a1[q], bq, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int nb = n0 + 16 * blk + 4 * lk;   // rows nb .. nb+3 of this lane
      if (nb >= N) continue;
      const pn_f4 acc = blk ? acc1 : acc0;
      float v[4] = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        v[r] += bv[blk][r];
        if (sAdd) v[r] += sAdd[n * PN_IMG + li];
        if (relu) v[r] = fmaxf(v[r], 0.f);
        if (!(mv[blk][r] > 0.f) || img >= I) v[r] = 0.f;
        sOut[n * PN_IMG + li] = v[r];
      }
      if (img < I) *reinterpret_cast<float4*>(gOut + (size_t)img * N + nb) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// refined poses of the 16 images of tile `tile` (all 256 threads of a workgroup; 25 KiB of LDS)
__device__ __forceinline__ void pose_mlp_fwd_body(const PoseNetArgs& a, const int tile) {
  __shared__ float sT[12 * PN_IMG], sX[128 * PN_IMG], sY[128 * PN_IMG], sZ[128 * PN_IMG];
  const int t = threadIdx.x, i0 = tile * PN_IMG;
  if (t < 12 * PN_IMG) {
    const int k = t / PN_IMG, i = t % PN_IMG;
    sT[k * PN_IMG + i] = (i0 + i < a.I) ? a.T0[(size_t)(i0 + i) * 16 + k] : 0.f;
  }
  const float* P = a.P;
  const float* Wt = a.Wt;   // [4][k][n] transposed copies of conv2, conv3, fc1, fc2: consecutive lanes read consecutive floats
  PnA<12> A12;
  PnA<128> Aa, Ab;
  pn_fetch<12>(A12, P + PN_C1_W, 12, 1, 128);
  pn_fetch<128>(Aa, Wt + 0 * 16384, 1, 128, 128);
  pn_mfma_layer<12>(A12, 128, P + PN_C1_B, sT, true, nullptr, nullptr, sX, a.a1, i0, a.I);              // x1 = relu(conv1(T))
  pn_fetch<128>(Ab, Wt + 1 * 16384, 1, 128, 128);
  pn_mfma_layer<128>(Aa, 128, P + PN_C2_B, sX, true, nullptr, nullptr, sY, a.a2, i0, a.I);              // x2
  pn_fetch<12>(A12, P + PN_SKIP_W, 12, 1, 128);
  pn_mfma_layer<128>(Ab, 128, P + PN_C3_B, sY, true, nullptr, nullptr, sZ, a.a3, i0, a.I);              // x3
  pn_fetch<128>(Aa, Wt + 2 * 16384, 1, 128, 128);
  pn_mfma_layer<12>(A12, 128, P + PN_SKIP_B, sT, false, sZ, nullptr, sX, a.r, i0, a.I);                 // res = head_skip(T) + x3
  pn_fetch<128>(Ab, Wt + 3 * 16384, 1, 128, 128);
  pn_mfma_layer<128>(Aa, 128, P + PN_F1_B, sX, true, nullptr, nullptr, sY, a.f1, i0, a.I);              // relu(fc1(res))
  pn_fetch<128>(Aa, P + PN_F3_W, 128, 1, 12);
  pn_mfma_layer<128>(Ab, 128, P + PN_F2_B, sY, true, nullptr, nullptr, sZ, a.f2, i0, a.I);              // relu(fc2(.))
  float* sD = sX;   // [12][16]
  pn_mfma_layer<128>(Aa, 12, P + PN_F3_B, sZ, false, nullptr, nullptr, sD, a.delta, i0, a.I);           // fc3: the pose update
  __syncthreads();
  // P = T + w * delta and the Gram-Schmidt step, one thread per image
  if (t < PN_IMG && i0 + t < a.I) {
    float Pm[12], o[16];
#pragma unroll
    for (int k = 0; k < 12; ++k) Pm[k] = sT[k * PN_IMG + t] + a.w * sD[k * PN_IMG + t];
    pose_orthonormalise(Pm, a.ortho, o);
    float* dst = a.pose_cur + (size_t)(i0 + t) * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) dst[k] = o[k];
  }
}

__global__ __launch_bounds__(256) void pose_mlp_fwd_kernel(PoseNetArgs a) {
  if (a.active && !*a.active) return;
  pose_mlp_fwd_body(a, blockIdx.x);
}

// Wt[l][k][n] = W_l[n][k] for the four 128 x 128 layers (conv2, conv3, fc1, fc2): 64 K elements, once per step
__global__ __launch_bounds__(256) void pose_transpose_kernel(const float* P, float* Wt, const int* active) {
  if (active && !*active) return;
  __shared__ float tile[32][33];
  const int64_t off[4] = {PN_C2_W, PN_C3_W, PN_F1_W, PN_F2_W};
  const int l = blockIdx.z, bx = blockIdx.x * 32, by = blockIdx.y * 32;   // W tile rows n = by.., cols k = bx..
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* W = P + off[l];
#pragma unroll
  for (int j = 0; j < 32; j += 8) tile[ty + j][tx] = W[(by + ty + j) * 128 + bx + tx];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 32; j += 8) Wt[(size_t)l * 16384 + (bx + ty + j) * 128 + by + tx] = tile[tx][ty + j];
}

// compose backward + the chain of input gradients for the 16 images of tile `tile`. dT16: gradient wrt the refined poses of these 16
// images, [16][12] (global or LDS); sD / sX / sY: LDS, 12 x 16, 128 x 16 and 128 x 16 floats
__device__ __forceinline__ void pose_mlp_bwd_body(const PoseNetArgs& a, const int tile, const float* dT16, float* sD, float* sX, float* sY) {
  const int t = threadIdx.x, i0 = tile * PN_IMG;
  // compose backward (one thread per image): gradient wrt the refined pose -> gradient wrt the network output
  if (t < PN_IMG) {
    float o[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) o[k] = 0.f;
    const int i = i0 + t;
    if (i < a.I) {
      float Pm[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) Pm[k] = a.T0[(size_t)i * 16 + k] + a.w * a.delta[(size_t)i * 12 + k];
      pose_orthonormalise_bwd(Pm, dT16 + t * 12, a.ortho, a.w, o);
#pragma unroll
      for (int k = 0; k < 12; ++k) a.ddelta[(size_t)i * 12 + k] = o[k];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) sD[k * PN_IMG + t] = o[k];
  }
  const float* P = a.P;
  // dX[k][img] = sum_n W[n][k] dY[n][img]: A(k, n) = W[n * 128 + k] -> si = 1, sk = 128
  PnA<12> A12;
  PnA<128> Aa, Ab;
  pn_fetch<12>(A12, P + PN_F3_W, 1, 128, 128);
  pn_fetch<128>(Aa, P + PN_F2_W, 1, 128, 128);
  pn_mfma_layer<12>(A12, 128, nullptr, sD, false, nullptr, a.f2, sX, a.dz2, i0, a.I);     // through fc3, relu'(fc2 out)
  pn_fetch<128>(Ab, P + PN_F1_W, 1, 128, 128);
  pn_mfma_layer<128>(Aa, 128, nullptr, sX, false, nullptr, a.f1, sY, a.dz1, i0, a.I);     // through fc2, relu'(fc1 out)
  pn_fetch<128>(Aa, P + PN_C3_W, 1, 128, 128);
  pn_mfma_layer<128>(Ab, 128, nullptr, sY, false, nullptr, nullptr, sX, a.dr, i0, a.I);   // through fc1: gradient of res
  pn_fetch<128>(Ab, P + PN_C2_W, 1, 128, 128);
  // x3 = relu(conv3(x2)): its pre-activation gradient is d(res) masked
  __syncthreads();
  for (int idx = t; idx < 128 * PN_IMG; idx += 256) {
    const int k = idx / PN_IMG, i = idx % PN_IMG, img = i0 + i;
    float v = sX[idx];
    if (!(img < a.I && a.a3[(size_t)img * 128 + k] > 0.f)) v = 0.f;
    sY[idx] = v;
    if (img < a.I) a.dzc3[(size_t)img * 128 + k] = v;
  }
  pn_mfma_layer<128>(Aa, 128, nullptr, sY, false, nullptr, a.a2, sX, a.dzc2, i0, a.I);
  pn_mfma_layer<128>(Ab, 128, nullptr, sX, false, nullptr, a.a1, sY, a.dzc1, i0, a.I);
}

__global__ __launch_bounds__(256) void pose_mlp_bwd_kernel(PoseNetArgs a) {
  if (a.active && !*a.active) return;
  __shared__ float sD[12 * PN_IMG], sX[128 * PN_IMG], sY[128 * PN_IMG];
  pose_mlp_bwd_body(a, blockIdx.x, a.dT + (size_t)blockIdx.x * PN_IMG * 12, sD, sX, sY);
}

// dW[o][k] = sum_i dY[i][o] X[i][k], db[o] = sum_i dY[i][o] for the seven layers, on the fp32 matrix cores straight from global
// memory: a workgroup owns one 16 x 16 tile of one layer's dW; its four waves take a quarter of the images each
// (A(o, img) = dY[img][o], B(img, k) = X[img][k], 4 images per MFMA step) and are combined in wave order -> the FINAL gradient,
// no partial buffers. Tiles with k-block 0 also produce the bias gradient (B = 1).
// torch.optim.AdamW for one element of the pose parameters. No fma contraction: the function is inlined into adamw_small_kernel and
// into the fused epilogue of pose_mlp_wgrad_kernel, and both must round identically.
__device__ __forceinline__ float adamw_small_one(float p, float gv, float& mv, float& vv, const AdamScalars& s) {
#pragma clang fp contract(off)
  const float pv = p * s.decay;
  mv = mv + (gv - mv) * s.one_minus_beta1;
  vv = vv * s.beta2 + s.one_minus_beta2 * gv * gv;
  const float denom = sqrtf(vv) / s.bc2_sqrt + s.eps;
  return pv - s.step_size * (mv / denom);
}

struct PoseWgradArgs {
  const float* dY[7]; const float* X[7];
  int O[7], K[7], xpitch[7];
  int64_t offW[7], offB[7];
  int I;
  float* grad;            // pose gradient vector (d_grad + n_params + 4)
  const int* active;
  int job_start[8];       // prefix sums of ceil(O / 16) * ceil(K / 16) over the layers
  // Fused optimiser (single-GPU step: the tile's gradient is final inside this workgroup, so AdamW -- adamw_small_kernel's arithmetic --
  // is applied on the spot and the transposed forward copies Wt[wt_slot] of the four 128 x 128 layers are refreshed): fuse = 0 in
  // the backward / all-reduce / update flow, where adamw_small_kernel and pose_transpose_kernel do this after the all-reduce.
  int fuse;
  float *p, *m, *v, *Wt;
  int wt_slot[7];         // index into Wt[4][128][128] of the layer, -1: none
  const AdamScalars* sc;
  const int* enable;      // st->pose_enable (ace_trainer.py:634-636)
  const int* fault;       // rowseq fault word: an abandoned step updates nothing
  unsigned long long* trace;   // diagnostics build (ACEZ_POSE_TRACE=1): [jobs][16] s_memtime stamps of thread 0; else null
};
// PW_WAVES waves per workgroup: each takes 1 / PW_WAVES of the images (8 waves at 1000 images: 32 MFMA steps = two batches of 16 loads;
// ten waves with ONE batch of 26 -- measured in round 5 -- shortened the slowest workgroup from 10.7 to 8.5 us and still lengthened the
// launch by 18 %: 640-thread workgroups take longer to place than the round trip they save)
template <int PW_BATCH, int PW_WAVES>
__global__ __launch_bounds__(64 * PW_WAVES, PW_BATCH <= 32 ? PW_WAVES / 2 : 2) void pose_mlp_wgrad_kernel(PoseWgradArgs a) {   // (<= 128 registers: two workgroups per CU, the 354 jobs in one round)
  // the launch's three flag words in ONE scalar round trip (tested one after the other -- `active` in front of everything, the other
  // two in front of the optimiser's operands -- they were two dependent ones: tools/pose_trace.py, 1.6 us from entry to the first request)
  const int f_active = a.active ? *a.active : 1, f_enable = a.fuse ? *a.enable : 0, f_fault = a.fuse ? *a.fault : 0;
  if (!f_active) return;
  const bool upd = a.fuse && f_enable && !f_fault;
  PN_STAMP(a, blockIdx.x, 0);
  __shared__ float sAcc[PW_WAVES][2][16][17];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = l & 15, lk = l >> 4;
  int layer = 0;   // (all seven bounds in one scalar load, compared in registers: the search loop was up to six dependent kernel-argument loads)
#pragma unroll
  for (int q = 1; q <= 6; ++q) layer += (int)blockIdx.x >= a.job_start[q] ? 1 : 0;
  const int O = a.O[layer], K = a.K[layer], xp = a.xpitch[layer];
  const int kb = (K + 15) / 16;
  const int job = (int)blockIdx.x - a.job_start[layer];
  const int o0 = (job / kb) * 16, k0 = (job % kb) * 16;
  const float* dY = a.dY[layer];
  const float* X = a.X[layer];
  const int per = ((a.I + PW_WAVES - 1) / PW_WAVES + 3) / 4 * 4;   // images per wave, multiple of 4
  const int ib = w * per, ie = min(a.I, ib + per);
  const bool vo = o0 + li < O, vk = k0 + li < K;
  const int oc = min(o0 + li, O - 1), kc = min(k0 + li, K - 1);
  pn_f4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;   // sum over this lane's images (4 g + lk) of dY[img][o0 + li]
  const bool want_bias = k0 == 0;
  // full groups of 4 images: plain pointer walks (the address arithmetic, not the MFMA, was the cost of this loop)
  const int nfull = (ie > ib) ? (ie - ib) / 4 : 0;
  const float* pa = dY + (size_t)(ib + lk) * O + oc;
  const float* pb = X + (size_t)(ib + lk) * xp + kc;
  const size_t sa = (size_t)4 * O, sb = (size_t)4 * xp;
  // the optimiser's operands (fused epilogue) do not depend on anything computed here: requested first
  float pw[4], mw[4], vw[4], pb4[4], mb4[4], vb4[4];
  if (upd && w == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = min(o0 + 4 * lk + r, O - 1), kk = min(k0 + li, K - 1);
      const size_t iw = a.offW[layer] + (size_t)o * K + kk, ib = a.offB[layer] + o;
      pw[r] = a.p[iw]; mw[r] = a.m[iw]; vw[r] = a.v[iw];
      pb4[r] = a.p[ib]; mb4[r] = a.m[ib]; vb4[r] = a.v[ib];
    }
  }
  PN_STAMP(a, blockIdx.x, 1);
  // PW_BATCH steps of operands are requested together (the compiler keeps a runtime-trip-count loop at one step per L2 round trip;
  // with 16 per batch a wave's 63 steps at 1000 images were four dependent round trips, with 64 they are one)
  for (int g0 = 0; g0 < nfull; g0 += PW_BATCH) {
    float av[PW_BATCH], bv[PW_BATCH];
#pragma unroll
    for (int j = 0; j < PW_BATCH; ++j) {
      const bool ok = g0 + j < nfull;
      const int g = min(g0 + j, nfull - 1);
      const float x = pa[g * sa], y = pb[g * sb];
      av[j] = (ok && vo) ? x : 0.f;
      bv[j] = (ok && vk) ? y : 0.f;
    }
    // two accumulator chains (even / odd steps), summed at the end: a dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles, two
    // independent ones every 32
#pragma unroll
    for (int j = 0; j < PW_BATCH; j += 2) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j + 1], bv[j + 1], acc2, 0, 0, 0);
    }
    // the bias gradient (tiles with k-block 0): this lane's images summed in order on the vector ALU. (As a second MFMA chain against
    // B = 1 it doubled the matrix time of exactly the tiles that end the launch: 8.5 us against 6.5 us for the others, tools/pose_trace.py.)
    if (want_bias) {
#pragma unroll
      for (int j = 0; j < PW_BATCH; ++j) bsum += av[j];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
  PN_STAMP(a, blockIdx.x, 2);
  if (ie > ib && ib + 4 * nfull < ie) {   // ragged last group
    const int img = ib + 4 * nfull + lk;
    const bool vi = img < ie;
    const int ic = min(img, a.I - 1);
    float av = dY[(size_t)ic * O + oc], bv = X[(size_t)ic * xp + kc];
    if (!(vi && vo)) av = 0.f;
    if (!(vi && vk)) bv = 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    if (want_bias) bsum += av;
  }
  // lane: column li (k), rows 4 * lk + r (o)
#pragma unroll
  for (int r = 0; r < 4; ++r) sAcc[w][0][4 * lk + r][li] = acc[r];
  if (want_bias) {   // the four image residues of the wave, (0 + 1) + (2 + 3) in every lane, then one value per output row
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lk == 0) sAcc[w][1][0][li] = bsum;
  }
  PN_STAMP(a, blockIdx.x, 3);
  __syncthreads();
  PN_STAMP(a, blockIdx.x, 4);
  if (w == 0) {
    AdamScalars s{};
    if (upd) s = *a.sc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int oo = 4 * lk + r, o = o0 + oo, kk = k0 + li;
      float g = sAcc[0][0][oo][li];
#pragma unroll
      for (int wv = 1; wv < PW_WAVES; ++wv) g += sAcc[wv][0][oo][li];   // wave order: fixed
      if (o < O && kk < K) {
        const size_t iw = a.offW[layer] + (size_t)o * K + kk;
        a.grad[iw] = g;
        if (upd) {
          const float pn = adamw_small_one(pw[r], g, mw[r], vw[r], s);
          a.p[iw] = pn; a.m[iw] = mw[r]; a.v[iw] = vw[r];
          if (a.wt_slot[layer] >= 0) a.Wt[(size_t)a.wt_slot[layer] * 16384 + (size_t)kk * 128 + o] = pn;
        }
      }
      if (want_bias && li == 0 && o < O) {
        float gb = sAcc[0][1][0][oo];
#pragma unroll
        for (int wv = 1; wv < PW_WAVES; ++wv) gb += sAcc[wv][1][0][oo];
        const size_t ib = a.offB[layer] + o;
        a.grad[ib] = gb;
        if (upd) {
          a.p[ib] = adamw_small_one(pb4[r], gb, mb4[r], vb4[r], s);
          a.m[ib] = mb4[r]; a.v[ib] = vb4[r];
        }
      }
    }
  }
  PN_STAMP(a, blockIdx.x, 5);
}

// dT[i][:] = sum over the batch rows whose image is i of row_dT[row][:], rows in increasing order (fixed order, no atomics).
// A workgroup owns 16 images. Every dependent global access costs an L2 round trip (~1 us), so the kernel is three phases with ONE
// round trip each: (1) each wave loads its quarter of the row -> image table into registers and appends the hits of its images, in row
// order, to an LDS list (ballot + popcount); (2) all hit rows are fetched together into LDS; (3) thread (image, component) walks the
// lists in order and adds from LDS.
// The three phases as a body over caller-provided LDS (MAX_HITS: list capacity per workgroup and pass, a multiple of 256):
//   sRow int[MAX_HITS], sRel u8[MAX_HITS], sTag u8[MAX_HITS] (8-byte aligned), sVal float[MAX_HITS][12], sCnt int[W];  out192: [TILE][12] sums (LDS or
// global), written by threads 0..12 TILE - 1. Every thread of the workgroup (W waves) must call it.
template <int MAX_HITS, int TILE = 16, int W = 4>
__device__ __forceinline__ void pose_grad_reduce_body(const float* __restrict__ row_dT, const int* __restrict__ row_image, const int n, const int i0,
                                                      int* sRow, unsigned char* sRel, unsigned char* sTag, float (*sVal)[12], int* sCnt, float* out192,
                                                      unsigned long long* stamps = nullptr) {
#ifdef ACEZ_DIAG
#define PGR_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (stamps && threadIdx.x == 0) stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PGR_STAMP(i) do { } while (0)
#endif
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  constexpr int NT = 64 * W;                   // threads of the workgroup (W waves: sCnt int[W])
  constexpr int cap = MAX_HITS / W;            // list capacity per wave
  constexpr int SCAN = 80 / W;                 // table entries per lane in flight: a wave's share of a 5120-row batch in ONE round trip
  // One pass over all rows when the lists fit (many images: ~5 rows per image). Otherwise (few images) the rows are taken in several
  // passes: first of a size chosen from the number of hits the failed pass counted (half-full lists on average), and if a list still
  // overflows, of MAX_HITS rows, which cannot. Row order is preserved in every case.
  int chunk = n;
  float result = 0.f;
  for (int attempt = 0; attempt < 3; ++attempt) {
    float acc = 0.f;                               // thread (image t / 12, component t % 12)
    bool overflow = false;
    for (int cb = 0; cb < n; cb += chunk) {
      // the lists are rewritten below: every reader of the pass before must be done with them (without this barrier a fast wave's
      // appends overwrote entries other waves were still summing -- wrong pose gradients whenever a tile needed more than one pass)
      if (cb > 0) __syncthreads();
      const int ce = min(n, cb + chunk);
      const int q = ((ce - cb + W - 1) / W + 63) / 64 * 64;      // rows per wave in this pass
      const int rb = cb + w * q, re = min(ce, rb + q);
      int cnt = 0;
      for (int c0 = rb; c0 < re; c0 += 64 * SCAN) {
        int rel[SCAN];
#pragma unroll
        for (int j = 0; j < SCAN; ++j) {
          const int r = c0 + 64 * j + lane;
          const int v = row_image[min(r, n - 1)];   // unconditional load (a load under a branch is waited for on the spot)
          rel[j] = (r < re) ? v - i0 : -1;
        }
#pragma unroll
        for (int j = 0; j < SCAN; ++j) {
          const bool hit = rel[j] >= 0 && rel[j] < TILE;
          const unsigned long long m = __ballot(hit);
          if (m == 0ull) continue;                 // (most groups of 64 rows hold no row of the tile's images)
          if (hit) {
            const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < cap) { sRow[w * cap + pos] = c0 + 64 * j + lane; sRel[w * cap + pos] = (unsigned char)rel[j]; }
          }
          cnt += __popcll(m);
        }
      }
      PGR_STAMP(0);                                // (this thread's table scan done)
      __syncthreads();                             // previous pass' readers of the lists are done
      if (lane == 0) sCnt[w] = cnt;
      __syncthreads();
      int cn[W], pre[W + 1];                       // the lists' counts and their prefix sums (list order = wave order = row order)
      pre[0] = 0;
      bool over = false;
#pragma unroll
      for (int v = 0; v < W; ++v) { cn[v] = sCnt[v]; pre[v + 1] = pre[v] + cn[v]; over = over || cn[v] > cap; }
      const int H = pre[W];
      if (over) {
        overflow = true;
        // hits per row seen in this pass -> rows per pass that would leave the lists half full (a multiple of the workgroup's threads, at least MAX_HITS)
        const long rows = (long)(ce - cb) * (MAX_HITS / 2) / H;
        chunk = attempt == 0 ? (int)max((long)MAX_HITS, rows / NT * NT) : MAX_HITS;   // MAX_HITS / W rows per wave: the lists cannot overflow
        break;
      }
      {   // fetch all hit rows, COMPACTED: hit hh of the lists' concatenation (wave order = row order) goes to sVal[hh], its image tag
          // to sTag[hh]. One load per thread covers 64 W / 12 hits (a 4-image tile of a 5120-row batch has ~20); more hits, more rounds.
        for (int f0 = 0; f0 < H * 12; f0 += NT) {
          const int f = f0 + t;
          const int hh = min(f / 12, H - 1), c = f % 12;
          int wv = 0, base = 0;
#pragma unroll
          for (int v = 1; v < W; ++v) { wv += hh >= pre[v] ? 1 : 0; base = hh >= pre[v] ? pre[v] : base; }
          const int slot = wv * cap + (hh - base);
          const float x = row_dT[(size_t)sRow[slot] * 12 + c];
          if (f < H * 12) {
            sVal[hh][c] = x;
            if (c == 0) sTag[hh] = sRel[slot];
          }
        }
      }
      __syncthreads();
      PGR_STAMP(1);                                // hit rows in LDS
      if (t < TILE * 12) {
        // the hits in row order, eight at a time: the image tags of a group are ONE 8-byte read and its values eight unconditional
        // reads, all requested together, then added in order -- one LDS round trip per group (hit by hit, with the value read under
        // the tag's branch, it was two dependent round trips each: 2.2 us for the ~20 hits of a 4-image tile, tools/pose_trace.py).
        // Entries past H are read (inside the arrays) and not added.
        const int im = t / 12, c = t % 12;
        for (int h0 = 0; h0 < H; h0 += 8) {
          const uint2 tg = *reinterpret_cast<const uint2*>(sTag + h0);
          float val[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) val[u] = sVal[min(h0 + u, MAX_HITS - 1)][c];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int tag = (int)(((u < 4 ? tg.x : tg.y) >> (8 * (u & 3))) & 0xffu);
            if (h0 + u < H && tag == im) acc += val[u];
          }
        }
      }
    }
    if (!overflow) { result = acc; break; }
    __syncthreads();
  }
  if (t < TILE * 12) out192[t] = result;
}

constexpr int PGR_MAX_HITS = 1024;   // per workgroup (16 images) and pass
__global__ __launch_bounds__(256) void pose_grad_reduce2_kernel(const float* row_dT /*[n][12]*/, const int* row_image /*[n]*/, int n,
                                                                float* dT /*[I][12]*/, int n_images, const int* active) {
  if (active && !*active) return;
  __shared__ int sRow[PGR_MAX_HITS];
  __shared__ unsigned char sRel[PGR_MAX_HITS];
  __shared__ __attribute__((aligned(8))) unsigned char sTag[PGR_MAX_HITS];
  __shared__ float sVal[PGR_MAX_HITS][12];
  __shared__ int sCnt[4];
  __shared__ float sOut[16 * 12];
  const int t = threadIdx.x, i0 = blockIdx.x * 16;
  pose_grad_reduce_body<PGR_MAX_HITS>(row_dT, row_image, n, i0, sRow, sRel, sTag, sVal, sCnt, sOut);
  if (t < 16 * 12 && i0 + t / 12 < n_images) dT[(size_t)(i0 + t / 12) * 12 + t % 12] = sOut[t];   // (each thread reads its own word back)
}

// ---------------------------------------------------------------------------------------------------
// S1 = per-image reduction of the loss kernel's per-row pose gradients + compose backward + the input-gradient chain of the pose
// network, for one tile of 16 images, in ONE workgroup pass (was two launches, 13 + 18 us at 1000 images). 28 KiB of LDS so that
// the workgroups can sit beside the optimiser's (adamw_pose_kernel, pose_fused.hip); the reduction's hit lists and the chain's
// activation tiles share storage.
// ---------------------------------------------------------------------------------------------------
constexpr int PS1_HITS = 512;
constexpr int PS1_SMEM_BYTES = PS1_HITS * 4 + 2 * PS1_HITS + 16 + 16 * 12 * 4 + PS1_HITS * 12 * 4;   // sRow, sRel, sTag, sCnt, sDT, sVal | (sD, sX, sY)
static_assert(PS1_HITS * 12 * 4 >= (12 + 128 + 128) * PN_IMG * 4, "the chain's tiles must fit in the hit-value area");
__device__ __forceinline__ void pose_s1_body(const PoseNetArgs& a, const float* row_dT, const int* row_image, const int n, const int tile, char* smem) {
  int* sRow = reinterpret_cast<int*>(smem);
  unsigned char* sRel = reinterpret_cast<unsigned char*>(smem + PS1_HITS * 4);
  unsigned char* sTag = reinterpret_cast<unsigned char*>(smem + PS1_HITS * 5);
  int* sCnt = reinterpret_cast<int*>(smem + PS1_HITS * 6);
  float* sDT = reinterpret_cast<float*>(smem + PS1_HITS * 6 + 16);
  float* area = sDT + 16 * 12;
  pose_grad_reduce_body<PS1_HITS>(row_dT, row_image, n, tile * PN_IMG, sRow, sRel, sTag, reinterpret_cast<float (*)[12]>(area), sCnt, sDT);
  __syncthreads();   // sDT complete; the hit values are dead, their area becomes the chain's tiles
  pose_mlp_bwd_body(a, tile, sDT, area, area + 12 * PN_IMG, area + (12 + 128) * PN_IMG);
}
__global__ __launch_bounds__(256) void pose_s1_kernel(PoseNetArgs a, const float* row_dT, const int* row_image, int n) {
  if (a.active && !*a.active) return;
  __shared__ __attribute__((aligned(16))) char smem[PS1_SMEM_BYTES];
  pose_s1_body(a, row_dT, row_image, n, blockIdx.x, smem);
}

// torch.optim.AdamW on a small flat parameter vector with its own step counter; the gradient is the fixed-order
// sum of `nz` partial vectors. Applied only when *enable != 0 (ace_trainer.py:634-636).
__global__ __launch_bounds__(256) void adamw_small_kernel(float* p, float* m, float* v, const float* gpart, int64_t gstride, int nz, int64_t n,
                                                          const AdamScalars* sc, const int* enable, const int* active, const int* fault) {
  if (active && !*active) return;
  if (fault && *fault) return;   // abandoned step (rowseq fault, head_kernels.hip)
  if (enable && !*enable) return;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const AdamScalars s = *sc;
  float gv = 0.f;
  for (int z = 0; z < nz; ++z) gv += gpart[(int64_t)z * gstride + i];
  float mv = m[i], vv = v[i];
  p[i] = adamw_small_one(p[i], gv, mv, vv, s);
  m[i] = mv;
  v[i] = vv;
}

}  // namespace acez
