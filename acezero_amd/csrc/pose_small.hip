// pose_small.hip -- the pose-refinement network on SMALL image tiles (T = 4 or 8 images per workgroup) with
// v_mfma_f32_4x4x1_16b_f32. Same arithmetic as pose_kernels.hip (refine_poses.py:15-72,152-176: fp32 products, fp32 accumulation),
// same global layouts ([I][128] activations / output gradients, [I][12] update, [I][16] refined poses), so that the weight-gradient
// kernel and the oracle tests do not care which tile size produced them.
//
// Why: with 16-image tiles (v_mfma_f32_16x16x4_f32) the network is a chain of 6-7 dependent layers, each 64 MFMAs of 32 cycles on
// every SIMD of the 63 CUs that 1000 images occupy (0.85 us of matrix time per layer and workgroup, 193 CUs idle): 14.6 us forward,
// 18 us backward. The 4x4x1 instruction multiplies sixteen 4 x 1 by 1 x 4 blocks at the same FLOP rate; with the weights' rows
// spread over the 16 blocks x 4 rows = 64 lanes and the SAME four images in every block, one instruction produces 64 output rows
// for 4 images: a tile of T images costs T / 16 of the matrix time per layer and the network spreads over 16 / T times more CUs.
//
// Roles (256 threads = 4 waves): wave w owns output rows 64 (w & 1) + lane and the reduction half k in [K/2 (w >> 1), K/2 (w >> 1) + K/2);
// the two halves are combined through LDS in a fixed order (lower half + upper half). Operand A (weights) sits in registers, one
// value per lane and k, fetched one layer ahead with coalesced loads: forward from the transposed copies Wt[k][n] (lane = n),
// backward from W[n][k] itself (lane = k: the output index of the transposed product). Operand B (activations [K][T] in LDS) is a
// broadcast read: lane l needs image l & 3 of its group.
//   D layout: lane 4 b + j holds rows 4 b + 0..3 (4 registers) of image j -> one float4 store per lane into the [I][N] activation arrays.
#include "pose_kernels.hip"

namespace acez {

// Row pitch of the 128-wide LDS tiles: 132 floats. A lane's B operands are 16-byte reads of image j = lane & 3: at a pitch of 128 floats
// the four images start in the same bank (4-way conflict on every read of the product loop), at 132 they are 16 bytes apart.
#ifndef PN4_PITCH
#define PN4_PITCH 132
#endif
__device__ __forceinline__ constexpr int pn4_ld(int n) { return n == 128 ? PN4_PITCH : n; }

template <int K>
struct Pn4A {
  float a[(K + 1) / 2];
};
// A(n, k) = Wa[n * si + k * sk] for this lane's row n = 64 (w & 1) + lane (zero past N) and this wave's half of k
template <int K>
__device__ __forceinline__ void pn4_fetch(Pn4A<K>& A, const float* __restrict__ Wa, int si, int sk, int N) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int KH = K / 2;
  const int n = 64 * (w & 1) + lane;
  const bool valid = n < N;
  const float* p = Wa + (size_t)min(n, N - 1) * si + (size_t)((w >> 1) * KH) * sk;
#pragma unroll
  for (int q = 0; q < KH; ++q) {
    const float x = p[(size_t)q * sk];
    A.a[q] = valid ? x : 0.f;
  }
}

// out[img][n] = epi( sum_k A(n, k) in[img][k] ) for the T images of the tile. LDS tiles are IMAGE-major: sIn [T][K], sOut [T][N],
// sPart / sAdd [T][128] (the upper reduction half on its way to the lower one; the residual input), gOut global [I][N]: a lane's B
// operands of four consecutive k are one 16-byte LDS read (broadcast: only four distinct addresses per instruction), and its four
// result rows one 16-byte LDS write. Every B operand of the wave's reduction half is read BEFORE the first MFMA -- left to the
// compiler, each pair of MFMAs sat behind its own ds_read + s_waitcnt lgkmcnt(0): 32 serial LDS round trips = 1.3 us per layer,
// which is why the 4-image tiles were no faster than the 16-image ones at first.
// epi = (+ bias, + sAdd, relu) forward / (mask by the stored activation M > 0, MASKED) backward. All 256 threads must call it.
// The relu masks of the backward chain (stored activations of the forward launch, [I][128]) are independent of everything the chain
// computes: pn4_mask requests a lane's four values of every image group as ONE 16-byte load, at the top of the chain for all of its
// layers. (Requested inside pn4_layer as four scalar loads under `gMask ? .. : 1.f`, the compiler folded the `> 0` test into the
// branch that loads: four SERIAL load -> s_waitcnt vmcnt(0) -> v_cmp round trips before the barrier of each masked layer, each of
// them also waiting for the weight prefetch of the next layer.)
template <int T>
struct Pn4Mask {
  float4 m[T / 4];
};
template <int T>
__device__ __forceinline__ void pn4_mask(Pn4Mask<T>& M, const float* __restrict__ gMask, int i0, int I) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nb = 64 * (w & 1) + 4 * (lane >> 2), j = lane & 3;
#pragma unroll
  for (int g = 0; g < T / 4; ++g) M.m[g] = *reinterpret_cast<const float4*>(gMask + (size_t)min(i0 + 4 * g + j, I - 1) * 128 + nb);
}

// The weights of a LATER layer, requested from inside this layer's product loop (KN != 0): one load after each step's MFMAs. A layer's
// 64 KiB of weights are 256 load instructions per workgroup, and the CU's one vector-memory path takes them at 64 B / clock: 0.5 us
// during which -- issued as a block ahead of the layer, as pn4_fetch does -- no wave can start its products (tools/pose_trace.py: of a
// 1.37 us layer, 0.50 us were the requests and 0.56 us the 64 MFMAs). Interleaved, the matrix pipe and the memory pipe run side by
// side. Rows past NN read row NN - 1 (results of such rows are never stored), so no select sits between a load and its use.
template <int KN>
struct Pn4Next {
  Pn4A<(KN > 0 ? KN : 2)>* A;
  const float* Wa;
  int si, sk, N;
};
template <int K, int T, bool MASKED = false, int KN = 0>
__device__ __forceinline__ void pn4_layer(const Pn4A<K>& A, int N, const float* __restrict__ bias, const float* sIn, bool relu, const float* sAdd,
                                          const Pn4Mask<T>* M, float* sOut, float* sPart, float* __restrict__ gOut, int i0, int I,
                                          unsigned long long* tp = nullptr, const Pn4Next<KN> nx = Pn4Next<KN>{}) {
#ifdef ACEZ_DIAG
#define PN4_STAMP(i) do { if (tp && threadIdx.x == 0) tp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PN4_STAMP(i) do { } while (0)
#endif
  constexpr int G = T / 4, KH = K / 2;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int rh = w & 1, kh = w >> 1;
  const int blk = lane >> 2, j = lane & 3;
  const int nb = 64 * rh + 4 * blk;            // rows nb .. nb + 3 of this lane's results
  // epilogue operands of the lower-half waves: independent of the products, requested before them
  float bv[4];
  if (kh == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = min(nb + r, N - 1);
      bv[r] = bias ? bias[n] : 0.f;
    }
  }
  PN4_STAMP(0);
  __syncthreads();   // sIn complete; the readers of sOut / sPart of the layer before are done
  PN4_STAMP(1);
  float bq[G][KH];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float* sB = sIn + (size_t)(4 * g + j) * pn4_ld(K) + kh * KH;
    if constexpr (KH % 4 == 0) {
#pragma unroll
      for (int q = 0; q < KH; q += 4) {
        const float4 x = *reinterpret_cast<const float4*>(sB + q);
        bq[g][q] = x.x; bq[g][q + 1] = x.y; bq[g][q + 2] = x.z; bq[g][q + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < KH; ++q) bq[g][q] = sB[q];
    }
  }
  // (the scheduler undoes the block of reads above to shorten their live ranges -- one ds_read_b128 + s_waitcnt lgkmcnt(0) per four
  // MFMAs in the fc layers, one read ahead in the others: 16 LDS round trips of ~130 cycles under 64 MFMAs of 8: nothing may cross here)
  __builtin_amdgcn_sched_barrier(0);
#ifndef PN4_CHAINS
#define PN4_CHAINS 2
#endif
  constexpr int CH = PN4_CHAINS;
  pn_f4 acc[G][CH];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[g][c] = pn_f4{0.f, 0.f, 0.f, 0.f};
  auto accsum = [&](int g, int r) { return CH == 2 ? acc[g][0][r] + acc[g][1][r] : (acc[g][0][r] + acc[g][1][r]) + (acc[g][2 % CH][r] + acc[g][3 % CH][r]); };
  constexpr int KHN = KN / 2;
  const float* pN = nullptr;
  if constexpr (KN > 0) pN = nx.Wa + (size_t)min(64 * rh + lane, nx.N - 1) * nx.si + (size_t)(kh * KHN) * nx.sk;
#pragma unroll
  for (int q = 0; q < KH; ++q) {
#pragma unroll
    for (int g = 0; g < G; ++g)   // two accumulator chains per image group (even / odd k): back-to-back dependent 4x4x1 MFMAs would stall
      acc[g][q % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(A.a[q], bq[g][q], acc[g][q % CH], 0, 0, 0);
    if constexpr (KN > 0) {
      if (q < KHN) nx.A->a[q] = pN[(size_t)q * nx.sk];
      __builtin_amdgcn_sched_group_barrier(0x008, G, 0);    // this step's MFMAs ...
      if (q < KHN) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // ... then one weight request
    }
  }
  if constexpr (KN > 0) {
#pragma unroll
    for (int q = KH; q < KHN; ++q) nx.A->a[q] = pN[(size_t)q * nx.sk];
  }
  PN4_STAMP(2);
  if (kh == 1) {
#pragma unroll
    for (int g = 0; g < G; ++g)
      *reinterpret_cast<float4*>(sPart + (4 * g + j) * PN4_PITCH + nb) =
          make_float4(accsum(g, 0), accsum(g, 1), accsum(g, 2), accsum(g, 3));
  }
  __syncthreads();   // the upper halves are in sPart
  PN4_STAMP(3);
  if (kh == 0 && nb < N) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int img = i0 + 4 * g + j;
      const float4 up = *reinterpret_cast<const float4*>(sPart + (4 * g + j) * PN4_PITCH + nb);
      float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sAdd) ad = *reinterpret_cast<const float4*>(sAdd + (4 * g + j) * PN4_PITCH + nb);
      const float upv[4] = {up.x, up.y, up.z, up.w}, adv[4] = {ad.x, ad.y, ad.z, ad.w};
      float mv[4] = {1.f, 1.f, 1.f, 1.f};
      if constexpr (MASKED) { mv[0] = M->m[g].x; mv[1] = M->m[g].y; mv[2] = M->m[g].z; mv[3] = M->m[g].w; }
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        v[r] = accsum(g, r) + upv[r];
        v[r] += bv[r];
        v[r] += adv[r];
        if (relu) v[r] = fmaxf(v[r], 0.f);
        if (!(mv[r] > 0.f) || img >= I || n >= N) v[r] = 0.f;
      }
      if (nb + 3 < N) {   // (N is 128 or 12: a lane's four rows are all inside or all outside)
        *reinterpret_cast<float4*>(sOut + (4 * g + j) * pn4_ld(N) + nb) = make_float4(v[0], v[1], v[2], v[3]);
        if (img < I) *reinterpret_cast<float4*>(gOut + (size_t)img * N + nb) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

// refined poses of the T images of tile `tile` (forward of the network + compose / orthonormalisation). smem: (12 + 4 * 128) * T floats
template <int T>
__device__ __forceinline__ void pose4_fwd_body(const PoseNetArgs& a, const int tile, float* smem) {
  float* sX = smem;                  // [T][128] x 3 + the partial tile
  float* sY = sX + PN4_PITCH * T;
  float* sZ = sY + PN4_PITCH * T;
  float* sP = sZ + PN4_PITCH * T;
  float* sT = sP + PN4_PITCH * T;          // [T][12]
  const int t = threadIdx.x, i0 = tile * T;
  PN_STAMP(a, tile, 0);
  if (t < 12 * T) {
    const int i = t / 12, k = t % 12;
    sT[i * 12 + k] = (i0 + i < a.I) ? a.T0[(size_t)(i0 + i) * 16 + k] : 0.f;
  }
  const float* P = a.P;
  const float* Wt = a.Wt;   // [4][k][n] transposed copies of conv2, conv3, fc1, fc2: lane = n reads consecutive floats
  Pn4A<12> A12;
  Pn4A<128> Aa, Ab;
  unsigned long long* tp = a.trace ? a.trace + (size_t)tile * 16 + 9 : nullptr;
  pn4_fetch<12>(A12, P + PN_C1_W, 12, 1, 128);
  pn4_fetch<128>(Aa, Wt + 0 * 16384, 1, 128, 128);
  // every 128-deep layer requests the weights of the next one while it multiplies (Pn4Next), into the array the layer before it used
  pn4_layer<12, T>(A12, 128, P + PN_C1_B, sT, true, nullptr, nullptr, sX, sP, a.a1, i0, a.I);                                 // x1 = relu(conv1(T))
  PN_STAMP(a, tile, 1);
  pn4_layer<128, T, false, 128>(Aa, 128, P + PN_C2_B, sX, true, nullptr, nullptr, sY, sP, a.a2, i0, a.I, tp,
                                Pn4Next<128>{&Ab, Wt + 1 * 16384, 1, 128, 128});                                              // x2
  PN_STAMP(a, tile, 2);
  pn4_fetch<12>(A12, P + PN_SKIP_W, 12, 1, 128);
  pn4_layer<128, T, false, 128>(Ab, 128, P + PN_C3_B, sY, true, nullptr, nullptr, sZ, sP, a.a3, i0, a.I, nullptr,
                                Pn4Next<128>{&Aa, Wt + 2 * 16384, 1, 128, 128});                                              // x3
  PN_STAMP(a, tile, 3);
  pn4_layer<12, T>(A12, 128, P + PN_SKIP_B, sT, false, sZ, nullptr, sX, sP, a.r, i0, a.I);                                    // res = head_skip(T) + x3
  PN_STAMP(a, tile, 4);
  pn4_layer<128, T, false, 128>(Aa, 128, P + PN_F1_B, sX, true, nullptr, nullptr, sY, sP, a.f1, i0, a.I, nullptr,
                                Pn4Next<128>{&Ab, Wt + 3 * 16384, 1, 128, 128});                                              // relu(fc1(res))
  PN_STAMP(a, tile, 5);
  pn4_layer<128, T, false, 128>(Ab, 128, P + PN_F2_B, sY, true, nullptr, nullptr, sZ, sP, a.f2, i0, a.I, nullptr,
                                Pn4Next<128>{&Aa, P + PN_F3_W, 128, 1, 12});                                                  // relu(fc2(.))
  PN_STAMP(a, tile, 6);
  float* sD = sX;   // [T][12]
  pn4_layer<128, T>(Aa, 12, P + PN_F3_B, sZ, false, nullptr, nullptr, sD, sP, a.delta, i0, a.I);                              // fc3: the pose update
  PN_STAMP(a, tile, 7);
  __syncthreads();
  if (t < T && i0 + t < a.I) {   // P = T + w * delta and the orthonormalisation, one thread per image
    float Pm[12], o[16];
#pragma unroll
    for (int k = 0; k < 12; ++k) Pm[k] = sT[t * 12 + k] + a.w * sD[t * 12 + k];
    pose_orthonormalise(Pm, a.ortho, o);
    float* dst = a.pose_cur + (size_t)(i0 + t) * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) dst[k] = o[k];
  }
  PN_STAMP(a, tile, 8);
}
template <int T>
constexpr int pose4_fwd_smem_floats() { return (12 + 4 * PN4_PITCH) * T; }

// compose backward + the chain of input gradients for the T images of tile `tile`; dTt: [T][12] gradient wrt their refined poses
// (LDS). sD [T][12], sX, sY, sP [T][128] (sD last: the 16-byte alignment of the big tiles does not depend on T).
// What the chain needs that does not depend on the reduction before it -- the raw pose of the thread's image (T0 and the network's
// update, six 16-byte loads), the first mask and the weights of the first two layers -- requested at the ENTRY of S1: their round trip
// (the update and the masks were written by the forward launch, on another XCD as a rule: ~1.3 us) passes behind the table scan.
template <int T>
struct Pn4BwdPre {
  float4 t0[3], dl[3];
  Pn4A<12> A12;
  Pn4A<128> Aa;
  Pn4Mask<T> Ma;
};
template <int T>
__device__ __forceinline__ void pose4_bwd_prefetch(const PoseNetArgs& a, const int tile, Pn4BwdPre<T>& pre) {
  const int i0 = tile * T, i = min(i0 + (int)(threadIdx.x & (T - 1)), a.I - 1);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    pre.t0[k] = *reinterpret_cast<const float4*>(a.T0 + (size_t)i * 16 + 4 * k);
    pre.dl[k] = *reinterpret_cast<const float4*>(a.delta + (size_t)i * 12 + 4 * k);
  }
  pn4_mask<T>(pre.Ma, a.f2, i0, a.I);
  // dX[k][img] = sum_n W[n][k] dY[n][img]: A(k, n) = W[n * 128 + k] -> si = 1 (lane = k: coalesced), sk = 128
  pn4_fetch<12>(pre.A12, a.P + PN_F3_W, 1, 128, 128);
  pn4_fetch<128>(pre.Aa, a.P + PN_F2_W, 1, 128, 128);
}
template <int T>
__device__ __forceinline__ void pose4_bwd_body(const PoseNetArgs& a, const int tile, Pn4BwdPre<T>& pre, const float* dTt, float* sX, float* sY, float* sP,
                                               float* sD) {
  const int t = threadIdx.x, i0 = tile * T;
  const float* P = a.P;
  Pn4A<12>& A12 = pre.A12;
  Pn4A<128>& Aa = pre.Aa;
  Pn4A<128> Ab;
  Pn4Mask<T>& Ma = pre.Ma;
  Pn4Mask<T> Mb;
  if (t < T) {   // compose backward, one thread per image: gradient wrt the refined pose -> gradient wrt the network's update
    float o[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) o[k] = 0.f;
    const int i = i0 + t;
    if (i < a.I) {
      const float t0[12] = {pre.t0[0].x, pre.t0[0].y, pre.t0[0].z, pre.t0[0].w, pre.t0[1].x, pre.t0[1].y, pre.t0[1].z, pre.t0[1].w,
                            pre.t0[2].x, pre.t0[2].y, pre.t0[2].z, pre.t0[2].w};
      const float dl[12] = {pre.dl[0].x, pre.dl[0].y, pre.dl[0].z, pre.dl[0].w, pre.dl[1].x, pre.dl[1].y, pre.dl[1].z, pre.dl[1].w,
                            pre.dl[2].x, pre.dl[2].y, pre.dl[2].z, pre.dl[2].w};
      float Pm[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) Pm[k] = t0[k] + a.w * dl[k];
      PN_STAMP(a, tile, 11);
      pose_orthonormalise_bwd(Pm, dTt + t * 12, a.ortho, a.w, o);
      PN_STAMP(a, tile, 12);
#pragma unroll
      for (int k = 0; k < 12; ++k) a.ddelta[(size_t)i * 12 + k] = o[k];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) sD[t * 12 + k] = o[k];
  }
  PN_STAMP(a, tile, 2);
  pn4_mask<T>(Mb, a.f1, i0, a.I);
  pn4_layer<12, T, true>(A12, 128, nullptr, sD, false, nullptr, &Ma, sX, sP, a.dz2, i0, a.I);      // through fc3, relu'(fc2 out)
  PN_STAMP(a, tile, 3);
  float m3[T / 2];   // the mask of the element-wise step between the fc1 and conv3 layers (thread t: elements t + 256 u of the [T][128] tile)
#pragma unroll
  for (int u = 0; u < T / 2; ++u) m3[u] = a.a3[(size_t)min(i0 + (t + 256 * u) / 128, a.I - 1) * 128 + (t & 127)];
  pn4_layer<128, T, true, 128>(Aa, 128, nullptr, sX, false, nullptr, &Mb, sY, sP, a.dz1, i0, a.I, nullptr,
                               Pn4Next<128>{&Ab, P + PN_F1_W, 1, 128, 128});                       // through fc2, relu'(fc1 out)
  PN_STAMP(a, tile, 4);
  pn4_mask<T>(Ma, a.a2, i0, a.I);
  pn4_layer<128, T, false, 128>(Ab, 128, nullptr, sY, false, nullptr, nullptr, sX, sP, a.dr, i0, a.I, nullptr,
                                Pn4Next<128>{&Aa, P + PN_C3_W, 1, 128, 128});                      // through fc1: gradient of res
  PN_STAMP(a, tile, 5);
  pn4_mask<T>(Mb, a.a1, i0, a.I);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < T / 2; ++u) {   // x3 = relu(conv3(x2)): its pre-activation gradient is d(res) masked
    const int idx = t + 256 * u, i = idx / 128, k = idx % 128, img = i0 + i;
    float v = sX[i * PN4_PITCH + k];
    if (!(img < a.I && m3[u] > 0.f)) v = 0.f;
    sY[i * PN4_PITCH + k] = v;
    if (img < a.I) a.dzc3[(size_t)img * 128 + k] = v;
  }
  PN_STAMP(a, tile, 6);
  pn4_layer<128, T, true, 128>(Aa, 128, nullptr, sY, false, nullptr, &Ma, sX, sP, a.dzc2, i0, a.I, nullptr,
                               Pn4Next<128>{&Ab, P + PN_C2_W, 1, 128, 128});
  PN_STAMP(a, tile, 7);
  pn4_layer<128, T, true>(Ab, 128, nullptr, sX, false, nullptr, &Mb, sY, sP, a.dzc1, i0, a.I);
  PN_STAMP(a, tile, 8);
}

// S1 on a small tile: per-image reduction of the per-row pose gradients + pose4_bwd_body. LDS carve: hit lists (PS4_HITS entries),
// the [T][12] sums, then one area shared by the hit values and the chain's tiles.
constexpr int PS4_HITS = 512;
template <int T>
constexpr int pose4_s1_smem_bytes() {
  constexpr int chain = (12 + 3 * PN4_PITCH) * T * 4, vals = PS4_HITS * 12 * 4;
  return PS4_HITS * 5 + 16 + T * 12 * 4 + (chain > vals ? chain : vals);
}
template <int T>
__device__ __forceinline__ void pose4_s1_body(const PoseNetArgs& a, const float* row_dT, const int* row_image, const int n, const int tile, char* smem) {
  int* sRow = reinterpret_cast<int*>(smem);
  unsigned char* sRel = reinterpret_cast<unsigned char*>(smem + PS4_HITS * 4);
  int* sCnt = reinterpret_cast<int*>(smem + PS4_HITS * 5);
  float* sDT = reinterpret_cast<float*>(smem + PS4_HITS * 5 + 16);
  float* area = sDT + T * 12;
  PN_STAMP(a, tile, 0);
  Pn4BwdPre<T> pre;
  pose4_bwd_prefetch<T>(a, tile, pre);
  pose_grad_reduce_body<PS4_HITS, T>(row_dT, row_image, n, tile * T, sRow, sRel, reinterpret_cast<float (*)[12]>(area), sCnt, sDT,
                                     a.trace ? a.trace + (size_t)tile * 16 + 9 : nullptr);
  __syncthreads();   // sDT complete; the hit values are dead, their area becomes the chain's tiles
  PN_STAMP(a, tile, 1);
  pose4_bwd_body<T>(a, tile, pre, sDT, area, area + PN4_PITCH * T, area + 2 * PN4_PITCH * T, area + 3 * PN4_PITCH * T);
}

}  // namespace acez
