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
// Roles (W waves, 64 W threads; W = 8 in the forward, 4 in the backward chain): wave w owns output rows 64 (w & 1) + lane and the
// reduction slice k in [K/(W/2) (w >> 1), ...); the W / 2 slices are combined through LDS in a fixed order (((s0 + s1) + s2) + s3) by
// an element-per-thread epilogue that all waves share. Operand A (weights) sits in registers, one value per lane and k, requested one
// layer ahead with coalesced loads: forward from the transposed copies Wt[k][n] (lane = n), backward from W[n][k] itself (lane = k: the
// output index of the transposed product). Operand B (activations [T][K] in LDS) is a broadcast read: lane l needs image l & 3 of its
// group.
//   D layout: lane 4 b + j holds rows 4 b + 0..3 (4 registers) of image j -> one 16-byte LDS write per lane into its slice's partial tile.
//
// What bounds a layer (tools/pose_trace.py, s_memtime stamps of every workgroup): a pose workgroup is alone on its CU and everything it
// runs is a dependent chain. (1) A layer's 64 KiB of fp32 weights enter the CU at 64 B / clock -- 0.43 us -- and all 250 workgroups
// read the same lines at the same time, which is the L2's whole bandwidth as well: the floor of a 128-deep layer. (2) A SIMD with ONE
// wave issues an instruction every ~8.5 cycles whatever the instruction is: a 282-instruction layer of the 4-wave version was 1.0 us;
// with eight waves (two per SIMD, half the loads / products / LDS reads each, one epilogue element per thread) it is 0.8-0.9 us.
#include "pose_kernels.hip"

namespace acez {

// W = waves of the workgroup (template parameter of everything below): W / 2 reduction slices, 64 W threads. The forward (S3) runs with
// 8 waves, the reduce + backward chain (S1) with 4: its launch is shared with the optimiser's 256-thread workgroups (twice the waves to
// dispatch cost more than the shorter layers gave back), and its gradients keep the two-slice summation order.
constexpr int PN4_W_FWD = 8, PN4_W_BWD = 4;

// Row pitch of the 128-wide LDS tiles: 132 floats. A lane's B operands are 16-byte reads of image j = lane & 3: at a pitch of 128 floats
// the four images start in the same bank (4-way conflict on every read of the product loop), at 132 they are 16 bytes apart.
constexpr int PN4_PITCH = 132;
__device__ __forceinline__ constexpr int pn4_ld(int n) { return n == 128 ? PN4_PITCH : n; }

template <int K, int W>
struct Pn4A {
  static_assert(K % (W / 2) == 0, "the reduction must split evenly over the slices");
  float a[K / (W / 2)];
};
// A(n, k) = Wa[n * si + k * sk] for this lane's row n = 64 (w & 1) + lane and this wave's slice of k. Rows past N read row N - 1:
// results of such rows are never stored, and no select sits between a load and its use.
template <int K, int W>
__device__ __forceinline__ void pn4_fetch(Pn4A<K, W>& A, const float* __restrict__ Wa, int si, int sk, int N) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int KP = K / (W / 2);
  const float* p = Wa + (size_t)min(64 * (w & 1) + lane, N - 1) * si + (size_t)((w >> 1) * KP) * sk;
#pragma unroll
  for (int q = 0; q < KP; ++q) A.a[q] = p[(size_t)q * sk];
}

// The relu masks of the backward chain (stored activations of the forward launch, [I][128]) are independent of everything the chain
// computes: pn4_mask requests the values of a thread's epilogue elements a layer ahead of their use. (Requested inside the layer as
// scalar loads under `gMask ? .. : 1.f`, the compiler folded the `> 0` test into the branch that loads: SERIAL load -> s_waitcnt
// vmcnt(0) -> v_cmp round trips before the barrier of each masked layer, each of them also waiting for the weight prefetch.)
template <int T, int W>
struct Pn4Mask {
  float m[2 * T / W];   // the thread's elements t + 64 W u of the [T][128] tile
};
template <int T, int W>
__device__ __forceinline__ void pn4_mask(Pn4Mask<T, W>& M, const float* __restrict__ gMask, int i0, int I) {
  const int t = threadIdx.x;
#pragma unroll
  for (int u = 0; u < 2 * T / W; ++u) M.m[u] = gMask[(size_t)min(i0 + (t + 64 * W * u) / 128, I - 1) * 128 + (t & 127)];
}

// The weights of a LATER layer, requested from inside this layer's product loop (KN != 0): one load after each step's MFMAs, so that
// the matrix pipe and the memory pipe of a wave alternate instead of a block of requests ahead of the layer.
template <int KN, int W>
struct Pn4Next {
  Pn4A<(KN > 0 ? KN : W / 2), W>* A;
  const float* Wa;
  int si, sk, N;
};

// out[img][n] = epi( sum_k A(n, k) in[img][k] ) for the T images of the tile. LDS tiles are IMAGE-major: sIn [T][K], sOut [T][N],
// sAdd [T][128] (the residual input), sPart [W / 2][T][128] (the slices' partial sums), gOut global [I][N]: a lane's B operands of four
// consecutive k are one 16-byte LDS read (broadcast: only four distinct addresses per instruction), and its four result rows one 16-byte
// LDS write. Every B operand of the wave's slice is read BEFORE the first MFMA.
// epi = (+ bias, + sAdd, relu) forward / (mask by the stored activation M > 0, MASKED) backward, element e = t + 512 u of the [T][N]
// tile per thread (W = 8; t + 256 u for W = 4). gOut2 != null (the layer in front of the element-wise relu' of conv3): gOut gets the unmasked value, the value
// masked by M goes on to sOut and gOut2. All 64 W threads must call it.
template <int K, int T, int W, bool MASKED = false, int KN = 0>
__device__ __forceinline__ void pn4_layer(const Pn4A<K, W>& A, int N, const float* __restrict__ bias, const float* sIn, bool relu, const float* sAdd,
                                          const Pn4Mask<T, W>* M, float* sOut, float* sPart, float* __restrict__ gOut, int i0, int I,
                                          const Pn4Next<KN, W> nx = Pn4Next<KN, W>{}, float* __restrict__ gOut2 = nullptr) {
  constexpr int PN4_NT = 64 * W, PN4_KS = W / 2;
  constexpr int G = T / 4, KP = K / PN4_KS, E = 2 * T / W;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int rh = w & 1, ks = w >> 1;
  const int blk = lane >> 2, j = lane & 3;
  const int nb = 64 * rh + 4 * blk;            // rows nb .. nb + 3 of this lane's products
  // this thread's epilogue elements: N = 128: image (t + 64 W u) / 128, row t & 127; N = 12: image t / 12, row t % 12 (t < 12 T)
  const bool wide = N == 128;
  const int en = wide ? (t & 127) : t % 12;
  const float bv = bias ? bias[min(en, N - 1)] : 0.f;   // requested before the products
  __syncthreads();   // sIn complete; the readers of sOut / sPart of the layer before are done
  float bq[G][KP];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float* sB = sIn + (size_t)(4 * g + j) * pn4_ld(K) + ks * KP;
    if constexpr (KP % 4 == 0) {
#pragma unroll
      for (int q = 0; q < KP; q += 4) {
        const float4 x = *reinterpret_cast<const float4*>(sB + q);
        bq[g][q] = x.x; bq[g][q + 1] = x.y; bq[g][q + 2] = x.z; bq[g][q + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < KP; ++q) bq[g][q] = sB[q];
    }
  }
  __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the reads between the MFMAs, each behind its own full wait)
  pn_f4 acc[G][2];
#pragma unroll
  for (int g = 0; g < G; ++g) { acc[g][0] = pn_f4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = pn_f4{0.f, 0.f, 0.f, 0.f}; }
  constexpr int KPN = KN / PN4_KS;
  const float* pN = nullptr;
  if constexpr (KN > 0) pN = nx.Wa + (size_t)min(64 * rh + lane, nx.N - 1) * nx.si + (size_t)(ks * KPN) * nx.sk;
#pragma unroll
  for (int q = 0; q < KP; ++q) {
#pragma unroll
    for (int g = 0; g < G; ++g)   // two accumulator chains per image group (even / odd k)
      acc[g][q & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(A.a[q], bq[g][q], acc[g][q & 1], 0, 0, 0);
    if constexpr (KN > 0) {
      if (q < KPN) nx.A->a[q] = pN[(size_t)q * nx.sk];
      __builtin_amdgcn_sched_group_barrier(0x008, G, 0);                  // this step's MFMAs ...
      if (q < KPN) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // ... then one weight request
    }
  }
  if constexpr (KN > 0) {
#pragma unroll
    for (int q = KP; q < KPN; ++q) nx.A->a[q] = pN[(size_t)q * nx.sk];
  }
#pragma unroll
  for (int g = 0; g < G; ++g)
    *reinterpret_cast<float4*>(sPart + (size_t)ks * (T * PN4_PITCH) + (4 * g + j) * PN4_PITCH + nb) =
        make_float4(acc[g][0][0] + acc[g][1][0], acc[g][0][1] + acc[g][1][1], acc[g][0][2] + acc[g][1][2], acc[g][0][3] + acc[g][1][3]);
  __syncthreads();   // every slice's partial sums are in sPart
  if (wide || t < 12 * T) {
#pragma unroll
    for (int u = 0; u < E; ++u) {
      if (!wide && u > 0) break;
      const int ei = wide ? (t + PN4_NT * u) >> 7 : t / 12;     // image of the element inside the tile
      const int img = i0 + ei;
      const float* pp = sPart + ei * PN4_PITCH + en;
      float v = pp[0];
#pragma unroll
      for (int s = 1; s < PN4_KS; ++s) v += pp[(size_t)s * (T * PN4_PITCH)];
      v += bv;
      if (sAdd) v += sAdd[ei * PN4_PITCH + en];
      if (relu) v = fmaxf(v, 0.f);
      if (img >= I) v = 0.f;
      if (gOut2) {
        if (img < I) gOut[(size_t)img * N + en] = v;
        if (!(M->m[u] > 0.f)) v = 0.f;
        sOut[ei * pn4_ld(N) + en] = v;
        if (img < I) gOut2[(size_t)img * N + en] = v;
      } else {
        if constexpr (MASKED) { if (!(M->m[u] > 0.f)) v = 0.f; }
        sOut[ei * pn4_ld(N) + en] = v;
        if (img < I) gOut[(size_t)img * N + en] = v;
      }
    }
  }
}

// refined poses of the T images of tile `tile` (forward of the network + compose / orthonormalisation)
template <int T>
constexpr int pose4_fwd_smem_floats() { return (12 + (3 + PN4_W_FWD / 2) * PN4_PITCH) * T; }
template <int T>
__device__ __forceinline__ void pose4_fwd_body(const PoseNetArgs& a, const int tile, float* smem) {
  constexpr int W = PN4_W_FWD, PN4_KS = W / 2;
  float* sX = smem;                        // [T][128] x 3, the slices' partial tiles, [T][12]
  float* sY = sX + PN4_PITCH * T;
  float* sZ = sY + PN4_PITCH * T;
  float* sP = sZ + PN4_PITCH * T;
  float* sT = sP + PN4_KS * PN4_PITCH * T;
  const int t = threadIdx.x, i0 = tile * T;
  PN_STAMP(a, tile, 0);
  if (t < 12 * T) {
    const int i = t / 12, k = t % 12;
    sT[i * 12 + k] = (i0 + i < a.I) ? a.T0[(size_t)(i0 + i) * 16 + k] : 0.f;
  }
  const float* P = a.P;
  const float* Wt = a.Wt;   // [4][k][n] transposed copies of conv2, conv3, fc1, fc2: lane = n reads consecutive floats
  Pn4A<12, W> A12;
  Pn4A<128, W> Aa, Ab;
  pn4_fetch<12, W>(A12, P + PN_C1_W, 12, 1, 128);
  pn4_fetch<128, W>(Aa, Wt + 0 * 16384, 1, 128, 128);
  // every 128-deep layer requests the weights of the next one while it multiplies (Pn4Next), into the array the layer before it used
  pn4_layer<12, T, W>(A12, 128, P + PN_C1_B, sT, true, nullptr, nullptr, sX, sP, a.a1, i0, a.I);                                 // x1 = relu(conv1(T))
  PN_STAMP(a, tile, 1);
  pn4_layer<128, T, W, false, 128>(Aa, 128, P + PN_C2_B, sX, true, nullptr, nullptr, sY, sP, a.a2, i0, a.I,
                                Pn4Next<128, W>{&Ab, Wt + 1 * 16384, 1, 128, 128});                                              // x2
  PN_STAMP(a, tile, 2);
  pn4_fetch<12, W>(A12, P + PN_SKIP_W, 12, 1, 128);
  pn4_layer<128, T, W, false, 128>(Ab, 128, P + PN_C3_B, sY, true, nullptr, nullptr, sZ, sP, a.a3, i0, a.I,
                                Pn4Next<128, W>{&Aa, Wt + 2 * 16384, 1, 128, 128});                                              // x3
  PN_STAMP(a, tile, 3);
  pn4_layer<12, T, W>(A12, 128, P + PN_SKIP_B, sT, false, sZ, nullptr, sX, sP, a.r, i0, a.I);                                    // res = head_skip(T) + x3
  PN_STAMP(a, tile, 4);
  pn4_layer<128, T, W, false, 128>(Aa, 128, P + PN_F1_B, sX, true, nullptr, nullptr, sY, sP, a.f1, i0, a.I,
                                Pn4Next<128, W>{&Ab, Wt + 3 * 16384, 1, 128, 128});                                              // relu(fc1(res))
  PN_STAMP(a, tile, 5);
  pn4_layer<128, T, W, false, 128>(Ab, 128, P + PN_F2_B, sY, true, nullptr, nullptr, sZ, sP, a.f2, i0, a.I,
                                Pn4Next<128, W>{&Aa, P + PN_F3_W, 128, 1, 12});                                                  // relu(fc2(.))
  PN_STAMP(a, tile, 6);
  float* sD = sX;   // [T][12]
  pn4_layer<128, T, W>(Aa, 12, P + PN_F3_B, sZ, false, nullptr, nullptr, sD, sP, a.delta, i0, a.I);                              // fc3: the pose update
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

// What the chain needs that does not depend on the reduction before it -- the raw pose of the thread's image (T0 and the network's
// update, six 16-byte loads), the first mask and the weights of the first two layers -- requested at the ENTRY of S1: their round trip
// (the update and the masks were written by the forward launch, on another XCD as a rule: ~1.3 us) passes behind the table scan.
template <int T>
struct Pn4BwdPre {
  float4 t0[3], dl[3];
  Pn4A<12, PN4_W_BWD> A12;
  Pn4A<128, PN4_W_BWD> Aa;
  Pn4Mask<T, PN4_W_BWD> Ma;
};
template <int T>
__device__ __forceinline__ void pose4_bwd_prefetch(const PoseNetArgs& a, const int tile, Pn4BwdPre<T>& pre) {
  constexpr int W = PN4_W_BWD;
  const int i0 = tile * T, i = min(i0 + (int)(threadIdx.x & (T - 1)), a.I - 1);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    pre.t0[k] = *reinterpret_cast<const float4*>(a.T0 + (size_t)i * 16 + 4 * k);
    pre.dl[k] = *reinterpret_cast<const float4*>(a.delta + (size_t)i * 12 + 4 * k);
  }
  pn4_mask<T, W>(pre.Ma, a.f2, i0, a.I);
  // dX[k][img] = sum_n W[n][k] dY[n][img]: A(k, n) = W[n * 128 + k] -> si = 1 (lane = k: coalesced), sk = 128
  pn4_fetch<12, W>(pre.A12, a.P + PN_F3_W, 1, 128, 128);
  pn4_fetch<128, W>(pre.Aa, a.P + PN_F2_W, 1, 128, 128);
}
// compose backward + the chain of input gradients for the T images of tile `tile`; dTt: [T][12] gradient wrt their refined poses
// (LDS). sX, sY [T][128], sP [W / 2][T][128], sD [T][12] (sD last: the 16-byte alignment of the big tiles does not depend on T).
template <int T>
__device__ __forceinline__ void pose4_bwd_body(const PoseNetArgs& a, const int tile, Pn4BwdPre<T>& pre, const float* dTt, float* sX, float* sY, float* sP,
                                               float* sD) {
  constexpr int W = PN4_W_BWD;
  const int t = threadIdx.x, i0 = tile * T;
  const float* P = a.P;
  Pn4A<12, W>& A12 = pre.A12;
  Pn4A<128, W>& Aa = pre.Aa;
  Pn4A<128, W> Ab;
  Pn4Mask<T, W>& Ma = pre.Ma;
  Pn4Mask<T, W> Mb, M3;
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
  // every later mask is requested one layer ahead of its use
  pn4_mask<T, W>(Mb, a.f1, i0, a.I);
  pn4_layer<12, T, W, true>(A12, 128, nullptr, sD, false, nullptr, &Ma, sX, sP, a.dz2, i0, a.I);      // through fc3, relu'(fc2 out)
  PN_STAMP(a, tile, 3);
  pn4_mask<T, W>(M3, a.a3, i0, a.I);
  pn4_layer<128, T, W, true, 128>(Aa, 128, nullptr, sX, false, nullptr, &Mb, sY, sP, a.dz1, i0, a.I,
                               Pn4Next<128, W>{&Ab, P + PN_F1_W, 1, 128, 128});                       // through fc2, relu'(fc1 out)
  PN_STAMP(a, tile, 4);
  pn4_mask<T, W>(Ma, a.a2, i0, a.I);
  // through fc1: the gradient of res (a.dr), and -- x3 = relu(conv3(x2)) -- masked by x3 the gradient of conv3's output (a.dzc3, sX)
  pn4_layer<128, T, W, false, 128>(Ab, 128, nullptr, sY, false, nullptr, &M3, sX, sP, a.dr, i0, a.I,
                                Pn4Next<128, W>{&Aa, P + PN_C3_W, 1, 128, 128}, a.dzc3);
  PN_STAMP(a, tile, 5);
  pn4_mask<T, W>(Mb, a.a1, i0, a.I);
  PN_STAMP(a, tile, 6);
  pn4_layer<128, T, W, true, 128>(Aa, 128, nullptr, sX, false, nullptr, &Ma, sY, sP, a.dzc2, i0, a.I,
                               Pn4Next<128, W>{&Ab, P + PN_C2_W, 1, 128, 128});
  PN_STAMP(a, tile, 7);
  pn4_layer<128, T, W, true>(Ab, 128, nullptr, sY, false, nullptr, &Mb, sX, sP, a.dzc1, i0, a.I);
  PN_STAMP(a, tile, 8);
}

// S1 on a small tile: per-image reduction of the per-row pose gradients + pose4_bwd_body. LDS carve: hit lists (PS4_HITS entries: rows,
// tags, compacted tags), the lists' counts, the [T][12] sums, then one area shared by the hit values and the chain's tiles.
constexpr int PS4_HITS = 512;
template <int T>
constexpr int pose4_s1_smem_bytes() {
  constexpr int chain = (12 + (2 + PN4_W_BWD / 2) * PN4_PITCH) * T * 4, vals = PS4_HITS * 12 * 4;
  return PS4_HITS * 6 + 4 * PN4_W_BWD + T * 12 * 4 + (chain > vals ? chain : vals);
}
template <int T>
__device__ __forceinline__ void pose4_s1_body(const PoseNetArgs& a, const float* row_dT, const int* row_image, const int n, const int tile, char* smem) {
  int* sRow = reinterpret_cast<int*>(smem);
  unsigned char* sRel = reinterpret_cast<unsigned char*>(smem + PS4_HITS * 4);
  unsigned char* sTag = reinterpret_cast<unsigned char*>(smem + PS4_HITS * 5);
  int* sCnt = reinterpret_cast<int*>(smem + PS4_HITS * 6);
  float* sDT = reinterpret_cast<float*>(smem + PS4_HITS * 6 + 4 * PN4_W_BWD);
  float* area = sDT + T * 12;
  PN_STAMP(a, tile, 0);
  Pn4BwdPre<T> pre;
  pose4_bwd_prefetch<T>(a, tile, pre);
  pose_grad_reduce_body<PS4_HITS, T, PN4_W_BWD>(row_dT, row_image, n, tile * T, sRow, sRel, sTag, reinterpret_cast<float (*)[12]>(area), sCnt, sDT,
                                            a.trace ? a.trace + (size_t)tile * 16 + 9 : nullptr);
  __syncthreads();   // sDT complete; the hit values are dead, their area becomes the chain's tiles
  PN_STAMP(a, tile, 1);
  pose4_bwd_body<T>(a, tile, pre, sDT, area, area + PN4_PITCH * T, area + 2 * PN4_PITCH * T, area + (2 + PN4_W_BWD / 2) * PN4_PITCH * T);
}

}  // namespace acez
