// encoder_api.hip -- the ACE feature encoder (ace_network.py:14-59) on gfx950 and its C ABI (include/acez.h).
//
// the encoder is the step right before both hot paths (it fills the training buffer and it
// produces the features the head turns into scene coordinates at registration time).
//
// Data layout: activations NHWC 16-bit ([frame][y][x][channel]; a pixel's channels are contiguous, so a pixel is a "row" of
// an implicit GEMM and the final [F*h*w][512] tensor is exactly the row layout of the training buffer / acez_head_forward).
// Weights: 16-bit [Co][Kp], k = (ky*3 + kx) * Ci + ci, Kp = K rounded up to 64 (zero padded).
// Every kernel is instantiated on the element trait of gemm_common.h: EltBf16 (v_mfma_f32_*_bf16) and EltF16 (v_mfma_f32_*_f16: the
// operand format the reference's autocast runs this network in, ace_trainer.py:366-367, register_mapping.py:209-210); fp32 accumulation,
// one rounding per layer output in both. The context's compute_dtype selects the instantiation (acez_encoder_create).
//
//   conv1 + conv2           conv12p_kernel: both layers in one launch, the conv1 map never leaves LDS
//   3 x 3, stride 1         conv3x3r_kernel: 256 x 256 tiles, the input kept as an LDS patch (85 % of the encoder's FLOPs)
//   every other layer       implicit GEMM Out[p][co] = act(sum_k In[pix(p, tap(k))][ci(k)] * W[co][k] + b): convgemm512_kernel
//                           (256 x 256 tiles), convgemm256_kernel (256 x 128), convgemm_kernel (80-row x NT-column tiles, 4 multiplier
//                           waves + 4 loader waves, same structure as rowgemm80 in head_kernels.hip), chosen by size. 4-slot LDS-DMA
//                           ring of 64-wide K stages. The im2col never exists in memory: a loader lane computes, per stage, the
//                           source address of its 16-byte chunk (8 input channels of one tap of one pixel) or points at a zero page
//                           for the padding border / K padding / rows past the end.
#include <algorithm>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "../../include/acez.h"
#include "acez_common.h"
#include "gemm_common.h"
#include "conv_launch.h"

namespace acez {


// ---------------------------------------------------------------------------------------------------
// conv12p: conv1 (1 -> 32, stride 1) and conv2 (32 -> 64, stride 2) fused, persistent over 4 x 32 output tiles of conv2 and
// software-pipelined across tiles. As separate kernels these two layers cost 27 % of the encoder's time for 5 % of its FLOPs:
// conv1's 32-channel map (19.7 MB per 480 x 640 frame) is written and read back, and conv2's 3 x 3 taps re-read it nine times from
// L2 into a GEMM that is only 64 columns wide. Here a workgroup keeps the conv1 patch of its tile in LDS:
//   1. image patch 11 x 67 (grey, fp32 as in memory, by LDS-DMA; rounded to 16 bits where conv1 gathers its taps);
//   2. conv1 on the matrix cores: 32-pixel fragments of the 9 x 65 patch, B = the 9 taps gathered from the image patch (K = 16),
//      A = conv1's weights (one register quad), bias + ReLU + 16-bit -> conv1 patch [2 column-parity planes][9][33][32 ch] in LDS
//      (zero outside the image: that is conv2's padding);
//   3. conv2: wave w owns output row w of the tile (32 pixels x 64 channels); its B fragments are read straight from the patch
//      (tap (ky, kx) of output x = plane kx & 1, column x + (kx >> 1): unit stride, swizzled 16-byte chunks), its A fragments (all
//      of conv2's 64 x 288 weights) live in 144 registers for the whole kernel;
//   4. bias + ReLU + 16-bit through a wave-private staging row, 4 KiB contiguous store per output row.
// The two layers of DIFFERENT tiles run beside each other: waves 4 .. 7 compute conv1 of tile i + 1 into one of two LDS patches while
// waves 0 .. TR-1 run conv2 of tile i from the other (and stage the image patch of tile i + 2); waves w and w + 4 share a SIMD, so every
// SIMD has one MFMA-bound and one VALU-bound wave. One s_barrier per tile. (Round 1's phase-by-phase kernel on 8 x 32 tiles -- 7.1 us per
// tile for 1.1 us of MFMA time, 545 us per 64 frames against 352 -- is in the git history; this kernel's output is bit-identical to it.)
// ---------------------------------------------------------------------------------------------------
struct Conv12Args {
  const float* img;        // [F][H][W] fp32
  const uint16_t* w1;      // 16-bit [32][16]: k = tap (9 used)
  const float* b1;         // [32]
  const uint16_t* w2;      // 16-bit [64][Kp2], k = tap * 32 + ci
  const float* b2;         // [64]
  uint16_t* out;           // NHWC 16-bit [F][H2][W2][64]
  int F, H, W, H2, W2, Kp2, tiles_y, tiles_x, n_tiles;
  const float* zero;       // >= 4 bytes of zeros: source of the image-patch DMA outside the image
};

constexpr int C12_IMG_PITCH = 68;

// 16-byte chunk swizzle of conv12p's conv1 patch (a pixel = 32 channels = four chunks; q = column index inside a parity plane). conv2's
// B-fragment reads take 16 consecutive q with one chunk index: conflict free iff the swizzle differs between q, q + 4, q + 8, q + 12;
// conv1's epilogue writes 8 consecutive pixels = 4 consecutive q x 2 planes per lane group: conflict poor iff it also differs between
// q .. q + 3. (q >> 2) & 3 (round 1) does the first only -- the writes were 4-way conflicts, 180 of the kernel's 573 us (ablation, round
// 5); ((q >> 2) + q) & 3 does both.
__device__ __forceinline__ int c12p_swz(int q) { return ((q >> 2) + q) & 3; }

__device__ __forceinline__ void wait_vmcnt_dyn_c12(int n) {   // s_waitcnt vmcnt(n), n = 0 .. 4 wave-uniform
  switch (n) {
    case 1: ACEZ_VMCNT(1); break;
    case 2: ACEZ_VMCNT(2); break;
    case 3: ACEZ_VMCNT(3); break;
    case 4: ACEZ_VMCNT(4); break;
    default: ACEZ_VMCNT(0); break;
  }
}
#ifndef C12_ABL
#define C12_ABL 0   // timing-only ablation of conv12p (tools/c12_variants.sh): 1 = no conv1, 2 = no conv2, 4 = no output stores, 8 = no patch writes
#endif
template <class E, int TR>
__global__ __launch_bounds__(512) void conv12p_kernel(Conv12Args a) {
  typedef typename E::frag frag;
  constexpr int PR = 2 * TR + 1;                 // conv1 patch rows
  constexpr int IMG_N = (PR + 2) * C12_IMG_PITCH;   // image patch: PR + 2 rows of 67 (+ 1 pad) grey values
  constexpr int PLANE = PR * 33 * 32;            // elements per column-parity plane of a conv1 patch
  constexpr int NF = (PR * 65 + 31) / 32;        // 32-pixel conv1 fragments per tile
  static_assert(IMG_N <= 3 * 256, "three image entries per staging thread");
  __shared__ __attribute__((aligned(16))) float s_img[2][3 * 256];   // fp32 as in memory (LDS-DMA); rounded to bf16 where conv1 gathers its taps
  __shared__ __attribute__((aligned(16))) uint16_t s_patch[2][2 * PLANE];
  __shared__ __attribute__((aligned(16))) uint16_t s_out[TR * 32 * 64];
  __shared__ __attribute__((aligned(16))) float s_bias[32 + 64];   // b1 | b2
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = l & 31, fh = l >> 5;
  if (t < 96) s_bias[t] = t < 32 ? a.b1[t] : a.b2[t - 32];   // visible after the first barrier
  const int tiles_y = (a.H2 + TR - 1) / TR, tiles_x = (a.W2 + 31) / 32, tpf = tiles_y * tiles_x;
  const int n_tiles = a.F * tpf;
  const int K = ((int)blockIdx.x < n_tiles) ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // tiles of this workgroup
  auto tile_of = [&](int k) { return (int)blockIdx.x + k * (int)gridDim.x; };

  if (w < 4) {
    // ------------------------------------------------------------------ conv2 waves (w < TR multiply; all four stage the image patches)
    frag a2[9][2][2];
    if (w < TR) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            a2[tap][kk][i] = *reinterpret_cast<const frag*>(a.w2 + (size_t)(i * 32 + fr) * a.Kp2 + tap * 32 + kk * 16 + 8 * fh);
    }
    // image patch staging by LDS-DMA, one dword per lane: entries t, t + 256, t + 512 of the [PR + 2][68] patch (coordinates are tile
    // independent); outside the image (and past the patch) the source is a zero word. No registers, no conversion here, and the
    // transfers are OLDER than this iteration's output stores, so a counted wait certifies them without draining the stores.
    int epy[3], epx[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = t + 256 * i;
      epy[i] = min(e, IMG_N - 1) / C12_IMG_PITCH;
      epx[i] = (e < IMG_N) ? e - epy[i] * C12_IMG_PITCH : 67;   // 67 = the pad column: never valid
    }
    auto stage_img = [&](int k) {   // tile k of this workgroup -> s_img[k & 1]
      const int tl = tile_of(k);
      const int f = tl / tpf, r = tl - f * tpf;
      const int ty = r / tiles_x, tx = r - ty * tiles_x;
      const float* base = a.img + (size_t)f * a.H * a.W;
      float* dst = s_img[k & 1] + w * 64;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int iy = 2 * TR * ty - 2 + epy[i], ix = 64 * tx - 2 + epx[i];
        const bool ok = epx[i] < 67 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const float* g = ok ? base + (size_t)iy * a.W + ix : a.zero;
        __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(dst + 256 * i), 4, 0, 0);
      }
    };
    if (0 < K) stage_img(0);
    for (int j = -2; j < K; ++j) {
      // ---- image patch of tile j + 2 -> s_img[j & 1] (read by conv1 of tile j, one iteration ago); tile 0's went out above
      int n_stores = 0;
      if (j + 2 < K && j + 2 > 0) stage_img(j + 2);
      // ---- conv2 of tile j: output row w, pixels x = fr, channels 2 x 32
      if (j >= 0 && w < TR && !(C12_ABL & 2)) {
        const int tile = tile_of(j);
        const int f = tile / tpf, r = tile - f * tpf;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const uint16_t* sp = s_patch[j & 1];
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int q = fr + (kx >> 1);
            const uint16_t* src = sp + (kx & 1) * PLANE + ((2 * w + ky) * 33 + q) * 32;
            const int sw = c12p_swz(q);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
              const frag b = *reinterpret_cast<const frag*>(src + (((kk * 2 + fh) ^ sw) << 3));
#pragma unroll
              for (int i = 0; i < 2; ++i) acc[i] = E::mfma32(a2[ky * 3 + kx][kk][i], b, acc[i]);
            }
          }
        // bias + ReLU -> wave-private staging row [32 px][64 ch] -> 4 KiB contiguous store
        uint16_t* so = s_out + w * (32 * 64);
        float4 b2v[2][4];   // (all eight LDS reads in flight before the first is used: as eight read-wait pairs they were eight serial round trips per tile)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) b2v[i][g] = *reinterpret_cast<const float4*>(s_bias + 32 + i * 32 + 8 * g + 4 * fh);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ch = i * 32 + 8 * g + 4 * fh;
            const float4 b = b2v[i][g];
            const uint2 y = E::pk4(fmaxf(acc[i][4 * g + 0] + b.x, 0.f), fmaxf(acc[i][4 * g + 1] + b.y, 0.f), fmaxf(acc[i][4 * g + 2] + b.z, 0.f),
                                  fmaxf(acc[i][4 * g + 3] + b.w, 0.f));
            *reinterpret_cast<uint2*>(so + fr * 64 + ((((ch >> 3) ^ (fr & 7)) << 3) | (ch & 7))) = y;
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int oy = TR * ty + w, ox0 = 32 * tx;
        if (oy < a.H2 && !(C12_ABL & 4)) {
          n_stores = min(4, max(0, (a.W2 - ox0 + 7) >> 3));   // store instructions with at least one active lane (the others are branched over)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int pxl = it * 8 + (l >> 3), chk = l & 7;
            const int ox = ox0 + pxl;
            if (ox < a.W2)
              *reinterpret_cast<uint4*>(a.out + (((size_t)f * a.H2 + oy) * a.W2 + ox) * 64 + chk * 8) =
                  *reinterpret_cast<const uint4*>(so + pxl * 64 + ((chk ^ (pxl & 7)) << 3));
          }
        }
      }
      // the image patch requested at the top of this iteration must have landed before conv1 reads it in the next one; it is older than
      // this tile's output stores, which may stay in flight (in-order completion). Raw barrier: __syncthreads() would drain them.
      wait_vmcnt_dyn_c12(n_stores);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    // ------------------------------------------------------------------ conv1 waves: tile j + 1 while the others run conv2 of tile j
    const int lw = w - 4;
    const frag a1 = *reinterpret_cast<const frag*>(a.w1 + fr * 16 + 8 * fh);
    // this lane's sixteen conv1 bias values, in registers for the whole kernel (read from LDS inside the fragment loop each of the four
    // reads was followed by a full lgkmcnt(0) wait: four serial LDS round trips per 32-pixel fragment -- found in the ISA, round 5)
    float4 b1v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) b1v[g] = *reinterpret_cast<const float4*>(a.b1 + 8 * g + 4 * fh);
    // per-lane constants of this wave's fragments fg = lw, lw + 4, ... (tile independent): patch pixel p = fg * 32 + fr -> (py, px), the
    // offset of its taps in the image patch and of its 64-byte record in the conv1 patch (the division by 65 and the address arithmetic
    // ran once per fragment and tile)
    constexpr int NFW = (NF + 3) / 4;
    int f_py[NFW], f_px[NFW], f_ip[NFW], f_dst[NFW], f_sw[NFW];
#pragma unroll
    for (int u = 0; u < NFW; ++u) {
      const int p = (lw + 4 * u) * 32 + fr;
      const int py = min(p / 65, PR - 1), px = p - (p / 65) * 65, q = px >> 1;
      f_py[u] = (lw + 4 * u < NF && p < PR * 65) ? py : -1;   // -1: no such pixel (nothing is written)
      f_px[u] = px;
      f_ip[u] = py * C12_IMG_PITCH + px;
      f_dst[u] = (px & 1) * PLANE + (py * 33 + q) * 32 + 4 * fh;
      f_sw[u] = c12p_swz(q);
    }
    for (int j = -2; j < K; ++j) {
      const int c = j + 1;
      if (c >= 0 && c < K && !(C12_ABL & 1)) {
        const int tile = tile_of(c);
        const int f = tile / tpf, r = tile - f * tpf;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        (void)f;
        const float* si = s_img[c & 1];
        uint16_t* sp = s_patch[c & 1];
        const int cy0 = 2 * TR * ty - 1, cx0 = 64 * tx - 1;          // conv1 pixel of patch position (0, 0)
        // a tile whose whole patch lies inside the image (all but the border tiles) needs no zeroing of outside pixels
        const bool interior = cy0 >= 0 && cy0 + PR <= a.H && cx0 >= 0 && cx0 + 65 <= a.W;
        auto fragment = [&](int u, auto chk) {
          constexpr bool CHECK = decltype(chk)::value;
          const float* ip = si + f_ip[u];   // taps: ip[ky * 68 + kx], rounded to 16 bits here (round to nearest even)
          float tp[9];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) tp[ky * 3 + kx] = ip[ky * C12_IMG_PITCH + kx];
          uint32_t bw[4];
          if (fh == 0) {
            bw[0] = E::pk2(tp[0], tp[1]); bw[1] = E::pk2(tp[2], tp[3]); bw[2] = E::pk2(tp[4], tp[5]); bw[3] = E::pk2(tp[6], tp[7]);
          } else {
            bw[0] = E::pk2(tp[8], 0.f); bw[1] = 0u; bw[2] = 0u; bw[3] = 0u;
          }
          const uint4 bq = make_uint4(bw[0], bw[1], bw[2], bw[3]);
          f32x16 c1;
#pragma unroll
          for (int q = 0; q < 16; ++q) c1[q] = 0.f;
          c1 = E::mfma32(a1, __builtin_bit_cast(frag, bq), c1);
          // conv1 pixel (cy, cx) of this lane; outside the image the map is ZERO (conv2's padding)
          bool inside = true;
          if (CHECK) {
            const int cy = cy0 + f_py[u], cx = cx0 + f_px[u];
            inside = cy >= 0 && cy < a.H && cx >= 0 && cx < a.W;
          }
          if (f_py[u] >= 0 && !(C12_ABL & 8)) {
            uint16_t* dst = sp + f_dst[u];
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // channels 8g + 4 fh .. +3 = half of logical chunk g
              const float4 b = b1v[g];
              float v0 = fmaxf(c1[4 * g + 0] + b.x, 0.f), v1 = fmaxf(c1[4 * g + 1] + b.y, 0.f);
              float v2 = fmaxf(c1[4 * g + 2] + b.z, 0.f), v3 = fmaxf(c1[4 * g + 3] + b.w, 0.f);
              if (CHECK && !inside) v0 = v1 = v2 = v3 = 0.f;
              *reinterpret_cast<uint2*>(dst + ((g ^ f_sw[u]) << 3)) = E::pk4(v0, v1, v2, v3);
            }
          }
        };
        if (interior) {
#pragma unroll
          for (int u = 0; u < NFW; ++u)
            if (lw + 4 * u < NF) fragment(u, std::integral_constant<bool, false>{});
        } else {
#pragma unroll
          for (int u = 0; u < NFW; ++u)
            if (lw + 4 * u < NF) fragment(u, std::integral_constant<bool, true>{});
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Augmented views (acez_buffer_warp_views, include/acez.h): the batched affine warp in front of the encoder when the buffer is filled with
// augmentation (dataset.py:283-343). HBM-bound by construction: 4 B read (gathered, cache-friendly: a rotation of a few degrees) + 4 B
// written per output pixel; the framework version moved an 8 B sampling-grid entry three times per pixel on top.
// Arithmetic follows ATen's grid sampler (GridSampler.h): unnormalise ((g + 1) * size - 1) / 2, reflect about -0.5 / size - 0.5, clip,
// four taps with bounds checks; the mask is "the zero-padded lookup into an all-ones image is positive" = source coordinate in (-1, size).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float warp_reflect(float x, int size) {   // reflect_coordinates(x, -1, 2 size - 1) then clip_coordinates
  const float mn = -0.5f, span = (float)size;
  x = fabsf(x - mn);
  const float extra = fmodf(x, span);
  const int flips = (int)floorf(x / span);
  x = (flips & 1) ? span - extra + mn : extra + mn;
  return fminf((float)(size - 1), fmaxf(x, 0.f));
}
__device__ __forceinline__ float warp_jitter(float v, float br, float ct, float m) {   // ColorJitter on the de-normalised grey value
  float g = fminf(fmaxf((v * 0.25f + 0.4f) * br, 0.f), 1.f);
  g = fminf(fmaxf((g - m) * ct + m, 0.f), 1.f);
  return (g - 0.4f) / 0.25f;
}
// mean over the frame of clamp((v * 0.25 + 0.4) * brightness, 0, 1): torchvision's adjust_contrast blends with the mean of the image it is
// given (the brightness-adjusted one). One workgroup per view, fixed summation order.
__global__ __launch_bounds__(1024) void warp_mean_kernel(const float* __restrict__ images, const int32_t* __restrict__ index, const float* __restrict__ jitter,
                                                         int hw, float* __restrict__ out_mean) {
  __shared__ float part[16];
  const int v = blockIdx.x, t = threadIdx.x;
  const float* img = images + (size_t)index[v] * hw;
  const float br = jitter[2 * v];
  float acc = 0.f;
  for (int i = t; i < hw; i += 1024) acc += fminf(fmaxf((img[i] * 0.25f + 0.4f) * br, 0.f), 1.f);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if ((t & 63) == 0) part[t >> 6] = acc;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += part[i];
    out_mean[v] = s / (float)hw;
  }
}
__global__ __launch_bounds__(256) void warp_views_kernel(const float* __restrict__ images, const int32_t* __restrict__ index, const float* __restrict__ theta,
                                                         const float* __restrict__ jitter, const float* __restrict__ mean, int H, int W, int hs, int ws,
                                                         float* __restrict__ out) {
  const int v = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hs * ws) return;
  const int y = p / ws, x = p - y * ws;
  const float* th = theta + 6 * v;
  const float xn = (2.f * x + 1.f) / ws - 1.f, yn = (2.f * y + 1.f) / hs - 1.f;     // affine_grid's base grid, align_corners = False
  const float gx = xn * th[0] + yn * th[1] + th[2], gy = xn * th[3] + yn * th[4] + th[5];
  const float ix = warp_reflect(((gx + 1.f) * W - 1.f) * 0.5f, W), iy = warp_reflect(((gy + 1.f) * H - 1.f) * 0.5f, H);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx, wy1 = iy - fy, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy;
  const float* img = images + (size_t)index[v] * H * W;
  float br = 1.f, ct = 1.f, m = 0.f;
  const bool jit = jitter != nullptr;
  if (jit) { br = jitter[2 * v]; ct = jitter[2 * v + 1]; m = mean[v]; }
  auto tap = [&](int yy, int xx) -> float {
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) return 0.f;
    const float val = img[(size_t)yy * W + xx];
    return jit ? warp_jitter(val, br, ct, m) : val;
  };
  out[((size_t)v * hs + y) * ws + x] = tap(y0, x0) * (wx0 * wy0) + tap(y0, x1) * (wx1 * wy0) + tap(y1, x0) * (wx0 * wy1) + tap(y1, x1) * (wx1 * wy1);
}
// the validity mask at feature resolution: cell (my, mx) reads view pixel (floor(my * hs / map_h), floor(mx * ws / map_w)) (the nearest-
// neighbour resize, ace_trainer.py:373-374), whose source coordinate must lie inside (-1, W) x (-1, H)
__global__ __launch_bounds__(256) void warp_mask_kernel(const float* __restrict__ theta, int H, int W, int hs, int ws, int mh, int mw, uint8_t* __restrict__ mask) {
  const int v = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= mh * mw) return;
  const int my = c / mw, mx = c - my * mw;
  const float sy = (float)hs / (float)mh, sx = (float)ws / (float)mw;
  const int y = min((int)floorf(my * sy), hs - 1), x = min((int)floorf(mx * sx), ws - 1);
  const float* th = theta + 6 * v;
  const float xn = (2.f * x + 1.f) / ws - 1.f, yn = (2.f * y + 1.f) / hs - 1.f;
  const float gx = xn * th[0] + yn * th[1] + th[2], gy = xn * th[3] + yn * th[4] + th[5];
  const float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
  mask[(size_t)v * mh * mw + c] = (ix > -1.f && ix < (float)W && iy > -1.f && iy < (float)H) ? 1 : 0;
}

// [80][64] staging tile of the 64-column variant: chunk index XOR row & 7
__device__ __forceinline__ int st_off64(int row, int col) { return row * 64 + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7)); }

template <class E, int NT, bool RELU, bool HAS_ADD>
__global__ __launch_bounds__(512) void convgemm_kernel(ConvGemmArgs a) {
  typedef typename E::frag frag;
  static_assert(NT == 64 || NT == 128, "column tile");
  static_assert(!(HAS_ADD && NT == 64), "the residual epilogue exists for 128-column tiles only");
  constexpr int CF = NT / 64;                 // 16-column fragments per multiplier wave
  constexpr int WI = NT / 32;                 // W DMA instructions per loader and stage (8 rows each)
  constexpr int IPS = WI + 3;                 // DMA instructions per loader and stage
  constexpr int STAGE = (NT + 96) * 64;       // elements per ring slot
  __shared__ __attribute__((aligned(16))) uint16_t smem[4 * STAGE + 80 * NT];
  uint16_t* const stO = smem + 4 * STAGE;     // `add` in / output tile
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = a.M, Co = a.Co, Kp = a.Kp;
  const int ntiles = Co / NT;
  const int mtiles = (M + 79) / 80;
  const int per_xcd = (mtiles + 7) >> 3;
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + jx / ntiles;   // the column tiles of a row tile share an XCD (and its L2)
  if (mt >= mtiles) return;
  const int n0 = (jx % ntiles) * NT, m0 = mt * 80;
  const int KT = Kp >> 6;

  if (w >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - 4;
    if (HAS_ADD) {
      // residual / skip tile -> staging (oldest DMA of this wave: complete before any stage it could be confused with)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int row = (lw * 5 + j) * 4 + (l >> 4);
        const uint16_t* g = a.add + (size_t)min(m0 + row, M - 1) * Co + n0 + (((l & 15) ^ (row & 15)) << 3);
        __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(stO + (lw * 5 + j) * 4 * 128), 16, 0, 0);
      }
    }
    const uint16_t* gW[WI];
#pragma unroll
    for (int j = 0; j < WI; ++j) {
      const int row = (lw * WI + j) * 8 + (l >> 3);
      gW[j] = a.W + (size_t)(n0 + row) * Kp + ((l & 7) ^ ((row >> 1) & 7)) * 8;
    }
    // this lane's three rows of the In tile: output pixel -> top-left input pixel of its receptive field
    const uint16_t* ibase[3];
    int iy0[3], ix0[3], kc[3];
    bool pv[3];
    const int hw = a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int row = (lw * 3 + j) * 8 + (l >> 3);
      const int p = m0 + row;
      pv[j] = row < 80 && p < M;
      const int pp = pv[j] ? p : 0;
      const int f = pp / hw, r = pp - f * hw;
      const int y = r / a.Wo, x = r - y * a.Wo;
      iy0[j] = y * a.stride - a.pad;
      ix0[j] = x * a.stride - a.pad;
      ibase[j] = a.In + (size_t)f * a.Hi * a.Wi * a.Ci;
      kc[j] = ((l & 7) ^ ((row >> 1) & 7)) * 8;   // logical K offset of this lane's chunk inside a stage
    }
    const uint16_t* zp = a.zeros + (l & 7) * 8;
    auto issue = [&](int kt) {
      uint16_t* slot = smem + (kt & 3) * STAGE;
#pragma unroll
      for (int j = 0; j < WI; ++j)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(gW[j] + kt * 64), (lvoid_t*)(slot + (lw * WI + j) * 8 * 64), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int k0 = kt * 64 + kc[j];
        const int tap = k0 >> a.ci_shift, ci = k0 & (a.Ci - 1);
        const int ky = (a.ksize == 3) ? (tap * 11) >> 5 : 0;   // tap / 3 for tap < 12
        const int kx = tap - 3 * ky;
        const int iy = iy0[j] + ky, ix = ix0[j] + kx;
        const bool ok = pv[j] && k0 < a.K && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
        const uint16_t* g = ok ? ibase[j] + (((size_t)iy * a.Wi + ix) << a.ci_shift) + ci : zp;
        __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(slot + NT * 64 + (lw * 3 + j) * 8 * 64), 16, 0, 0);
      }
    };
    for (int kt = 0; kt < 4 && kt < KT; ++kt) issue(kt);
    for (int kt = 0; kt < KT; ++kt) {
      // stages issued so far: 0..3 at kt = 0, 0..kt+2 afterwards (in-order completion)
      const int later = (kt == 0) ? min(3, KT - 1) : min(2, KT - 1 - kt);
      if (later >= 3) ACEZ_VMCNT_C(3 * IPS);
      else if (later == 2) ACEZ_VMCNT_C(2 * IPS);
      else if (later == 1) ACEZ_VMCNT_C(IPS);
      else ACEZ_VMCNT(0);
      __builtin_amdgcn_s_barrier();   // stage kt has landed; the multipliers are done with stage kt - 1
      if (kt >= 1 && kt + 3 < KT) issue(kt + 3);
    }
    __builtin_amdgcn_s_barrier();     // the multipliers have left the K loop (ring free)
    __builtin_amdgcn_s_barrier();     // ... and have written the output tile
  } else {
    // ------------------------------------------------------------------ multiplier waves
    f32x4 acc[CF][5];
#pragma unroll
    for (int i = 0; i < CF; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const int fr = l & 15, fq = l >> 4;
    float4 bias[CF];
#pragma unroll
    for (int i = 0; i < CF; ++i) bias[i] = *reinterpret_cast<const float4*>(a.bias + n0 + w * (NT / 4) + i * 16 + 4 * fq);
    for (int kt = 0; kt < KT; ++kt) {
      __builtin_amdgcn_s_barrier();
      const uint16_t* sW = smem + (kt & 3) * STAGE;
      const uint16_t* sI = sW + NT * 64;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int c = kk * 4 + fq;
        frag fa[CF], fb[5];
#pragma unroll
        for (int i = 0; i < CF; ++i) fa[i] = *reinterpret_cast<const frag*>(&sW[swz(w * (NT / 4) + i * 16 + fr, c)]);
#pragma unroll
        for (int j = 0; j < 5; ++j) fb[j] = *reinterpret_cast<const frag*>(&sI[swz(j * 16 + fr, c)]);
#pragma unroll
        for (int i = 0; i < CF; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[i][j] = E::mfma16(fa[i], fb[j], acc[i][j]);
      }
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int ml = j * 16 + fr;
#pragma unroll
      for (int i = 0; i < CF; ++i) {
        const int nl = w * (NT / 4) + i * 16 + 4 * fq;
        float v[4] = {acc[i][j][0] + bias[i].x, acc[i][j][1] + bias[i].y, acc[i][j][2] + bias[i].z, acc[i][j][3] + bias[i].w};
        if (RELU) {
          v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        uint16_t* po = &stO[NT == 128 ? st_off(ml, nl) : st_off64(ml, nl)];
        if (HAS_ADD) {
          float ad[4];
          E::un4(*reinterpret_cast<const uint2*>(po), ad);
          if (a.round_before_add) E::un4(E::pk4(v[0], v[1], v[2], v[3]), v);
          v[0] += ad[0]; v[1] += ad[1]; v[2] += ad[2]; v[3] += ad[3];
        }
        *reinterpret_cast<uint2*>(po) = E::pk4(v[0], v[1], v[2], v[3]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // ------------------------------------------------------------------ all eight waves: copy the tile out, full rows
  constexpr int CH = NT / 8;   // 16-byte chunks per tile row
#pragma unroll
  for (int it = 0; it < (80 * CH + 511) / 512; ++it) {
    const int q = t + 512 * it, row = q / CH, ch = q % CH, m = m0 + row;
    if (q < 80 * CH && m < M) {
      const int so = NT == 128 ? row * 128 + ((ch ^ (row & 15)) << 3) : row * 64 + ((ch ^ (row & 7)) << 3);
      *reinterpret_cast<uint4*>(a.out + (size_t)m * Co + n0 + ch * 8) = *reinterpret_cast<const uint4*>(&stO[so]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// convgemm256: the large-M variant (encoder layers with >= 128 output channels, i.e. 97 % of its FLOPs). 256 rows x 128
// columns per workgroup halves the L2->LDS bytes per FLOP of the 80-row tile (the measured bound of that kernel at
// ~50-70 GB/s of LDS-DMA fill per CU). 16 waves: 8 multipliers (4 x 2 grid of 64 x 64 sub-tiles, 2 x 2
// v_mfma_f32_32x32x16_bf16 fragments: 4 ds_read_b128 feed 4 MFMAs) and 8 loaders (6 DMA instructions each per 64-wide
// K stage: 2 for the W tile, 4 for the In tile). 3-slot ring of 48 KiB stages; the slot rotation is chosen so that the
// LAST stage sits in slot 2, which leaves slots 0-1 free for the [256][128] epilogue tile one stage early: the loaders
// fetch the residual / skip tile into it while the multipliers work on the last stage.
// ---------------------------------------------------------------------------------------------------
template <class E, bool RELU, bool HAS_ADD>
__global__ __launch_bounds__(1024) void convgemm256_kernel(ConvGemmArgs a) {
  typedef typename E::frag frag;
  constexpr int STAGE = (128 + 256) * 64;     // elements per ring slot
  __shared__ __attribute__((aligned(16))) uint16_t smem[3 * STAGE];
  uint16_t* const stO = smem;                 // epilogue tile [256][128] (slots 0-1)
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = a.M, Co = a.Co, Kp = a.Kp;
  const int ntiles = Co >> 7;
  const int mtiles = (M + 255) >> 8;
  const int per_xcd = (mtiles + 7) >> 3;
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + jx / ntiles;
  if (mt >= mtiles) return;
  const int n0 = (jx % ntiles) << 7, m0 = mt << 8;
  const int KT = Kp >> 6;
  const int rot = (3 - (KT % 3)) % 3;         // slot(kt) = (kt + rot) % 3 with slot(KT - 1) == 2

  if (w >= 8) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - 8;
    const uint16_t* gW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (lw * 2 + j) * 8 + (l >> 3);
      gW[j] = a.W + (size_t)(n0 + row) * Kp + ((l & 7) ^ ((row >> 1) & 7)) * 8;
    }
    const uint16_t* ibase[4];
    int iy0[4], ix0[4], kc[4];
    bool pv[4];
    const int hw = a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (lw * 4 + j) * 8 + (l >> 3);
      const int p = m0 + row;
      pv[j] = p < M;
      const int pp = pv[j] ? p : 0;
      const int f = pp / hw, r = pp - f * hw;
      const int y = r / a.Wo, x = r - y * a.Wo;
      iy0[j] = y * a.stride - a.pad;
      ix0[j] = x * a.stride - a.pad;
      ibase[j] = a.In + (size_t)f * a.Hi * a.Wi * a.Ci;
      kc[j] = ((l & 7) ^ ((row >> 1) & 7)) * 8;
    }
    const uint16_t* zp = a.zeros + (l & 7) * 8;
    auto issue = [&](int kt) {
      uint16_t* slot = smem + ((kt + rot) % 3) * STAGE;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(gW[j] + kt * 64), (lvoid_t*)(slot + (lw * 2 + j) * 8 * 64), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k0 = kt * 64 + kc[j];
        const int tap = k0 >> a.ci_shift, ci = k0 & (a.Ci - 1);
        const int ky = (a.ksize == 3) ? (tap * 11) >> 5 : 0;
        const int kx = tap - 3 * ky;
        const int iy = iy0[j] + ky, ix = ix0[j] + kx;
        const bool ok = pv[j] && k0 < a.K && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
        const uint16_t* g = ok ? ibase[j] + (((size_t)iy * a.Wi + ix) << a.ci_shift) + ci : zp;
        __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(slot + 128 * 64 + (lw * 4 + j) * 8 * 64), 16, 0, 0);
      }
    };
    const bool do_loads = !(ACEZ_DBG(a.dbg) & 4);
    if (do_loads) for (int kt = 0; kt < 3 && kt < KT; ++kt) issue(kt);
    for (int kt = 0; kt < KT; ++kt) {
      // issued so far: 0..2 at kt = 0, 0..kt+1 afterwards; 6 DMA instructions per stage, in-order completion
      const int later = (kt == 0) ? min(2, KT - 1) : min(1, KT - 1 - kt);
      if (later >= 2) ACEZ_VMCNT(12);
      else if (later == 1) ACEZ_VMCNT(6);
      else ACEZ_VMCNT(0);
      __builtin_amdgcn_s_barrier();   // stage kt has landed; the multipliers are done with stage kt - 1
      if (do_loads && kt >= 1 && kt + 2 < KT) issue(kt + 2);
      if (HAS_ADD && kt == KT - 1) {
        // slots 0-1 are free from here on (KT >= 3 for every layer that has a residual input): residual tile -> stO
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int row = (lw * 8 + j) * 4 + (l >> 4);
          const uint16_t* g = a.add + (size_t)min(m0 + row, M - 1) * Co + n0 + (((l & 15) ^ (row & 15)) << 3);
          __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(stO + (lw * 8 + j) * 4 * 128), 16, 0, 0);
        }
      }
    }
    ACEZ_VMCNT(0);
    __builtin_amdgcn_s_barrier();     // K loop finished, residual tile landed
    __builtin_amdgcn_s_barrier();     // output tile written
  } else {
    // ------------------------------------------------------------------ multiplier waves
    const int wm = w >> 1, wn = w & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fr = l & 31, fh = l >> 5;
    for (int kt = 0; kt < KT; ++kt) {
      __builtin_amdgcn_s_barrier();
      if (ACEZ_DBG(a.dbg) & 2) continue;
      const uint16_t* sW = smem + ((kt + rot) % 3) * STAGE;
      const uint16_t* sI = sW + 128 * 64;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int c = kk * 2 + fh;
        frag fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const frag*>(&sW[swz(wn * 64 + i * 32 + fr, c)]);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const frag*>(&sI[swz(wm * 64 + j * 32 + fr, c)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = E::mfma32(fa[i], fb[j], acc[i][j]);
      }
    }
    __builtin_amdgcn_s_barrier();
    // the eight bias vectors of this lane, fetched once before the tile is touched (inside the loops every one of the
    // 16-32 loads was followed by a full wait: as many serial L2 round trips per tile)
    float4 bv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const float4*>(a.bias + n0 + wn * 64 + i * 32 + 8 * q + 4 * fh);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ml = wm * 64 + j * 32 + fr;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nl = wn * 64 + i * 32 + 8 * q + 4 * fh;
          const float4 b = bv[i][q];
          float v[4] = {acc[i][j][4 * q + 0] + b.x, acc[i][j][4 * q + 1] + b.y, acc[i][j][4 * q + 2] + b.z, acc[i][j][4 * q + 3] + b.w};
          if (RELU) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
          }
          uint16_t* po = &stO[st_off(ml, nl)];
          if (HAS_ADD) {
            float ad[4];
            E::un4(*reinterpret_cast<const uint2*>(po), ad);
            if (a.round_before_add) E::un4(E::pk4(v[0], v[1], v[2], v[3]), v);
            v[0] += ad[0]; v[1] += ad[1]; v[2] += ad[2]; v[3] += ad[3];
          }
          *reinterpret_cast<uint2*>(po) = E::pk4(v[0], v[1], v[2], v[3]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // ------------------------------------------------------------------ all sixteen waves: copy the tile out, full rows
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = t + 1024 * it, row = q >> 4, ch = q & 15, m = m0 + row;
    if (m < M)
      *reinterpret_cast<uint4*>(a.out + (size_t)m * Co + n0 + ch * 8) = *reinterpret_cast<const uint4*>(&stO[row * 128 + ((ch ^ (row & 15)) << 3)]);
  }
}

// ---------------------------------------------------------------------------------------------------
// convgemm512: 256 rows x 256 columns per workgroup for the layers with >= 256 output channels when there are enough
// tiles to fill the chip several times. Why: every GEMM kernel of this package ends up with ~96 KiB of LDS-DMA in flight per
// CU (the ring is bounded by the 160 KiB LDS) and measures ~70-77 GB/s of fill per CU, i.e. ~1.3 us of latency under load
// (Little's law) -- loads-only and MFMA-only ablations of convgemm256 take the same time and ADD. The only lever left is
// FLOP per byte: 256 x 256 needs 1.5x fewer bytes per FLOP than 256 x 128 (131 FLOP/B: 75 GB/s per CU then feeds the full
// MFMA rate) and its 128 x 64 wave tiles need 0.75 KiB of fragment reads per MFMA instead of 1 KiB.
// 12 waves: 8 multipliers (2 x 4 grid of 128-row x 64-column sub-tiles = 2 x 4 fragments of v_mfma_f32_32x32x16_bf16,
// 128 accumulator registers) and 4 loaders (8 DMA instructions each per stage). K stages are 32 wide (32 KiB), 4-slot ring.
// The [256][256] bf16 epilogue tile needs the whole ring, so a residual input is fetched after the K loop.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz32(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3); }
// [256][256] bf16 epilogue tile: chunk index (0..31) XOR row & 31
__device__ __forceinline__ int st_off256(int row, int col) { return row * 256 + ((((col >> 3) ^ (row & 31)) << 3) | (col & 7)); }

template <class E, bool RELU, bool HAS_ADD>
__global__ __launch_bounds__(768) void convgemm512_kernel(ConvGemmArgs a) {
  typedef typename E::frag frag;
  constexpr int STAGE = 512 * 32;             // elements per ring slot: [W 256 x 32 | In 256 x 32]
  __shared__ __attribute__((aligned(16))) uint16_t smem[4 * STAGE];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = a.M, Co = a.Co, Kp = a.Kp;
  const int ntiles = Co >> 8;
  const int mtiles = (M + 255) >> 8;
  const int per_xcd = (mtiles + 7) >> 3;
  const int KT = Kp >> 5;
  // One workgroup per tile. A PERSISTENT walk over the tiles (one workgroup per CU, the next tile's first stages requested as soon as the
  // output tile is out of LDS) was measured in round 5 (tools/conv_trace.py: a tile is 3.8 us to its first stage, 13.2 us of K loop, 1.6 us
  // of epilogue, 2.4 us until its stores are acknowledged = 21.1 us of a 26.7 us period): bit-identical and 9 % SLOWER (2.55 against 2.34 ms
  // for the head's eight layers) -- behind a tile's own 128 KiB of stores the next first stage lands after 6 us, and the hardware's
  // overlap of one workgroup's drain with the next one's start is better than the in-workgroup sequence.
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + jx / ntiles;
  if (mt >= mtiles) return;
  const int n0 = (jx % ntiles) << 8, m0 = mt << 8;
#ifdef ACEZ_DIAG   // tools/conv_trace.py: stamp i of this tile (slot 4 + i for the first loader wave)
#define CG_STAMP(i) do { if (a.trace && (t == 0 || t == 512)) a.trace[((size_t)(mt * ntiles + jx % ntiles)) * 8 + (t ? 4 : 0) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CG_STAMP(i) do { } while (0)
#endif
  CG_STAMP(0);

  if (w >= 8) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - 8;
    const int lrow = l >> 2, lch = l & 3;     // a DMA instruction covers 16 rows x 64 bytes
    const uint16_t* gW[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (lw * 4 + j) * 16 + lrow;
      gW[j] = a.W + (size_t)(n0 + row) * Kp + (lch ^ ((row >> 2) & 3)) * 8;
    }
    const uint16_t* ibase[4];
    int iy0[4], ix0[4], kc[4];
    bool pv[4];
    const int hw = a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (lw * 4 + j) * 16 + lrow;
      const int p = m0 + row;
      pv[j] = p < M;
      const int pp = pv[j] ? p : 0;
      const int f = pp / hw, r = pp - f * hw;
      const int y = r / a.Wo, x = r - y * a.Wo;
      iy0[j] = y * a.stride - a.pad;
      ix0[j] = x * a.stride - a.pad;
      ibase[j] = a.In + (size_t)f * a.Hi * a.Wi * a.Ci;
      kc[j] = (lch ^ ((row >> 2) & 3)) * 8;
    }
    const uint16_t* zp = a.zeros + lch * 8;
    auto issue = [&](int kt) {
      uint16_t* slot = smem + (kt & 3) * STAGE;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(gW[j] + kt * 32), (lvoid_t*)(slot + (lw * 4 + j) * 16 * 32), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k0 = kt * 32 + kc[j];
        const int tap = k0 >> a.ci_shift, ci = k0 & (a.Ci - 1);
        const int ky = (a.ksize == 3) ? (tap * 11) >> 5 : 0;
        const int kx = tap - 3 * ky;
        const int iy = iy0[j] + ky, ix = ix0[j] + kx;
        const bool ok = pv[j] && k0 < a.K && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
        const uint16_t* g = ok ? ibase[j] + (((size_t)iy * a.Wi + ix) << a.ci_shift) + ci : zp;
        __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(slot + 256 * 32 + (lw * 4 + j) * 16 * 32), 16, 0, 0);
      }
    };
    const bool do_loads = !(ACEZ_DBG(a.dbg) & 4);
    if (do_loads) for (int kt = 0; kt < 4 && kt < KT; ++kt) issue(kt);
    for (int kt = 0; kt < KT; ++kt) {
      const int later = (kt == 0) ? min(3, KT - 1) : min(2, KT - 1 - kt);
      if (later >= 3) ACEZ_VMCNT(24);
      else if (later == 2) ACEZ_VMCNT(16);
      else if (later == 1) ACEZ_VMCNT(8);
      else ACEZ_VMCNT(0);
      __builtin_amdgcn_s_barrier();   // stage kt has landed; the multipliers are done with stage kt - 1
      if (kt == 0) CG_STAMP(1);
      if (do_loads && kt >= 1 && kt + 3 < KT) issue(kt + 3);
    }
    __builtin_amdgcn_s_barrier();     // the multipliers have left the K loop: the ring is free
    CG_STAMP(2);
    if (HAS_ADD) {
      // residual tile [256][256] -> ring space, 128 DMA instructions of 2 rows x 512 bytes (32 per loader)
      for (int j = 0; j < 32; ++j) {
        const int row = (lw * 32 + j) * 2 + (l >> 5);
        const uint16_t* g = a.add + (size_t)min(m0 + row, M - 1) * Co + n0 + (((l & 31) ^ (row & 31)) << 3);
        __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(smem + (lw * 32 + j) * 2 * 256), 16, 0, 0);
      }
      ACEZ_VMCNT(0);
      __builtin_amdgcn_s_barrier();   // residual tile landed
    }
    __builtin_amdgcn_s_barrier();     // output tile written
  } else {
    // ------------------------------------------------------------------ multiplier waves
    const int wm = w >> 2, wn = w & 3;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fr = l & 31, fh = l >> 5;
    for (int kt = 0; kt < KT; ++kt) {
      __builtin_amdgcn_s_barrier();
      if (ACEZ_DBG(a.dbg) & 2) continue;
      const uint16_t* sW = smem + (kt & 3) * STAGE;
      const uint16_t* sI = sW + 256 * 32;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int c = kk * 2 + fh;
        frag fa[2], fb[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const frag*>(&sW[swz32(wn * 64 + i * 32 + fr, c)]);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const frag*>(&sI[swz32(wm * 128 + j * 32 + fr, c)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = E::mfma32(fa[i], fb[j], acc[i][j]);
      }
    }
    __builtin_amdgcn_s_barrier();     // ring free
    CG_STAMP(1);
    if (HAS_ADD) __builtin_amdgcn_s_barrier();
    // the eight bias vectors of this lane, fetched once before the tile is touched (inside the loops every one of the
    // 16-32 loads was followed by a full wait: as many serial L2 round trips per tile)
    float4 bv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const float4*>(a.bias + n0 + wn * 64 + i * 32 + 8 * q + 4 * fh);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ml = wm * 128 + j * 32 + fr;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nl = wn * 64 + i * 32 + 8 * q + 4 * fh;
          const float4 b = bv[i][q];
          float v[4] = {acc[i][j][4 * q + 0] + b.x, acc[i][j][4 * q + 1] + b.y, acc[i][j][4 * q + 2] + b.z, acc[i][j][4 * q + 3] + b.w};
          if (RELU) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
          }
          uint16_t* po = &smem[st_off256(ml, nl)];
          if (HAS_ADD) {
            float ad[4];
            E::un4(*reinterpret_cast<const uint2*>(po), ad);
            if (a.round_before_add) E::un4(E::pk4(v[0], v[1], v[2], v[3]), v);
            v[0] += ad[0]; v[1] += ad[1]; v[2] += ad[2]; v[3] += ad[3];
          }
          *reinterpret_cast<uint2*>(po) = E::pk4(v[0], v[1], v[2], v[3]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // ------------------------------------------------------------------ all twelve waves: copy the tile out, full 512-byte rows
  CG_STAMP(2 + (t ? 1 : 0));   // (multiplier slot 2 / loader slot 3: the epilogue tile is complete)
  for (int q = t; q < 256 * 32; q += 768) {
    const int row = q >> 5, ch = q & 31, m = m0 + row;
    if (m < M)
      *reinterpret_cast<uint4*>(a.out + (size_t)m * Co + n0 + ch * 8) = *reinterpret_cast<const uint4*>(&smem[row * 256 + ((ch ^ (row & 31)) << 3)]);
  }
#ifdef ACEZ_DIAG
  if (a.trace && t == 0) { ACEZ_VMCNT(0); CG_STAMP(3); }   // this wave's stores acknowledged
#endif
}

// ---------------------------------------------------------------------------------------------------
// The 3 x 3, stride-1 layers: conv3x3r below (round 1's conv3x3p, the first kernel of this shape, is in the git history). The 3 x 3, stride-1 layers (res1_conv1/3, res2_conv1/3 = 85 % of the encoder's FLOPs) with the INPUT kept as an
// LDS patch. In the implicit-GEMM kernels above every tap's K stages DMA the same input pixels again (nine times per
// 32-channel chunk); here the loaders bring, per 32-channel chunk, ONE patch of 448 consecutive input pixels (the tile's 256
// output pixels in (frame, y, x) order plus one image row and one pixel on either side: NHWC frames are back to back, so
// "pixel p + (ky-1) * W + (kx-1)" is a plain linear offset and padding is a per-lane validity bit), and the nine tap stages
// of that chunk only stream weights. L2 -> LDS bytes per 32-wide K stage: 16 KiB of weights + 28 KiB / 9 of patch instead
// of 32 KiB. The multipliers read their B fragments straight from the patch (row q = output row + ky * W + kx, 16-byte chunk
// XOR (q >> 2) & 3: conflict free for the unit-stride rows of a fragment; invalid taps read a zero row).
// Tile 256 x 256, 8 multiplier + 4 loader waves, 4-slot weight ring (64 KiB) + 2 patch buffers (56 KiB); the epilogue tile
// takes the whole 128 KiB. Requires W <= 95 (448-row patch), Ci % 32 == 0, Co % 256 == 0.
// ---------------------------------------------------------------------------------------------------
constexpr int P3_ROWS = 448;

// ---------------------------------------------------------------------------------------------------
// conv3x3r: the lean stage loop (round 2). Ablation of conv3x3p on MI355X (tools/enc_kstats.sh): with the LDS-DMA AND
// the MFMAs switched off the 3x3 kernels still take 50 % of their time; loads add 10 %, MFMAs 40 %. The "skeleton" is the stage
// loop itself: per 32-wide K stage a wave executes ~180 scalar / vector / branch instructions (tap decode, nine-way validity
// selects, swizzled fragment addresses, the vmcnt switch, slot arithmetic) around its 16 MFMAs -- ~1250 cycles of in-order issue
// against 512 cycles of matrix work. Here everything that does not change is computed once per lane and kept in registers:
//   * tapaddr[tap][j]: LDS byte address of B fragment j for tap `tap` in patch slot 0 (validity folded in: padded taps point at a
//     zero row inside the slot); the second 16-wide K step is `address ^ 32`, the other patch slot `address ^ 0x8000` (the slots
//     are 32 KiB apart, flipped once per chunk);
//   * the nine taps are unrolled, so tap, validity and the vmcnt of a stage are compile-time constants; the last 32-channel chunk
//     has its own copy of the nine stage bodies (no branches on "is there a next stage / a next patch");
//   * DMA source pointers advance by scalar increments.
// A stage is then 16 MFMAs + 12 ds_read_b128 + 2 global_load_lds + ~14 VALU + ~10 SALU + one barrier. Eight waves that multiply and
// load their own operands (2 weight + amortised 0.5 patch DMA instructions per wave and stage), fragments double-buffered in
// registers (while the 8 MFMAs of one 16-wide K step run, the 6 fragment reads of the next are in flight; that alone, on top of
// conv3x3p's loop, measured +1 %: the loop's instruction count was the limiter, not LDS latency). Same tile, same K order, same
// rounding as conv3x3p.
// LDS (bytes): [0, 64 K) four weight slots; [64 K, 96 K) and [96 K, 128 K) patch slots of 512 rows x 64 B (rows 0..447 data, row 511
// zero); the epilogue tile reuses all 128 KiB.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned char lds_byte;
// SKIP (res2_conv3 + res2_skip, ace_network.py:57-58: x = res2_skip(res) + relu(res2_conv3(x))): after the last 3 x 3 stage the bias and
// the ReLU are applied to the accumulators IN REGISTERS, then Ci2 / 32 more stages multiply the skip layer's weights with its input at
// the tile's own 256 pixels onto the same accumulators -- the separate pointwise launch (159 us per 64 frames at 0.20 of the MFMA peak),
// its 16-bit output map and the epilogue's read of it (315 MB each way) are gone for +5.5 % of K. Skip stages live in a ring of four
// 32 KiB slots (16 KiB weights + 16 KiB input rows, the layouts of a weight slot / of patch rows): the patch slot the last chunk does
// not use takes stage 0 while the last chunk still multiplies; stages 1-3 go out behind the K loop's last barrier, under the
// bias / ReLU pass and stage 0's products. The skip product is not rounded on its own (the reference's half tensor is; one rounding less).
// B2B (res1_conv1 + res1_conv2, ace_network.py:48-49): a 256-channel layer's whole output row fits the 256 x 256 tile, so the pointwise
// layer that follows runs back to back on the finished tile -- out = relu(W2 . relu(conv3x3(In) + bias) + bias2): the 16-bit tile in LDS
// (rounded exactly as the unfused layer stores it) is the B operand, W2's fragments come straight from L2 in the MFMA operand layout (a lane's
// eight K elements are 16 contiguous bytes of a weight row; 256 KiB per tile, no ring, no barrier inside the product), the second
// accumulators replace the first. The 157 MB intermediate map is neither written nor read and the 83 us launch is gone.
template <class E, bool RELU, bool HAS_ADD, bool SKIP = false, bool B2B = false>
__global__ __launch_bounds__(512) void conv3x3r_kernel(ConvGemmArgs a) {
  static_assert(!(SKIP && HAS_ADD), "the fused skip replaces the residual add");
  static_assert(!(B2B && (SKIP || HAS_ADD)), "back-to-back pointwise layer: plain 3 x 3 layer in front");
  typedef typename E::frag frag;
  typedef __attribute__((address_space(3))) const frag lds_frag;
  constexpr unsigned WSLOT = 16384, PATCH0 = 65536, PSLOT = 32768, ZROW = 511 * 64;
  __shared__ __attribute__((aligned(16))) uint16_t smem[65536];
  lds_byte* const lds = (lds_byte*)smem;
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = a.M, Co = a.Co, Kp = a.Kp, Wi = a.Wi;
  const int ntiles = Co >> 8;
  const int mtiles = (M + 255) >> 8;
  const int per_xcd = (mtiles + 7) >> 3;
  const int jx = blockIdx.x >> 3;
  const int mt = (blockIdx.x & 7) * per_xcd + jx / ntiles;
  if (mt >= mtiles) return;
  const int n0 = (jx % ntiles) << 8, m0 = mt << 8;
  const int NC = a.Ci >> 5;                   // 32-channel chunks; stage s = 9 * chunk + tap
  // (Round 6: walking the chunks from a workgroup-dependent start -- the rotation that takes the head's whole-frame kernel off its L2-channel
  // queue, head_maps.hip -- was measured here and is 5-6 % SLOWER: 350 -> 370 us (res1_conv1), 1168 -> 1243 us (res2_conv3). These weight
  // panels are 1.2-4.7 MB; in lockstep every workgroup asks for the same lines at the same time and one fill serves all of them.)
  if (t < 32) {                               // the zero rows of both patch slots (visible after the first barrier)
    *(__attribute__((address_space(3))) unsigned*)(lds + PATCH0 + (t >> 4) * PSLOT + ZROW + (t & 15) * 4) = 0u;
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }

  // ---- LDS-DMA: this wave's share. A DMA instruction covers 16 rows x 64 bytes; lane: row l >> 2, 16-byte chunk l & 3
  const int lrow = l >> 2, lch = l & 3;
  const uint16_t* gW[2];                      // running source pointers of the next weight stage to issue
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (w * 2 + j) * 16 + lrow;
    gW[j] = a.W + (size_t)(n0 + row) * Kp + (lch ^ ((row >> 2) & 3)) * 8;
  }
  const uint16_t* gP[4];                      // running source pointers of the next patch to issue
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (w * 4 + (w == 7 ? 0 : j)) * 16 + lrow;   // wave 7 would cover rows 448..511 (padding + the zero row): it repeats rows 448..463
    const int g = min(max(m0 - Wi - 1 + row, 0), M - 1);
    gP[j] = a.In + ((size_t)g << a.ci_shift) + (lch ^ ((row >> 2) & 3)) * 8;
  }
  unsigned wdst = 0;                          // LDS byte offset of the slot the next weight stage goes to
  int wtap = 0;                               // its tap
  const int w_step = a.Ci, w_wrap = 32 - 8 * a.Ci;   // element increments of the weight pointers: next tap / next chunk
  auto issue_w = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((gvoid_t*)gW[j], (lvoid_t*)(lds + wdst + (w * 2 + j) * 1024), 16, 0, 0);
      gW[j] += (wtap == 8) ? w_wrap : w_step;
    }
    wtap = (wtap == 8) ? 0 : wtap + 1;
    wdst = (wdst + WSLOT) & (4 * WSLOT - 1);
  };
  unsigned pdst = PATCH0;
  auto issue_patch = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((gvoid_t*)gP[j], (lvoid_t*)(lds + pdst + (w * 4 + (w == 7 ? 0 : j)) * 1024), 16, 0, 0);
      gP[j] += 32;
    }
    pdst ^= PSLOT;
  };

  // fused skip: ring of four 32 KiB slots {free patch slot, 0, 32 K, other patch slot}; stage s -> ring[s & 3]
  const unsigned sk_free = PATCH0 + (NC & 1) * PSLOT;   // the patch slot chunk NC - 1 does NOT use
  auto skip_base = [&](int s2) -> unsigned {
    const int k = s2 & 3;
    return k == 0 ? sk_free : (k == 1 ? 0u : (k == 2 ? 32768u : (sk_free ^ PSLOT)));
  };
  auto issue_skip = [&](int s2) {                       // 2 weight + 2 input DMA instructions per wave
    const unsigned base = skip_base(s2);
    // an opaque zero in every address: the compiler cannot hoist this lane arithmetic in front of the K loop (where it was spilled)
    int opq;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opq));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (w * 2 + j) * 16 + lrow + opq;
      const int sw8 = (lch ^ ((row >> 2) & 3)) * 8 + s2 * 32;
      __builtin_amdgcn_global_load_lds((gvoid_t*)(a.W2 + (size_t)(n0 + row) * a.Kp2 + sw8), (lvoid_t*)(lds + base + (w * 2 + j) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gvoid_t*)(a.In2 + (size_t)min(m0 + row, M - 1) * a.Ci2 + sw8), (lvoid_t*)(lds + base + 16384 + (w * 2 + j) * 1024),
                                       16, 0, 0);
    }
  };
  bool skip0_now = false;                     // set for chunk NC - 2: skip stage 0 goes out where a next patch would

  // ---- per-lane constants of the multiplier side
  const int wm = w >> 2, wn = w & 3;
  const int fr = l & 31, fh = l >> 5;
  unsigned tapaddr[9][4];                     // B fragments, K step 0, patch slot 0
  {
    const int hw = a.Hi * Wi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = wm * 128 + j * 32 + fr;     // output row of the tile; patch origin is pixel m0 - Wi - 1
      const int p = m0 + r;
      const int rem = p % hw;
      const int y = rem / Wi, x = rem - y * Wi;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = y + ky - 1, ix = x + kx - 1;
          const bool ok = p < M && iy >= 0 && iy < a.Hi && ix >= 0 && ix < Wi;
          const int q = r + ky * Wi + kx;
          tapaddr[ky * 3 + kx][j] = ok ? PATCH0 + (unsigned)q * 64 + ((unsigned)(fh ^ ((q >> 2) & 3)) << 4) : PATCH0 + ZROW + ((unsigned)fh << 4);
        }
    }
  }
  unsigned wfrag[2];                          // A fragments, K step 0, byte offset inside a weight slot
#pragma unroll
  for (int i = 0; i < 2; ++i) wfrag[i] = (unsigned)swz32(wn * 64 + i * 32 + fr, fh) * 2;
  unsigned wsrc = 0;                          // LDS byte offset of the slot of the stage whose fragments are read next
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  frag faA[2], fbA[4], faB[2], fbB[4];
  auto multiply = [&](const frag (&fa)[2], const frag (&fb)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = E::mfma32(fa[i], fb[j], acc[i][j]);
  };

  bool pdst_pending = false;                  // set per chunk: is there a patch to issue at the next chunk boundary
  // one stage. Entering: faA / fbA hold K step 0 of (chunk, TAP). LAST: the chunk is the last one.
  auto stage = [&](auto tapc, auto lastc) {
    constexpr int TAP = decltype(tapc)::value;
    constexpr bool LAST = decltype(lastc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) faB[i] = *(lds_frag*)(lds + wsrc + (wfrag[i] ^ 32u));
#pragma unroll
    for (int j = 0; j < 4; ++j) fbB[j] = *(lds_frag*)(lds + (tapaddr[TAP][j] ^ 32u));
    multiply(faA, fbA);
    if (LAST && TAP == 8) {                   // the very last stage: nothing to advance to
      multiply(faB, fbB);
      return;
    }
    // advance to the next stage t: wait for this wave's pieces of W(t) (and of everything older). Younger transfers in flight:
    // W(t+1), W(t+2) (2 instructions each; fewer at the end of the last chunk) and, during the first three stages of a chunk that
    // is not the last one, the patch of the next chunk (4), issued right behind W(t+2) at the chunk's start.
    if (LAST) {
      if (SKIP && TAP <= 2) ACEZ_VMCNT(8);      // (+ skip stage 0, issued where a next patch would have been)
      else if (TAP <= 5) ACEZ_VMCNT(4);
      else if (TAP == 6) ACEZ_VMCNT(2);
      else ACEZ_VMCNT(0);
    } else {
      if (TAP <= 2) ACEZ_VMCNT(8);
      else ACEZ_VMCNT(4);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): this wave's reads of the current stage are complete (faB / fbB hold them)
    __builtin_amdgcn_s_barrier();             // W(t) (and its patch) landed everywhere; nobody reads the current stage any more
    if (!LAST || TAP < 5) issue_w();          // W(t+3) into the slot that just became free
    wsrc = (wsrc + WSLOT) & (4 * WSLOT - 1);
    if (TAP == 8) {                           // t is the first stage of the next chunk: the patch slot of this chunk is free
      if (pdst_pending) issue_patch();
      else if (SKIP && skip0_now) issue_skip(0);
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int j = 0; j < 4; ++j) tapaddr[tp][j] ^= PSLOT;
    }
    constexpr int NT = (TAP == 8) ? 0 : TAP + 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) faA[i] = *(lds_frag*)(lds + wsrc + wfrag[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) fbA[j] = *(lds_frag*)(lds + tapaddr[NT][j]);
    multiply(faB, fbB);
  };

  // ---- prologue: patch 0, W(0..3), patch 1
  issue_patch();
  for (int s = 0; s < 4; ++s) issue_w();      // S >= 9 > 4
  if (NC > 1) {
    issue_patch();
    ACEZ_VMCNT(10);                           // younger than W(0): W(1..3) and patch 1
  } else {
    ACEZ_VMCNT(6);
  }
  __builtin_amdgcn_s_barrier();               // W(0), patch 0 and the zero rows are in place
#pragma unroll
  for (int i = 0; i < 2; ++i) faA[i] = *(lds_frag*)(lds + wsrc + wfrag[i]);
#pragma unroll
  for (int j = 0; j < 4; ++j) fbA[j] = *(lds_frag*)(lds + tapaddr[0][j]);
  using std::integral_constant;
  for (int cc = 0; cc + 1 < NC; ++cc) {
    pdst_pending = cc + 2 < NC;               // at the boundary to chunk cc + 1: patch cc + 2 goes into this chunk's slot
    skip0_now = cc + 2 == NC;
    stage(integral_constant<int, 0>{}, integral_constant<bool, false>{});
    stage(integral_constant<int, 1>{}, integral_constant<bool, false>{});
    stage(integral_constant<int, 2>{}, integral_constant<bool, false>{});
    stage(integral_constant<int, 3>{}, integral_constant<bool, false>{});
    stage(integral_constant<int, 4>{}, integral_constant<bool, false>{});
    stage(integral_constant<int, 5>{}, integral_constant<bool, false>{});
    stage(integral_constant<int, 6>{}, integral_constant<bool, false>{});
    stage(integral_constant<int, 7>{}, integral_constant<bool, false>{});
    stage(integral_constant<int, 8>{}, integral_constant<bool, false>{});
  }
  stage(integral_constant<int, 0>{}, integral_constant<bool, true>{});
  stage(integral_constant<int, 1>{}, integral_constant<bool, true>{});
  stage(integral_constant<int, 2>{}, integral_constant<bool, true>{});
  stage(integral_constant<int, 3>{}, integral_constant<bool, true>{});
  stage(integral_constant<int, 4>{}, integral_constant<bool, true>{});
  stage(integral_constant<int, 5>{}, integral_constant<bool, true>{});
  stage(integral_constant<int, 6>{}, integral_constant<bool, true>{});
  stage(integral_constant<int, 7>{}, integral_constant<bool, true>{});
  stage(integral_constant<int, 8>{}, integral_constant<bool, true>{});

  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();               // everybody has left the K loop: all of LDS is free
  if (SKIP) {
    const int NS = a.Ci2 >> 5;                // skip stages (>= 4: launcher)
    // Everything this section needs per lane is computed HERE: an opaque zero (the compiler cannot see its value) rides in every address,
    // or the lane constants below are hoisted in front of the K loop and spilled (55 dwords of scratch in the first build).
    int opq;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opq));
    issue_skip(1); issue_skip(2); issue_skip(3);
    const int fro = fr + opq;
    unsigned inaddr[4];                       // B fragments of the skip input, K step 0, relative to a slot's input half
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned r = wm * 128 + j * 32 + fro;
      inaddr[j] = 16384u + r * 64 + ((unsigned)(fh ^ ((r >> 2) & 3)) << 4);
    }
    // stage 0 landed long ago (it is older than the K loop's last weight stages): its first fragments are requested before the bias pass
    frag fa0[2], fb0[4], fa1[2], fb1[4];
    {
      const unsigned base = skip_base(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) fa0[i] = *(lds_frag*)(lds + base + wfrag[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb0[j] = *(lds_frag*)(lds + base + inaddr[j]);
    }
    {   // bias + activation of the 3 x 3 layer on the accumulators (what the epilogue does for the unfused layer)
      const float* bp = a.bias + n0 + wn * 64 + 4 * fh + opq;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = *reinterpret_cast<const float4*>(bp + i * 32 + 8 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v[4] = {acc[i][j][4 * q + 0] + b.x, acc[i][j][4 * q + 1] + b.y, acc[i][j][4 * q + 2] + b.z, acc[i][j][4 * q + 3] + b.w};
            if (RELU) {
              v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            }
            if (a.round_before_add) E::un4(E::pk4(v[0], v[1], v[2], v[3]), v);   // fp16: relu(conv) is a half tensor before the add
            acc[i][j][4 * q + 0] = v[0]; acc[i][j][4 * q + 1] = v[1]; acc[i][j][4 * q + 2] = v[2]; acc[i][j][4 * q + 3] = v[3];
          }
        }
    }
    // stages in groups of four (NS % 4 == 0: launcher): ring position, wait count and "is there a stage to issue" are compile-time
    // constants of a stage body, and no body has a second exit (early returns inside the loop made the compiler keep copies of the
    // accumulators per exit: 500 dwords of spills)
    auto skip_stage = [&](auto kc, auto vmc, auto morec, int c) {
      constexpr int KR = decltype(kc)::value;   // c & 3
      constexpr int VM = decltype(vmc)::value;  // DMA instructions younger than stage c + 1
      const unsigned base = KR == 0 ? sk_free : (KR == 1 ? 0u : (KR == 2 ? 32768u : (sk_free ^ PSLOT)));
      const unsigned nb = KR == 3 ? sk_free : (KR == 0 ? 0u : (KR == 1 ? 32768u : (sk_free ^ PSLOT)));
#pragma unroll
      for (int i = 0; i < 2; ++i) fa1[i] = *(lds_frag*)(lds + base + (wfrag[i] ^ 32u));
#pragma unroll
      for (int j = 0; j < 4; ++j) fb1[j] = *(lds_frag*)(lds + base + (inaddr[j] ^ 32u));
      multiply(fa0, fb0);
      ACEZ_VMCNT_C(VM);
      __builtin_amdgcn_s_waitcnt(0xC07F);     // this wave's reads of stage c are complete
      __builtin_amdgcn_s_barrier();           // stage c + 1 landed everywhere; nobody reads stage c any more
      if (decltype(morec)::value) issue_skip(c + 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) fa0[i] = *(lds_frag*)(lds + nb + wfrag[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb0[j] = *(lds_frag*)(lds + nb + inaddr[j]);
      multiply(fa1, fb1);
    };
    using IC0 = integral_constant<int, 0>; using IC1 = integral_constant<int, 1>; using IC2 = integral_constant<int, 2>;
    using IC3 = integral_constant<int, 3>; using IC4 = integral_constant<int, 4>; using IC8 = integral_constant<int, 8>;
    using T = integral_constant<bool, true>; using Fl = integral_constant<bool, false>;
    int c = 0;
#pragma clang loop unroll(disable)
    for (; c + 4 < NS; c += 4) {
      skip_stage(IC0{}, IC8{}, T{}, c);
      skip_stage(IC1{}, IC8{}, T{}, c + 1);
      skip_stage(IC2{}, IC8{}, T{}, c + 2);
      skip_stage(IC3{}, IC8{}, T{}, c + 3);
    }
    skip_stage(IC0{}, IC8{}, Fl{}, c);        // the last four stages: nothing left to issue, the waits count down
    skip_stage(IC1{}, IC4{}, Fl{}, c + 1);
    skip_stage(IC2{}, IC0{}, Fl{}, c + 2);
    {                                         // stage NS - 1: ring position 3
      const unsigned base = sk_free ^ PSLOT;
#pragma unroll
      for (int i = 0; i < 2; ++i) fa1[i] = *(lds_frag*)(lds + base + (wfrag[i] ^ 32u));
#pragma unroll
      for (int j = 0; j < 4; ++j) fb1[j] = *(lds_frag*)(lds + base + (inaddr[j] ^ 32u));
      multiply(fa0, fb0);
      multiply(fa1, fb1);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();             // the skip stages are done: all of LDS is free
  }
  if (HAS_ADD) {
    // residual tile [256][256] -> LDS, 128 DMA instructions of 2 rows x 512 bytes (16 per wave)
    for (int j = 0; j < 16; ++j) {
      const int row = (w * 16 + j) * 2 + (l >> 5);
      const uint16_t* g = a.add + (size_t)min(m0 + row, M - 1) * Co + n0 + (((l & 31) ^ (row & 31)) << 3);
      __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(smem + (w * 16 + j) * 2 * 256), 16, 0, 0);
    }
    ACEZ_VMCNT(0);
    __builtin_amdgcn_s_barrier();             // residual tile landed
  }
  // the eight bias vectors of this lane, fetched once before the tile is touched (inside the loops every one of the
  // 16-32 loads was followed by a full wait: as many serial L2 round trips per tile)
  float4 bv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const float4*>((SKIP ? a.bias2 : a.bias) + n0 + wn * 64 + i * 32 + 8 * q + 4 * fh);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ml = wm * 128 + j * 32 + fr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = wn * 64 + i * 32 + 8 * q + 4 * fh;
        const float4 b = bv[i][q];
        float v[4] = {acc[i][j][4 * q + 0] + b.x, acc[i][j][4 * q + 1] + b.y, acc[i][j][4 * q + 2] + b.z, acc[i][j][4 * q + 3] + b.w};
        if (RELU && !SKIP) {                    // (SKIP: the activation went onto the accumulators before the skip stages)
          v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        uint16_t* po = &smem[st_off256(ml, nl)];
        if (HAS_ADD) {
          float ad[4];
          E::un4(*reinterpret_cast<const uint2*>(po), ad);
          if (a.round_before_add) E::un4(E::pk4(v[0], v[1], v[2], v[3]), v);
          v[0] += ad[0]; v[1] += ad[1]; v[2] += ad[2]; v[3] += ad[3];
        }
        *reinterpret_cast<uint2*>(po) = E::pk4(v[0], v[1], v[2], v[3]);
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (B2B) {
    // ---- second product: acc[i][j] = W2[wn*64 + i*32 .. +31][:] . tile[wm*128 + j*32 .. +31][:]  (K = 256 = 16 steps of 16)
    int opq;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opq));   // (keeps this lane arithmetic behind the K loop: see the skip stages)
    const uint16_t* wp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) wp[i] = a.W2 + (size_t)(wn * 64 + i * 32 + fr + opq) * a.Kp2 + 8 * fh;
    unsigned brow[4], bx[4];                  // B fragment of K step kk: lds + brow[j] + (((2 kk + fh) ^ bx[j]) << 4)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned r = wm * 128 + j * 32 + fr + opq;
      brow[j] = r * 512;
      bx[j] = (r & 31) ^ (unsigned)fh;        // (2 kk) ^ fh ^ (r & 31): fh and r & 31 folded
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#ifndef ACEZ_B2B_PF
#define ACEZ_B2B_PF 4
#endif
    constexpr int PF = ACEZ_B2B_PF;           // weight fragments requested PF steps ahead
    frag wa[PF][2];
#pragma unroll
    for (int k = 0; k < PF; ++k)
#pragma unroll
      for (int i = 0; i < 2; ++i) wa[k][i] = *reinterpret_cast<const frag*>(wp[i] + 16 * k);
    frag fbx[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fbx[0][j] = *(lds_frag*)(lds + brow[j] + (bx[j] << 4));
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      if (kk + 1 < 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fbx[(kk + 1) & 1][j] = *(lds_frag*)(lds + brow[j] + ((((unsigned)(2 * (kk + 1))) ^ bx[j]) << 4));
      }
      frag cur[2] = {wa[kk % PF][0], wa[kk % PF][1]};
      if (kk + PF < 16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) wa[kk % PF][i] = *reinterpret_cast<const frag*>(wp[i] + 16 * (kk + PF));
      }
      multiply(cur, fbx[kk & 1]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // every wave has read what it needs of the first tile
    float4 b2v[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) b2v[i][q] = *reinterpret_cast<const float4*>(a.bias2 + wn * 64 + i * 32 + 8 * q + 4 * fh + opq);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ml = wm * 128 + j * 32 + fr;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nl = wn * 64 + i * 32 + 8 * q + 4 * fh;
          const float4 b = b2v[i][q];
          *reinterpret_cast<uint2*>(&smem[st_off256(ml, nl)]) =
              E::pk4(fmaxf(acc[i][j][4 * q + 0] + b.x, 0.f), fmaxf(acc[i][j][4 * q + 1] + b.y, 0.f), fmaxf(acc[i][j][4 * q + 2] + b.z, 0.f),
                     fmaxf(acc[i][j][4 * q + 3] + b.w, 0.f));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int q = t; q < 256 * 32; q += 512) {
    const int row = q >> 5, ch = q & 31, m = m0 + row;
    if (m < M)
      *reinterpret_cast<uint4*>(a.out + (size_t)m * Co + n0 + ch * 8) = *reinterpret_cast<const uint4*>(&smem[row * 256 + ((ch ^ (row & 31)) << 3)]);
  }
}

#ifdef ACEZ_DIAG
static unsigned long long* g_conv_trace = nullptr;
extern "C" void diagz_conv_trace(void* buf) { g_conv_trace = static_cast<unsigned long long*>(buf); }   // tools/conv_trace.py (not an acez_ symbol: the two builds export the same C ABI)
#endif
// tile_mode: 0 = choose by size, 80 / 256 = force that row tile where the layer shape allows it (ACEZ_CONV_TILE, tests)
// the kernel instantiation of the context's 16-bit operand format (ConvGemmArgs::f16)
#define ACEZ_CONV_LAUNCH(kern, grid, blk, ...)                                                    \
  do {                                                                                            \
    if (g.f16) hipLaunchKernelGGL((kern<EltF16, __VA_ARGS__>), grid, blk, 0, s, g);               \
    else hipLaunchKernelGGL((kern<EltBf16, __VA_ARGS__>), grid, blk, 0, s, g);                    \
  } while (0)
void launch_convgemm(const ConvGemmArgs& g_in, bool relu, hipStream_t s, int tile_mode) {
  ConvGemmArgs g = g_in;
#ifdef ACEZ_DIAG
  g.trace = g_conv_trace;
#endif
  const bool patch_ok = g.ksize == 3 && g.stride == 1 && g.pad == 1 && g.Hi == g.Ho && g.Wi == g.Wo && g.Wi <= (P3_ROWS - 258) / 2 &&
                        g.Ci % 32 == 0 && g.Co % 256 == 0 && g.K == g.Kp;
  // the patch kernel pays from one tile per CU on (16 frames of 480x640 at Co = 256: 0.0925 -> 0.0775 ms per frame against the
  // 80-row / 256 x 128 kernels; 32 frames: 0.0715 -> 0.067); round 1's conv3x3p needed four waves of tiles to win
  static const int patch_min_tiles = [] { const char* e = ACEZ_DIAG_ENV("ACEZ_PATCH_MIN_TILES"); return e ? atoi(e) : 256; }();
  const bool use_patch = patch_ok && (tile_mode == 3 || (tile_mode == 0 && (int64_t)((g.M + 255) / 256) * (g.Co / 256) >= patch_min_tiles));
  if (g.W2 && !g.In2) {
    // a pointwise Co -> Co layer behind this one (res1_conv1 + res1_conv2): back to back on conv3x3r's finished tile where the layer runs
    // there and one tile holds a whole output row; else two launches through the scratch map
    if (use_patch && g.Co == 256 && g.Kp2 >= 256 && relu && !g.add) {
      // (launched below with the other conv3x3r forms)
    } else {
      if (!g.skip_scratch || g.add) abort();
      ConvGemmArgs k = g;
      k.W2 = nullptr; k.bias2 = nullptr; k.out = g.skip_scratch;
      launch_convgemm(k, relu, s, tile_mode);
      ConvGemmArgs p = g;
      p.In = g.skip_scratch; p.W = g.W2; p.bias = g.bias2; p.W2 = nullptr; p.bias2 = nullptr;
      p.Hi = g.Ho; p.Wi = g.Wo; p.Ci = g.Co; p.ci_shift = __builtin_ctz(g.Co); p.ksize = 1; p.stride = 1; p.pad = 0; p.K = g.Co; p.Kp = g.Kp2;
      launch_convgemm(p, true, s, tile_mode);
      return;
    }
  }
  if (g.In2 && !(use_patch && g.Ci >= 64 && g.Ci2 % 128 == 0 && g.Kp2 >= g.Ci2 && relu)) {
    // the unfused form: the pointwise skip as its own launch into the scratch map, added by the main layer's epilogue
    if (!g.skip_scratch || g.add) abort();
    ConvGemmArgs k = g;
    k.In = g.In2; k.W = g.W2; k.bias = g.bias2; k.add = nullptr; k.out = g.skip_scratch; k.In2 = nullptr; k.W2 = nullptr; k.bias2 = nullptr;
    k.Hi = g.Ho; k.Wi = g.Wo; k.Ci = g.Ci2; k.ci_shift = __builtin_ctz(g.Ci2); k.ksize = 1; k.stride = 1; k.pad = 0; k.K = g.Ci2; k.Kp = g.Kp2;
    launch_convgemm(k, false, s, tile_mode);
    g.add = g.skip_scratch; g.In2 = nullptr; g.W2 = nullptr; g.bias2 = nullptr;
  }
  if (use_patch) {
    const int ntiles = g.Co / 256, mtiles = (g.M + 255) / 256;
    const dim3 grid(8 * ntiles * ((mtiles + 7) / 8));
    if (!relu) abort();
    const dim3 blkq(512);
    if (g.In2) ACEZ_CONV_LAUNCH(conv3x3r_kernel, grid, blkq, true, false, true);
    else if (g.W2) ACEZ_CONV_LAUNCH(conv3x3r_kernel, grid, blkq, true, false, false, true);
    else if (g.add) ACEZ_CONV_LAUNCH(conv3x3r_kernel, grid, blkq, true, true);
    else ACEZ_CONV_LAUNCH(conv3x3r_kernel, grid, blkq, true, false);
    return;
  }
  const bool huge_ok = g.Co % 256 == 0 && g.Kp >= 256;
  if (huge_ok && (tile_mode == 512 || (tile_mode == 0 && (int64_t)((g.M + 255) / 256) * (g.Co / 256) >= 4 * 256))) {
    const int ntiles = g.Co / 256, mtiles = (g.M + 255) / 256;
    const dim3 grid(8 * ntiles * ((mtiles + 7) / 8)), blk(768);
    if (g.add) {
      if (!relu) abort();
      ACEZ_CONV_LAUNCH(convgemm512_kernel, grid, blk, true, true);
    } else if (relu) {
      ACEZ_CONV_LAUNCH(convgemm512_kernel, grid, blk, true, false);
    } else {
      ACEZ_CONV_LAUNCH(convgemm512_kernel, grid, blk, false, false);
    }
    return;
  }
  const bool big_ok = g.Co % 128 == 0 && g.Kp >= 192;
  if (big_ok && (tile_mode == 256 || (tile_mode == 0 && g.M >= 256 * 128))) {
    // enough rows to fill the chip with 256-row tiles
    const int ntiles = g.Co / 128, mtiles = (g.M + 255) / 256;
    const dim3 grid(8 * ntiles * ((mtiles + 7) / 8)), blk(1024);
    if (g.add) {
      if (!relu) abort();
      ACEZ_CONV_LAUNCH(convgemm256_kernel, grid, blk, true, true);
    } else if (relu) {
      ACEZ_CONV_LAUNCH(convgemm256_kernel, grid, blk, true, false);
    } else {
      ACEZ_CONV_LAUNCH(convgemm256_kernel, grid, blk, false, false);
    }
    return;
  }
  const int nt = (g.Co % 128 == 0) ? 128 : 64;
  const int ntiles = g.Co / nt;
  const int mtiles = (g.M + 79) / 80;
  const dim3 grid(8 * ntiles * ((mtiles + 7) / 8)), blk(512);
  if (nt == 64) {
    if (g.add || !relu) abort();
    ACEZ_CONV_LAUNCH(convgemm_kernel, grid, blk, 64, true, false);
  } else if (g.add) {
    if (!relu) abort();
    ACEZ_CONV_LAUNCH(convgemm_kernel, grid, blk, 128, true, true);
  } else if (relu) {
    ACEZ_CONV_LAUNCH(convgemm_kernel, grid, blk, 128, true, false);
  } else {
    ACEZ_CONV_LAUNCH(convgemm_kernel, grid, blk, 128, false, false);
  }
}

// ---------------------------------------------------------------------------------------------------
// Training-buffer sampling (ace_trainer.py:404-431): per view, `samples` feature rows are drawn uniformly WITH replacement
// among the pixels whose mask is set (torch.multinomial(mask, n, replacement=True) on equal weights) and appended to the
// buffer together with their target pixel 8 * (x + 0.5, y + 0.5) (ace_util.py:7-13) and the view index.
// The draw is a counter-based stream keyed by (seed, view id, sample): reproducible and independent of batching
// (torch's multinomial stream cannot be reproduced). One workgroup = one view x a slice of its samples:
// inclusive prefix counts of the mask in LDS, one wave per sample (binary search, then a 1 KiB row copy).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t smix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t sample_draw(uint64_t seed, uint64_t view_id, uint32_t s) {
  return (uint32_t)(smix64(smix64(seed ^ (view_id * 0xD1342543DE82EF95ull)) + s) >> 32);
}

constexpr int SAMPLE_MAX_HW = 24576;   // feature-map pixels per view the LDS prefix array holds (e.g. 128 x 192)

__global__ __launch_bounds__(256) void sample_views_kernel(const uint16_t* __restrict__ feat, const uint8_t* __restrict__ mask, int hw, int ow,
                                                           int channels, int samples, uint64_t seed, uint64_t first_view_id,
                                                           int view_index_base, uint16_t* __restrict__ out_feat, float* __restrict__ out_px,
                                                           int32_t* __restrict__ out_view, int32_t* __restrict__ out_pix) {
  __shared__ uint16_t pref[SAMPLE_MAX_HW];   // inclusive count of valid pixels up to p (hw <= 24576 < 65536)
  __shared__ int part[256];
  const int v = blockIdx.x, t = threadIdx.x;
  const uint8_t* mk = mask ? mask + (size_t)v * hw : nullptr;
  const int per = (hw + 255) / 256;
  const int lo = t * per, hi = min(hw, lo + per);
  int cnt = 0;
  for (int p = lo; p < hi; ++p) cnt += mk ? (mk[p] != 0) : 1;
  part[t] = cnt;
  __syncthreads();
  // exclusive scan of the 256 partial counts (Hillis-Steele, 8 rounds)
  for (int off = 1; off < 256; off <<= 1) {
    const int x = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += x;
    __syncthreads();
  }
  const int nvalid = part[255];
  int run = part[t] - cnt;
  for (int p = lo; p < hi; ++p) {
    run += mk ? (mk[p] != 0) : 1;
    pref[p] = (uint16_t)run;
  }
  __syncthreads();
  if (nvalid == 0) return;   // the host never passes such a view (ace_trainer.py:377-378 skips it)
  const int lane = t & 63, wave = t >> 6;
  const int per_block = (samples + gridDim.y - 1) / gridDim.y;
  const int s_lo = blockIdx.y * per_block, s_hi = min(samples, s_lo + per_block);
  for (int s = s_lo + wave; s < s_hi; s += 4) {
    const uint32_t r = sample_draw(seed, first_view_id + v, (uint32_t)s);
    const uint32_t k = (uint32_t)(((uint64_t)r * (uint32_t)nvalid) >> 32);   // uniform in [0, nvalid)
    // smallest p with pref[p] > k  == the (k+1)-th valid pixel
    int a = 0, b = hw - 1;
    while (a < b) {
      const int m = (a + b) >> 1;
      if (pref[m] > k) b = m; else a = m + 1;
    }
    const int pix = a;
    const size_t dst = (size_t)v * samples + s;
    const uint16_t* src = feat + ((size_t)v * hw + pix) * channels;
    for (int c = lane * 8; c < channels; c += 512)
      *reinterpret_cast<uint4*>(out_feat + dst * channels + c) = *reinterpret_cast<const uint4*>(src + c);
    if (lane == 0) {
      const int y = pix / ow, x = pix - y * ow;
      out_px[dst * 2 + 0] = 8.0f * ((float)x + 0.5f);
      out_px[dst * 2 + 1] = 8.0f * ((float)y + 0.5f);
      out_view[dst] = view_index_base + v;
      if (out_pix) out_pix[dst] = pix;
    }
  }
}

}  // namespace acez

using namespace acez;

namespace {

struct LayerDesc {
  const char* name;
  int ci, co, k, stride;
};
// Encoder.__init__ order (ace_network.py:26-40); co of the last two layers is the configurable feature size
const LayerDesc kLayers[ACEZ_ENCODER_LAYERS] = {
    {"conv1", 1, 32, 3, 1},         {"conv2", 32, 64, 3, 2},        {"conv3", 64, 128, 3, 2},       {"conv4", 128, 256, 3, 2},
    {"res1_conv1", 256, 256, 3, 1}, {"res1_conv2", 256, 256, 1, 1}, {"res1_conv3", 256, 256, 3, 1}, {"res2_conv1", 256, 512, 3, 1},
    {"res2_conv2", 512, 512, 1, 1}, {"res2_conv3", 512, 512, 3, 1}, {"res2_skip", 256, 512, 1, 1}};

uint16_t host_f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// fp32 -> IEEE half, round to nearest even (what torch .half() / v_cvt_f16_f32 do; overflow -> inf)
uint16_t host_f2h(float f) {
  const _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}

}  // namespace

struct acez_encoder {
  int device = 0, out_channels = 512, max_frames = 0, max_h = 0, max_w = 0, tile_mode = 0;
  bool f16 = false;                    // 16-bit operand format of every layer: bf16, or fp16 (what the reference's autocast runs the encoder in,
                                       // ace_trainer.py:366-367, register_mapping.py:209-210); fp32 accumulation in both
  uint16_t* w1b = nullptr;             // conv1 weights 16-bit [32][16] (k = tap, zero padded): A operand of conv12p_kernel
  float* bias[ACEZ_ENCODER_LAYERS] = {};
  uint16_t* W[ACEZ_ENCODER_LAYERS] = {};   // 16-bit [co][Kp] (layers 1..10)
  int K[ACEZ_ENCODER_LAYERS] = {}, Kp[ACEZ_ENCODER_LAYERS] = {}, co[ACEZ_ENCODER_LAYERS] = {};
  uint16_t* zeros = nullptr;
  uint16_t *a2 = nullptr, *a3 = nullptr, *r4 = nullptr, *x5 = nullptr, *x6 = nullptr, *r7 = nullptr, *x8 = nullptr,
           *x9 = nullptr, *sk = nullptr;
  std::vector<void*> allocs;
};

extern "C" void acez_encoder_destroy(acez_encoder* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (void* p : e->allocs) (void)hipFree(p);
  delete e;
}

extern "C" int acez_encoder_create(acez_encoder** out, const float* const* h_weights, const float* const* h_biases, int out_channels,
                                   int max_frames, int max_h, int max_w, int compute_dtype, int device) {
  ACEZ_REQUIRE(out && h_weights && h_biases, "null pointer");
  ACEZ_REQUIRE(out_channels > 0 && out_channels % 128 == 0, "out_channels must be a positive multiple of 128");
  ACEZ_REQUIRE(max_frames > 0 && max_h >= 8 && max_w >= 8, "bad capacity");
  ACEZ_REQUIRE(compute_dtype == ACEZ_DTYPE_BF16 || compute_dtype == ACEZ_DTYPE_FP16, "compute_dtype must be ACEZ_DTYPE_BF16 or ACEZ_DTYPE_FP16");
  for (int i = 0; i < ACEZ_ENCODER_LAYERS; ++i) ACEZ_REQUIRE(h_weights[i] && h_biases[i], "null layer pointer");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    acez::set_error("no HIP device visible: the encoder kernels need a gfx950 GPU (there is no CPU fallback)");
    return ACEZ_ERR_NODEVICE;
  }
  if (device < 0) ACEZ_HIP_CHECK(hipGetDevice(&device));
  ACEZ_HIP_CHECK(hipSetDevice(device));
  acez_encoder* e = new acez_encoder();
  e->device = device; e->out_channels = out_channels; e->max_frames = max_frames; e->max_h = max_h; e->max_w = max_w;
  if (const char* tm = ACEZ_DIAG_ENV("ACEZ_CONV_TILE")) e->tile_mode = atoi(tm);
  e->f16 = compute_dtype == ACEZ_DTYPE_FP16;
  auto cvt = [&](float f) { return e->f16 ? host_f2h(f) : host_f2bf(f); };
  auto A = [&](void** p, size_t bytes) -> hipError_t {
    hipError_t rc = hipMalloc(p, bytes);
    if (rc == hipSuccess) e->allocs.push_back(*p);
    return rc;
  };
#define ACEZ_ENC_ALLOC(ptr, bytes)                                   \
  do {                                                               \
    hipError_t rc_ = A((void**)&(ptr), (bytes));                     \
    if (rc_ != hipSuccess) {                                         \
      acez::set_error("hipMalloc failed: %s", hipGetErrorString(rc_)); \
      acez_encoder_destroy(e);                                       \
      return ACEZ_ERR_HIP;                                           \
    }                                                                \
  } while (0)
  ACEZ_ENC_ALLOC(e->zeros, 256);
  ACEZ_HIP_CHECK(hipMemset(e->zeros, 0, 256));
  for (int i = 0; i < ACEZ_ENCODER_LAYERS; ++i) {
    const LayerDesc& L = kLayers[i];
    const int co = (i >= 9) ? out_channels : L.co;
    e->co[i] = co;
    const int K = L.k * L.k * L.ci;
    e->K[i] = K; e->Kp[i] = (K + 63) / 64 * 64;
    ACEZ_ENC_ALLOC(e->bias[i], (size_t)co * sizeof(float));
    {
      // fp16: autocast hands conv2d its bias in half precision too (the kernels add it in fp32 to the fp32 accumulator, as cuDNN's fused
      // epilogue does); bf16 keeps the fp32 values
      std::vector<float> b(h_biases[i], h_biases[i] + co);
      if (e->f16)
        for (float& v : b) v = (float)(_Float16)v;
      ACEZ_HIP_CHECK(hipMemcpy(e->bias[i], b.data(), (size_t)co * sizeof(float), hipMemcpyHostToDevice));
    }
    if (i == 0) {
      std::vector<uint16_t> wb(32 * 16, 0);   // [co][1][3][3] is already [co][tap]
      for (int o = 0; o < 32; ++o)
        for (int tp = 0; tp < 9; ++tp) wb[o * 16 + tp] = cvt(h_weights[0][o * 9 + tp]);
      ACEZ_ENC_ALLOC(e->w1b, wb.size() * sizeof(uint16_t));
      ACEZ_HIP_CHECK(hipMemcpy(e->w1b, wb.data(), wb.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    } else {
      // torch layout [co][ci][ky][kx] -> [co][(ky*k + kx) * ci_n + ci], zero padded to Kp
      std::vector<uint16_t> w((size_t)co * e->Kp[i], 0);
      const int kk = L.k * L.k;
      for (int o = 0; o < co; ++o)
        for (int c = 0; c < L.ci; ++c)
          for (int tp = 0; tp < kk; ++tp)
            w[(size_t)o * e->Kp[i] + (size_t)tp * L.ci + c] = cvt(h_weights[i][((size_t)o * L.ci + c) * kk + tp]);
      ACEZ_ENC_ALLOC(e->W[i], w.size() * sizeof(uint16_t));
      ACEZ_HIP_CHECK(hipMemcpy(e->W[i], w.data(), w.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
  }
  const size_t F = max_frames;
  const size_t h2 = (max_h + 1) / 2, w2 = (max_w + 1) / 2, h4 = (h2 + 1) / 2, w4 = (w2 + 1) / 2, h8 = (h4 + 1) / 2, w8 = (w4 + 1) / 2;
  ACEZ_ENC_ALLOC(e->a2, F * h2 * w2 * 64 * 2);
  ACEZ_ENC_ALLOC(e->a3, F * h4 * w4 * 128 * 2);
  const size_t px = F * h8 * w8;
  ACEZ_ENC_ALLOC(e->r4, px * 256 * 2);
  ACEZ_ENC_ALLOC(e->x5, px * 256 * 2);
  ACEZ_ENC_ALLOC(e->x6, px * 256 * 2);
  ACEZ_ENC_ALLOC(e->r7, px * 256 * 2);
  ACEZ_ENC_ALLOC(e->x8, px * 512 * 2);
  ACEZ_ENC_ALLOC(e->x9, px * 512 * 2);
  ACEZ_ENC_ALLOC(e->sk, px * (size_t)out_channels * 2);
#undef ACEZ_ENC_ALLOC
  *out = e;
  return ACEZ_OK;
}

extern "C" int acez_encoder_output_size(int h, int w, int* out_h, int* out_w) {
  ACEZ_REQUIRE(out_h && out_w && h > 0 && w > 0, "bad argument");
  int hh = h, ww = w;
  for (int i = 0; i < 3; ++i) { hh = (hh + 1) / 2; ww = (ww + 1) / 2; }   // three 3x3 stride-2 pad-1 convolutions
  *out_h = hh; *out_w = ww;
  return ACEZ_OK;
}

extern "C" int acez_encoder_forward(acez_encoder* e, const float* d_images, int n_frames, int h, int w, void* d_features, void* stream) {
  ACEZ_REQUIRE(e && d_images && d_features, "null pointer");
  ACEZ_REQUIRE(n_frames > 0 && h >= 8 && w >= 8 && h <= e->max_h && w <= e->max_w, "frame size out of the context's capacity");
  ACEZ_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = (hipStream_t)stream;
  const int h2 = (h + 1) / 2, w2 = (w + 1) / 2, h4 = (h2 + 1) / 2, w4 = (w2 + 1) / 2, h8 = (h4 + 1) / 2, w8 = (w4 + 1) / 2;
  for (int f0 = 0; f0 < n_frames; f0 += e->max_frames) {
    const int F = (n_frames - f0 < e->max_frames) ? n_frames - f0 : e->max_frames;
    const float* img = d_images + (size_t)f0 * h * w;
    uint16_t* feat = (uint16_t*)d_features + (size_t)f0 * h8 * w8 * e->out_channels;
    {
      // conv1 + conv2 in one launch, software-pipelined over 4 x 32 output tiles (conv1 of tile i + 1 beside conv2 of tile i)
      Conv12Args c{};
      c.img = img; c.w1 = e->w1b; c.b1 = e->bias[0]; c.w2 = e->W[1]; c.b2 = e->bias[1]; c.out = e->a2;
      c.F = F; c.H = h; c.W = w; c.H2 = h2; c.W2 = w2; c.Kp2 = e->Kp[1]; c.zero = reinterpret_cast<const float*>(e->zeros);
      c.tiles_y = (h2 + 7) / 8; c.tiles_x = (w2 + 31) / 32; c.n_tiles = F * c.tiles_y * c.tiles_x;
      const int nt4 = F * ((h2 + 3) / 4) * c.tiles_x;
      if (e->f16) hipLaunchKernelGGL((conv12p_kernel<EltF16, 4>), dim3(nt4 < 256 ? nt4 : 256), dim3(512), 0, s, c);
      else hipLaunchKernelGGL((conv12p_kernel<EltBf16, 4>), dim3(nt4 < 256 ? nt4 : 256), dim3(512), 0, s, c);
    }
    auto conv = [&](int li, const uint16_t* in, int hi, int wi, uint16_t* outp, int ho, int wo, const uint16_t* add, bool relu, int skip_li = -1,
                    const uint16_t* skip_in = nullptr, int next_li = -1) {
      const LayerDesc& L = kLayers[li];
      ConvGemmArgs g{};
      if (next_li >= 0) {
        g.W2 = e->W[next_li]; g.bias2 = e->bias[next_li]; g.Kp2 = e->Kp[next_li]; g.skip_scratch = e->x5;
      }
      if (skip_li >= 0) {
        g.In2 = skip_in; g.W2 = e->W[skip_li]; g.bias2 = e->bias[skip_li]; g.Ci2 = kLayers[skip_li].ci; g.Kp2 = e->Kp[skip_li]; g.skip_scratch = e->sk;
      }
      g.In = in; g.W = e->W[li]; g.bias = e->bias[li]; g.add = add; g.out = outp; g.zeros = e->zeros;
      g.Hi = hi; g.Wi = wi; g.Ci = L.ci; g.ci_shift = __builtin_ctz(L.ci); g.Ho = ho; g.Wo = wo; g.Co = e->co[li];
      g.ksize = L.k; g.stride = L.stride; g.pad = L.k / 2; g.K = e->K[li]; g.Kp = e->Kp[li]; g.M = F * ho * wo; g.f16 = e->f16 ? 1 : 0;
      // fp16 = the reference's arithmetic: relu(conv(x)) is a half tensor BEFORE `res + x` / `skip + x` (ace_network.py:52,58), so the
      // activation is rounded before the residual is added and the sum is rounded again; bf16 keeps its single rounding of the fp32 sum
      g.round_before_add = e->f16 ? 1 : 0;
      if (const char* d = ACEZ_DIAG_ENV("ACEZ_CONV_DBG")) g.dbg = atoi(d);
      launch_convgemm(g, relu, s, e->tile_mode);
    };
    conv(2, e->a2, h2, w2, e->a3, h4, w4, nullptr, true);
    conv(3, e->a3, h4, w4, e->r4, h8, w8, nullptr, true);
    // relu(res1_conv2(relu(res1_conv1(res))))  (ace_network.py:48-49): the pointwise layer back to back on res1_conv1's tiles where that
    // layer runs on conv3x3r (two launches through e->x5 on small inputs)
    conv(4, e->r4, h8, w8, e->x6, h8, w8, nullptr, true, -1, nullptr, 5);
    conv(6, e->x6, h8, w8, e->r7, h8, w8, e->r4, true);     // res = res + relu(res1_conv3(x))      ace_network.py:50-52
    conv(7, e->r7, h8, w8, e->x8, h8, w8, nullptr, true);
    conv(8, e->x8, h8, w8, e->x9, h8, w8, nullptr, true);
    // res2_skip(res) + relu(res2_conv3(x))  (ace_network.py:57-58): the skip rides as extra K stages of res2_conv3 where that layer runs on
    // conv3x3r (launch_convgemm falls back to two launches through e->sk on small inputs)
    conv(9, e->x9, h8, w8, feat, h8, w8, nullptr, true, 10, e->r7);
  }
  ACEZ_HIP_CHECK(hipGetLastError());
  return ACEZ_OK;
}

extern "C" int acez_buffer_warp_views(const float* d_images, int n_images, int H, int W, const int32_t* d_image_index, const float* d_theta,
                                      const float* d_jitter, int n_views, int hs, int ws, float* d_out_views, uint8_t* d_out_mask, int map_h,
                                      int map_w, float* d_scratch, void* stream) {
  ACEZ_REQUIRE(d_images && d_image_index && d_theta && d_out_views, "null pointer");
  ACEZ_REQUIRE(n_images > 0 && H > 0 && W > 0 && n_views > 0 && hs > 0 && ws > 0, "bad shape");
  ACEZ_REQUIRE(!d_jitter || d_scratch, "jitter needs the per-view scratch");
  ACEZ_REQUIRE(!d_out_mask || (map_h > 0 && map_w > 0), "bad mask shape");
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
      (void)hipGetLastError();
      acez::set_error("no HIP device visible: the view warp runs on a gfx950 GPU (there is no CPU fallback)");
      return ACEZ_ERR_NODEVICE;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  if (d_jitter) hipLaunchKernelGGL(warp_mean_kernel, dim3(n_views), dim3(1024), 0, s, d_images, d_image_index, d_jitter, H * W, d_scratch);
  hipLaunchKernelGGL(warp_views_kernel, dim3((hs * ws + 255) / 256, n_views), dim3(256), 0, s, d_images, d_image_index, d_theta, d_jitter, d_scratch, H, W,
                     hs, ws, d_out_views);
  if (d_out_mask)
    hipLaunchKernelGGL(warp_mask_kernel, dim3((map_h * map_w + 255) / 256, n_views), dim3(256), 0, s, d_theta, H, W, hs, ws, map_h, map_w, d_out_mask);
  ACEZ_HIP_CHECK(hipGetLastError());
  return ACEZ_OK;
}

extern "C" int acez_buffer_sample_views(const void* d_view_features, const uint8_t* d_masks, int n_views, int map_h, int map_w, int channels,
                                        int samples_per_view, uint64_t seed, uint64_t first_view_id, int32_t view_index_base,
                                        void* d_out_features, float* d_out_target_px, int32_t* d_out_view_idx, int32_t* d_out_pixel,
                                        void* stream) {
  ACEZ_REQUIRE(d_view_features && d_out_features && d_out_target_px && d_out_view_idx, "null pointer");
  ACEZ_REQUIRE(n_views > 0 && map_h > 0 && map_w > 0 && samples_per_view > 0, "bad shape");
  ACEZ_REQUIRE(map_h * map_w <= SAMPLE_MAX_HW, "feature map too large for the sampling kernel (24576 pixels)");
  ACEZ_REQUIRE(channels > 0 && channels % 8 == 0, "channels must be a multiple of 8");
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
      (void)hipGetLastError();
      acez::set_error("no HIP device visible: buffer sampling runs on a gfx950 GPU (there is no CPU fallback)");
      return ACEZ_ERR_NODEVICE;
    }
  }
  const int hw = map_h * map_w;
  int split = (samples_per_view + 255) / 256;   // ~256 samples per workgroup
  if (split > 64) split = 64;
  hipLaunchKernelGGL(sample_views_kernel, dim3(n_views, split), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)d_view_features, d_masks, hw, map_w,
                     channels, samples_per_view, seed, first_view_id, (int)view_index_base, (uint16_t*)d_out_features, d_out_target_px,
                     d_out_view_idx, d_out_pixel);
  ACEZ_HIP_CHECK(hipGetLastError());
  return ACEZ_OK;
}
