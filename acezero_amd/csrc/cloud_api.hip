// cloud_api.hip -- point-cloud extraction filter on the device.
//
// Replaces the per-frame body of ace_vis_util.get_point_cloud_from_network (ace_vis_util.py:430-591) from the point where
// the scene-coordinate maps of a batch of frames exist in HBM (acez_head_forward_maps' output): reprojection error,
// scene-coordinate gradient, the escalating gradient thresholds, depth filter, reprojection filter with its relaxed
// k-th-error / random sub-sampling branches, and the compaction into one [N][3] point list in frame order.
//
// One 256-thread workgroup per frame; the frame's errors and per-pixel state live in LDS (<= 24576 map pixels), every
// count is an integer reduction and the k-th order statistics are radix selects over the fp32 bit patterns, so the keep
// masks are bit-exact against oracle/cloud_oracle.py.  This unit is compiled with -ffp-contract=off: the oracle rounds
// once per operation in the order written here.
#include <stdint.h>

#include "acez_common.h"

namespace acez {

constexpr int PC_MAX_HW = 24576;   // map pixels per frame the LDS arrays hold (e.g. 128 x 192)
constexpr int PC_THREADS = 256;

// per-pixel state byte
constexpr uint8_t PC_CLASS = 7;    // bits 0-2: index of the smallest gradient threshold the pixel passes (4 = none)
constexpr uint8_t PC_DEPTH = 8;    // camera depth below filter_depth
constexpr uint8_t PC_MASK = 16;    // passes gradient + depth (or "keep all" when nothing did)
constexpr uint8_t PC_VALID = 32;   // PC_MASK and reprojection error below the threshold
constexpr uint8_t PC_KEEP = 64;    // final decision

struct CloudArgs {
  const float* sc;      // [n][3][h][w]
  const float* pose;    // [n][12] world -> camera, rows of the 3x4
  const float* K;       // [n][9]
  int h, w;
  float filter_depth;
  int dense;
  int pts_min, pts_max;
  uint64_t seed, first_frame;
  uint8_t* keep;        // [n][h*w]
  int32_t* counts;      // [n]
};

__device__ __forceinline__ uint64_t cloud_smix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t cloud_draw(uint64_t seed, uint64_t frame, uint32_t j) {
  return (uint32_t)(cloud_smix64(cloud_smix64(seed ^ (frame * 0xA0761D6478BD642Full)) + j) >> 32);
}

// exclusive scan of one int per thread (256 threads); returns the exclusive prefix, *total = sum
__device__ __forceinline__ int block_scan(int v, int* part, int* total) {
  const int t = threadIdx.x;
  __syncthreads();
  part[t] = v;
  __syncthreads();
  for (int off = 1; off < PC_THREADS; off <<= 1) {
    const int x = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += x;
    __syncthreads();
  }
  *total = part[PC_THREADS - 1];
  return part[t] - v;
}

__device__ __forceinline__ int block_count(int v, int* slot) {
  const int t = threadIdx.x;
  __syncthreads();
  if (t == 0) *slot = 0;
  __syncthreads();
  int s = v;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((t & 63) == 0) atomicAdd(slot, s);
  __syncthreads();
  return *slot;
}

// k-th smallest (0-based) 32-bit key among the pixels whose state has `bit` set: four 8-bit histogram passes
__device__ uint32_t radix_select(const uint32_t* keys, const uint8_t* st, uint8_t bit, int hw, int k, int* hist, int* bc) {
  const int t = threadIdx.x;
  uint32_t prefix = 0, pmask = 0;
  for (int b = 3; b >= 0; --b) {
    __syncthreads();
    hist[t] = 0;
    __syncthreads();
    for (int p = t; p < hw; p += PC_THREADS)
      if ((st[p] & bit) && (keys[p] & pmask) == prefix) atomicAdd(&hist[(keys[p] >> (8 * b)) & 255], 1);
    __syncthreads();
    if (t == 0) {
      int cum = 0, bin = 0;
      for (; bin < 255; ++bin) {
        if (cum + hist[bin] > k) break;
        cum += hist[bin];
      }
      bc[0] = bin;
      bc[1] = k - cum;
    }
    __syncthreads();
    prefix |= (uint32_t)bc[0] << (8 * b);
    pmask |= 0xffu << (8 * b);
    k = bc[1];
  }
  __syncthreads();
  return prefix;
}

__global__ __launch_bounds__(PC_THREADS) void cloud_filter_kernel(CloudArgs a) {
  __shared__ float s_err[PC_MAX_HW];
  __shared__ uint8_t s_st[PC_MAX_HW];
  __shared__ int s_hist[256];
  __shared__ int s_part[PC_THREADS];
  __shared__ int s_cnt[8];
  const int f = blockIdx.x, t = threadIdx.x;
  const int h = a.h, w = a.w, hw = h * w;
  const float* X0 = a.sc + (size_t)f * 3 * hw;
  const float* X1 = X0 + hw;
  const float* X2 = X1 + hw;
  float P[12], K[9];
#pragma unroll
  for (int i = 0; i < 12; ++i) P[i] = a.pose[(size_t)f * 12 + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) K[i] = a.K[(size_t)f * 9 + i];
  if (t < 8) s_cnt[t] = 0;
  __syncthreads();

  // ---- pass 1: reprojection error (ace_vis_util.py:481-499), gradient class (:501-510), depth (:519) ----
  const float inf = __builtin_inff();
  for (int p = t; p < hw; p += PC_THREADS) {
    const int y = p / w, x = p - y * w;
    const float x0 = X0[p], x1 = X1[p], x2 = X2[p];
    float cam[3], px[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) cam[i] = ((P[4 * i] * x0 + P[4 * i + 1] * x1) + P[4 * i + 2] * x2) + P[4 * i + 3];
#pragma unroll
    for (int i = 0; i < 3; ++i) px[i] = (K[3 * i] * cam[0] + K[3 * i + 1] * cam[1]) + K[3 * i + 2] * cam[2];
    const float z = fmaxf(px[2], 0.1f);
    const float gx = 8.0f * ((float)x + 0.5f), gy = 8.0f * ((float)y + 0.5f);
    const float err = fabsf(px[0] / z - gx) + fabsf(px[1] / z - gy);
    // difference to the left neighbour; column 0 takes column 1's value (reflect padding of the difference array), same for rows
    const int xa = (x == 0) ? (p + 2) : p, xb = (x == 0) ? (p + 1) : (p - 1);
    const int ya = (y == 0) ? (p + 2 * w) : p, yb = (y == 0) ? (p + w) : (p - w);
    float d0 = X0[xa] - X0[xb], d1 = X1[xa] - X1[xb], d2 = X2[xa] - X2[xb];
    const float gdx = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    d0 = X0[ya] - X0[yb]; d1 = X1[ya] - X1[yb]; d2 = X2[ya] - X2[yb];
    const float gdy = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    const float g = (gdx != gdx || gdy != gdy) ? __builtin_nanf("") : fmaxf(gdx, gdy);   // torch.max propagates NaN
    uint8_t cls = (g < 0.1f) ? 0 : (g < 0.5f) ? 1 : (g < 1.0f) ? 2 : (g < inf) ? 3 : 4;
    if (cam[2] < a.filter_depth) cls |= PC_DEPTH;
    s_err[p] = err;
    s_st[p] = cls;
  }
  __syncthreads();
  {
    int c[4] = {0, 0, 0, 0};
    for (int p = t; p < hw; p += PC_THREADS) {
      const int k = s_st[p] & PC_CLASS;
      if (k < 4) c[k]++;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int s = c[k];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if ((t & 63) == 0 && s) atomicAdd(&s_cnt[k], s);
    }
  }
  __syncthreads();
  // tightest threshold that still leaves more than pts_min pixels (:512-516); the last one if none does
  int ti = 3;
  if (!a.dense) {
    int cum = 0;
    for (int k = 0; k < 4; ++k) {
      cum += s_cnt[k];
      if (cum > a.pts_min) { ti = k; break; }
    }
  }
  const float repro_thr = a.dense ? inf : 1.0f;

  // ---- pass 2: gradient + depth mask (:518-526), reprojection filter (:528-530) ----
  int m = 0;
  for (int p = t; p < hw; p += PC_THREADS) {
    const uint8_t s = s_st[p];
    const bool ok = (s & PC_CLASS) <= ti && (s & PC_DEPTH);
    if (ok) s_st[p] = s | PC_MASK;
    m += ok;
  }
  int n_mask = block_count(m, &s_cnt[4]);
  if (n_mask == 0) n_mask = hw;   // "if no points survive, keep all"
  const bool keep_all = (n_mask == hw);
  int v = 0;
  for (int p = t; p < hw; p += PC_THREADS) {
    uint8_t s = s_st[p];
    if (keep_all) s |= PC_MASK;
    const bool ok = (s & PC_MASK) && s_err[p] < repro_thr;
    s = ok ? (s | PC_VALID | PC_KEEP) : s;
    s_st[p] = s;
    v += ok;
  }
  const int n_valid = block_count(v, &s_cnt[5]);
  int n_keep = n_valid;

  if (n_valid < a.pts_min) {
    // ---- too few: the pts_min points with the lowest error among the masked ones (:535-544) ----
    const int k = min(a.pts_min, n_mask - 1);
    const uint32_t relaxed_bits = radix_select(reinterpret_cast<const uint32_t*>(s_err), s_st, PC_MASK, hw, k, s_hist, &s_cnt[6]);
    const float relaxed = __uint_as_float(relaxed_bits);
    v = 0;
    for (int p = t; p < hw; p += PC_THREADS) {
      uint8_t s = s_st[p] & (uint8_t)~PC_KEEP;
      const bool ok = (s & PC_MASK) && s_err[p] < relaxed;
      s_st[p] = ok ? (s | PC_KEEP) : s;
      v += ok;
    }
    n_keep = block_count(v, &s_cnt[5]);
  } else if (n_valid > a.pts_max) {
    // ---- too many: a uniformly random k-subset (:545-551), counter-based draw keyed by (seed, frame, rank of the point) ----
    const double keep_ratio = (double)a.pts_max / (double)n_valid;
    const int k = (int)(keep_ratio * (double)n_valid);
    const int per = (hw + PC_THREADS - 1) / PC_THREADS;
    const int lo = min(hw, t * per), hi = min(hw, lo + per);
    int c = 0;
    for (int p = lo; p < hi; ++p) c += (s_st[p] & PC_VALID) != 0;
    int total;
    int rank = block_scan(c, s_part, &total);
    uint32_t* key = reinterpret_cast<uint32_t*>(s_err);
    for (int p = lo; p < hi; ++p)
      if (s_st[p] & PC_VALID) key[p] = cloud_draw(a.seed, a.first_frame + f, (uint32_t)rank++);
    __syncthreads();
    const uint32_t hk = radix_select(key, s_st, PC_VALID, hw, k, s_hist, &s_cnt[6]);
    c = 0;
    int ties = 0;
    for (int p = lo; p < hi; ++p) {
      if (!(s_st[p] & PC_VALID)) continue;
      c += key[p] < hk;
      ties += key[p] == hk;
    }
    const int n_less = block_count(c, &s_cnt[5]);
    int tie_rank = block_scan(ties, s_part, &total);
    const int need = k - n_less;   // ties at the k-th key: the first `need` in pixel order
    for (int p = lo; p < hi; ++p) {
      uint8_t s = s_st[p];
      if (!(s & PC_VALID)) continue;
      const bool ok = key[p] < hk || (key[p] == hk && tie_rank++ < need);
      s_st[p] = ok ? s : (uint8_t)(s & ~PC_KEEP);
    }
    n_keep = k;
  }
  __syncthreads();
  uint8_t* out = a.keep + (size_t)f * hw;
  for (int p = t; p < hw; p += PC_THREADS) out[p] = (s_st[p] & PC_KEEP) ? 1 : 0;
  if (t == 0) a.counts[f] = n_keep;
}

// offsets[i] = sum of counts[0..i), offsets[n] = total
__global__ __launch_bounds__(PC_THREADS) void cloud_offsets_kernel(const int32_t* __restrict__ counts, int n, int32_t* __restrict__ offsets) {
  __shared__ int s_part[PC_THREADS];
  const int t = threadIdx.x;
  const int per = (n + PC_THREADS - 1) / PC_THREADS;
  const int lo = min(n, t * per), hi = min(n, lo + per);
  int c = 0;
  for (int i = lo; i < hi; ++i) c += counts[i];
  int total;
  int run = block_scan(c, s_part, &total);
  for (int i = lo; i < hi; ++i) {
    offsets[i] = run;
    run += counts[i];
  }
  if (t == 0) offsets[n] = total;
}

__global__ __launch_bounds__(PC_THREADS) void cloud_pack_kernel(const float* __restrict__ sc, const uint8_t* __restrict__ keep,
                                                               const int32_t* __restrict__ offsets, int hw, float flip,
                                                               float* __restrict__ xyz, int32_t* __restrict__ source) {
  __shared__ int s_part[PC_THREADS];
  const int f = blockIdx.x, t = threadIdx.x;
  const uint8_t* kp = keep + (size_t)f * hw;
  const float* X = sc + (size_t)f * 3 * hw;
  const int per = (hw + PC_THREADS - 1) / PC_THREADS;
  const int lo = min(hw, t * per), hi = min(hw, lo + per);
  int c = 0;
  for (int p = lo; p < hi; ++p) c += kp[p];
  int total;
  int run = offsets[f] + block_scan(c, s_part, &total);
  for (int p = lo; p < hi; ++p) {
    if (!kp[p]) continue;
    xyz[(size_t)run * 3 + 0] = X[p];
    xyz[(size_t)run * 3 + 1] = flip * X[hw + p];   // OpenCV -> OpenGL convention (ace_vis_util.py:585-587)
    xyz[(size_t)run * 3 + 2] = flip * X[2 * hw + p];
    if (source) source[run] = f * hw + p;
    ++run;
  }
}

}  // namespace acez

using namespace acez;

extern "C" int acez_point_cloud_filter(const float* d_scene_coords, const float* d_poses_inv, const float* d_intrinsics, int n_frames,
                                       int map_h, int map_w, float filter_depth, int dense_cloud, int points_per_image_min,
                                       int points_per_image_max, uint64_t seed, uint64_t first_frame_id, int opengl_convention,
                                       uint8_t* d_keep, int32_t* d_counts, int32_t* d_offsets, float* d_out_xyz,
                                       int32_t* d_out_source, void* stream) {
  ACEZ_REQUIRE(d_scene_coords && d_poses_inv && d_intrinsics && d_keep && d_counts && d_offsets && d_out_xyz, "null pointer");
  ACEZ_REQUIRE(n_frames > 0 && map_h >= 3 && map_w >= 3, "need at least one frame and a map of 3 x 3 (reflect padding of the gradient)");
  ACEZ_REQUIRE(map_h * map_w <= PC_MAX_HW, "scene-coordinate map larger than 24576 pixels");
  ACEZ_REQUIRE((int64_t)n_frames * map_h * map_w < (int64_t)1 << 31, "more than 2^31 map pixels in one call");
  ACEZ_REQUIRE(points_per_image_min >= 0 && points_per_image_max >= points_per_image_min, "bad per-image point budgets");
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
      (void)hipGetLastError();
      acez::set_error("no HIP device visible: point-cloud extraction runs on a gfx950 GPU (there is no CPU fallback)");
      return ACEZ_ERR_NODEVICE;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  CloudArgs a{d_scene_coords, d_poses_inv, d_intrinsics, map_h, map_w, filter_depth, dense_cloud, points_per_image_min, points_per_image_max,
              seed, first_frame_id, d_keep, d_counts};
  hipLaunchKernelGGL(cloud_filter_kernel, dim3(n_frames), dim3(PC_THREADS), 0, s, a);
  ACEZ_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(cloud_offsets_kernel, dim3(1), dim3(PC_THREADS), 0, s, (const int32_t*)d_counts, n_frames, d_offsets);
  ACEZ_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(cloud_pack_kernel, dim3(n_frames), dim3(PC_THREADS), 0, s, d_scene_coords, (const uint8_t*)d_keep, (const int32_t*)d_offsets,
                     map_h * map_w, opengl_convention ? -1.0f : 1.0f, d_out_xyz, d_out_source);
  ACEZ_HIP_CHECK(hipGetLastError());
  return ACEZ_OK;
}
