// ransac_api.hip -- DSAC* RGB registration on gfx950: one 256-thread workgroup per frame, frames batched over
// the grid. Replaces dsacstar_rgb_forward (dsacstar/dsacstar.cpp:66-186). Build with -ffp-contract=off.
//
// Mapping of the reference's stages onto the workgroup (4 wavefronts):
//   load        the frame's H*W scene coordinates are staged ONCE into LDS (3 x N floats, pixel order
//               p = x*H + y, the reference's x-outer/y-inner scan order) with coalesced global reads
//   sample      sampleHypotheses (dsacstar_util.h:135-221): ONE HYPOTHESIS PER WAVEFRONT, the up-to-max_tries
//               minimal-set draws run 64 at a time on the lanes (each lane: 4 draws -> P3P -> 4-point check);
//               __ballot + first-set-bit picks the first accepted try = what the sequential loop would keep
//   score       getReproErrs + getHypScores (:316-343,:356-446): the same wavefront projects all N pixels of its
//               hypothesis from LDS, lane-strided fp64 partial sums, one butterfly (order 32,16,..,1)
//   select      softMax + draw(argmax) (:684-752) by one lane
//   refine      refineHyp (:522-597): all 256 threads. Every round classifies all pixels against the threshold and
//               COMPACTS the inliers into an LDS list in scan order -- the reference's own localImgPts / localObjPts
//               (:545-560) -- so that the Levenberg-Marquardt passes of solvePnP(ITERATIVE) (cvFindExtrinsicCameraParams2
//               + CvLevMarq) touch inliers only: list entry j belongs to thread j % 256, J^T J / J^T e / |e|^2 are
//               accumulated per thread, reduced inside each wavefront by a TRANSPOSING butterfly (v_permlane32_swap,
//               v_permlane16_swap, DPP: 28 sums cost 29 cross-lane steps instead of 168) and summed over the 4
//               wavefronts in a fixed order through LDS; the 6x6 solve runs on wavefront 0.
// Nothing is written to HBM between stages except the diagnostics (hypothesis poses / scores).
// Roofline: per frame 57.6 KB in, ~5 KB out; the kernel is fp64-VALU/latency bound.
#include <hip/hip_runtime.h>
#include "ransac_math.h"
#include "acez_common.h"
#include <new>
#include <vector>

namespace {

using rsm::Cam;
using rsm::Pose;

struct FrameParam {
  float focal, ppx, ppy, pad;
  uint64_t frame_id;
};

struct RansacArgs {
  const float* sc;  // [n][3][H][W]
  const FrameParam* fp;
  float* big;       // [n][3][Npad] transposed copies, only for frames that do not fit the LDS (GC instantiation)
  int H, W, N, hyps, max_tries, sub, max_ref_steps;
  uint32_t h_magic;  // ceil(2^32 / H): p / H == __umulhi(p, h_magic) for p < 2^16 (H >= 2)
  float thr, alpha, max_reproj;
  uint64_t seed;
  double* hyp_poses;  // [n][hyps][6]
  double* scores;     // [n][hyps]
  int* best;          // [n]
  double* refined;    // [n][6]
  float* out_poses;   // [n][16]
  int* out_inliers;   // [n]
  uint8_t* out_masks; // [n][H][W] or null
};

// ---- cross-lane sums of doubles without the LDS crossbar --------------------------------------------------------------------
// All of them realise the butterfly v_l + v_(l ^ off), off = 32, 16, 8, 4, 2, 1 (the order of oracle tree64): IEEE addition is
// commutative, so which partner supplies the left operand does not matter and every lane ends with the same bits.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// a[32..63] <-> b[0..31]
__device__ __forceinline__ void swap_halves(double& a, double& b) {
  const u32x2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const u32x2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi.x, (int)lo.x);
  b = __hiloint2double((int)hi.y, (int)lo.y);
}
// odd 16-lane rows of a <-> even rows of b
__device__ __forceinline__ void swap_rows(double& a, double& b) {
  const u32x2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const u32x2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi.x, (int)lo.x);
  b = __hiloint2double((int)hi.y, (int)lo.y);
}
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_d(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, BANK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, BANK, false);
  return __hiloint2double(hi, lo);
}
// offsets 8, 4, 2, 1 inside each row of 16 lanes
__device__ __forceinline__ double row_tree(double v) {
  v = v + dpp_d<0x128, 0xf>(v, v);          // row_ror:8 = lane ^ 8
  double t = dpp_d<0x104, 0x5>(v, v);        // row_shl:4 on lanes 0-3, 8-11 (read lane + 4)
  t = dpp_d<0x114, 0xa>(t, v);               // row_shr:4 on lanes 4-7, 12-15 (read lane - 4)  -> lane ^ 4
  v = v + t;
  v = v + dpp_d<0x4E, 0xf>(v, v);            // quad_perm [2,3,0,1] = lane ^ 2
  v = v + dpp_d<0xB1, 0xf>(v, v);            // quad_perm [1,0,3,2] = lane ^ 1
  return v;
}
__device__ __forceinline__ double wave_tree(double v) {
  double a = v, b = v;
  swap_halves(a, b);
  v = a + b;
  a = v; b = v;
  swap_rows(a, b);
  v = a + b;
  return row_tree(v);
}
// 28 sums at once: after the two swap levels row r of the wavefront works on quantities 7r .. 7r+6 only
__device__ __forceinline__ void wave_tree28(const double acc[28], double out7[7]) {
  double h[14];
#pragma unroll
  for (int j = 0; j < 14; ++j) {
    double a = acc[j], b = acc[14 + j];
    swap_halves(a, b);   // lanes < 32: (own acc[j], acc[j] of lane + 32); lanes >= 32: (acc[14 + j] of lane - 32, own acc[14 + j])
    h[j] = a + b;
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    double a = h[j], b = h[7 + j];
    swap_rows(a, b);
    out7[j] = row_tree(a + b);
  }
}

// x = p / H for p < 2^16 (see RansacArgs::h_magic)
__device__ __forceinline__ int div_h(int p, int H, uint32_t magic) { return H == 1 ? p : (int)__umulhi((uint32_t)p, magic); }

// detm::exp_ with 32-bit integer steps (|x * INV_LN2| < 1100, so the int conversion, k / 2 and the exponent words are the same
// values as the 64-bit ones of det_math.h; every floating-point operation is identical)
__device__ __forceinline__ double exp_dev(double x) {
  if (x != x) return x;
  if (x > 709.0) return __hiloint2double(0x7ff00000, 0);
  if (x < -745.0) return 0.0;
  const double kf = x * detm::INV_LN2;
  const int k = (int)(kf + (kf >= 0 ? 0.5 : -0.5));
  const double kd = (double)k;
  const double r = (x - kd * detm::LN2_HI) - kd * detm::LN2_LO;
  const double p = (0x1.0000000000000p+0 + r * (0x1.0000000000000p+0 + r * (0x1.0000000000000p-1 + r * (0x1.5555555555555p-3 + r * (0x1.555555555555fp-5 + r * (0x1.1111111111114p-7 + r * (0x1.6c16c16c1321dp-10 + r * (0x1.a01a01a018130p-13 + r * (0x1.a01a01b4cf7bbp-16 + r * (0x1.71de3a60775bep-19 + r * (0x1.27e4beb6f974bp-22 + r * (0x1.ae6415b4ca760p-26 + r * (0x1.1f9e46dddd4e5p-29 + r * 0x1.61e0dace500bbp-33)))))))))))));
  const int k1 = k / 2, k2 = k - k1;
  return (p * __hiloint2double((k1 + 1023) << 20, 0)) * __hiloint2double((k2 + 1023) << 20, 0);
}

// reprojection error of one pixel exactly as getReproErrs stores it (float, clamped)
__device__ __forceinline__ float pixel_err(const double R[9], const double t[3], const Cam& k, float X, float Y, float Z, int pxx,
                                           int pyy, float max_reproj) {
  double u, v;
  rsm::project(R, t, k, X, Y, Z, &u, &v, nullptr, nullptr, nullptr);
  const float pu = (float)u, pv = (float)v;
  const float dx = (float)pxx - pu, dy = (float)pyy - pv;
  const double nrm = sqrt((double)dx * dx + (double)dy * dy);
  const float l = (float)nrm;
  return l < max_reproj ? l : max_reproj;
}

struct LMAccum {
  double JtJ[36], JtErr[6], errsq;
};

__device__ bool inv4x4(const double Ain[16], double out[16]) {
  double A[16], Bm[16];
  for (int i = 0; i < 16; ++i) {
    A[i] = Ain[i];
    Bm[i] = (i % 5 == 0) ? 1. : 0.;
  }
  const int m = 4, n = 4;
  for (int i = 0; i < m; i++) {
    int k = i;
    for (int j = i + 1; j < m; j++)
      if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
    if (fabs(A[k * m + i]) < 2.220446049250313e-16 * 100) return false;
    if (k != i) {
      for (int j = i; j < m; j++) { const double tmp = A[i * m + j]; A[i * m + j] = A[k * m + j]; A[k * m + j] = tmp; }
      for (int j = 0; j < n; j++) { const double tmp = Bm[i * n + j]; Bm[i * n + j] = Bm[k * n + j]; Bm[k * n + j] = tmp; }
    }
    const double d = -1 / A[i * m + i];
    for (int j = i + 1; j < m; j++) {
      const double alpha = A[j * m + i] * d;
      for (int kk = i + 1; kk < m; kk++) A[j * m + kk] += alpha * A[i * m + kk];
      for (int kk = 0; kk < n; kk++) Bm[j * n + kk] += alpha * Bm[i * n + kk];
    }
  }
  for (int i = m - 1; i >= 0; i--)
    for (int j = 0; j < n; j++) {
      double s = Bm[i * n + j];
      for (int k = i + 1; k < m; k++) s -= A[i * m + k] * Bm[k * n + j];
      Bm[i * n + j] = s / A[i * m + i];
    }
  for (int i = 0; i < 16; ++i) out[i] = Bm[i];
  return true;
}

// ---- LDS layout (host and device agree through these helpers) ---------------------------------------------------------------
//   [3][Npad] float   scene coordinates in scan order (LDS instantiation only)
//   [Npad]    uint16  the current inlier list
//   region            one fp64 area used twice: while sampling / scoring / selecting it holds the scores [hyps] and the sampled
//                     poses [hyps][6]; both are dead once the best hypothesis is chosen, and the refinement reuses it for the LM
//                     step hand-over (6 doubles, padded to 8), the double-buffered reduction scratch [2][4][28] and the compaction
//                     counts [64][4] int. Keeping the two apart cost 1.8 KB more and, with 64 hypotheses, the second workgroup
//                     of a CU (round 1: 20 ms vs 12 ms).
//   [8] int           best hypothesis
// 60x80 (7-Scenes): 70.1 KB, 60x93 (Mip-NeRF 360 at 480 px): 81.7 KB with 64 hypotheses -> two workgroups per CU in both cases.
constexpr int MAX_ROWS = 64;                        // pixels per thread (inlier flags are one 64-bit word): N <= 16384
constexpr int RED_DOUBLES = 8 + 2 * 4 * 28;         // step hand-over + reduction scratch
constexpr int REGION_MIN = RED_DOUBLES + MAX_ROWS * 4 / 2;
__host__ __device__ inline int region_doubles(int hyps) { return 7 * hyps > REGION_MIN ? 7 * hyps : REGION_MIN; }
__host__ __device__ inline size_t lds_bytes(int N, int hyps, bool coords_in_hbm) {
  const size_t Npad = (size_t)((N + 3) & ~3);
  return (coords_in_hbm ? 0 : 12 * Npad) + 2 * Npad + 8 * (size_t)region_doubles(hyps) + 8 * sizeof(int);
}

// GC: the frame does not fit the LDS (more than ~11 400 scene coordinates): its scan-order copy lives in an HBM workspace (L2
// resident, 12 N bytes) and every stage reads it from there; same arithmetic, same order, same bits.
template <bool GC>
__global__ __launch_bounds__(256, 2) void ransac_kernel(RansacArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = a.N, H = a.H, W = a.W;
  const int Npad = (N + 3) & ~3;
  const int frame = blockIdx.x;
  float* sX = GC ? a.big + (size_t)frame * 3 * Npad : reinterpret_cast<float*>(smem_raw);
  float* sY = sX + Npad;
  float* sZ = sY + Npad;
  uint16_t* sList = reinterpret_cast<uint16_t*>(smem_raw + (GC ? 0 : 12 * (size_t)Npad));
  double* sRegion = reinterpret_cast<double*>(sList + Npad);
  double* sScores = sRegion;                              // [hyps]
  double* sHyp = sScores + a.hyps;                        // [hyps][6] sampled poses
  double* sRed = sRegion + 8;                             // [2][4][28] (refinement only)
  int* sCnt = reinterpret_cast<int*>(sRegion + RED_DOUBLES);   // [rows][4] (refinement only)
  int* sInt = reinterpret_cast<int*>(sRegion + region_doubles(a.hyps));   // [8]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const FrameParam fp = a.fp[frame];
  const Cam k{(double)fp.focal, (double)fp.focal, (double)fp.ppx, (double)fp.ppy};
  const float* sc = a.sc + (size_t)frame * 3 * N;
  const uint32_t magic = a.h_magic;
  const int half = a.sub / 2;

  // ---- load: memory order i = y*W + x  ->  scan order p = x*H + y
  for (int i = tid; i < N; i += 256) {
    const int y = i / W, x = i - y * W;
    const int p = x * H + y;
    sX[p] = sc[i];
    sY[p] = sc[N + i];
    sZ[p] = sc[2 * N + i];
  }
  __syncthreads();

  // ---- sample + score: hypothesis h on wavefront h % 4
  const float inlierBeta = 5 / a.thr;
  const float score_scale = a.alpha / (float)W / (float)H;
  // Sampling (dsacstar_util.h:135-221): the sequential loop keeps, per hypothesis, the FIRST try whose P3P pose passes the
  // 4-point check. Tries are independent draws of the counter-based stream, so they are evaluated in parallel and the first
  // accepted one is picked with a ballot: a wavefront works on 8 of its hypotheses at once, 8 tries each (lane = 8 * slot +
  // try); almost every hypothesis is settled by the first batch, so the whole workgroup needs ONE P3P latency for 32
  // hypotheses instead of eight. Hypothesis h lives on wavefront h % 4 (as does its scoring below).
  for (int pass = 0; wave + 4 * (8 * pass) < a.hyps; ++pass) {
    const int slot = lane >> 3, tr = lane & 7;
    const int h = wave + 4 * (8 * pass + slot);
    const bool hvalid = h < a.hyps;
    bool settled = !hvalid;   // of this lane's slot
    for (int t0 = 0; t0 < a.max_tries; t0 += 8) {
      if (__ballot(!settled) == 0ull) break;
      const int t = t0 + tr;
      Pose cur;
#pragma unroll
      for (int i = 0; i < 3; ++i) cur.r[i] = cur.t[i] = 0;
      int status = 0;  // 0: PnP failed (zero pose), 1: solved but rejected by the 4-point check, 2: accepted
      if (!settled && t < a.max_tries) {
        const uint64_t key = rsm::try_key(a.seed, fp.frame_id, (uint32_t)h, (uint32_t)t);
        float obj[4][3], img[4][2];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int x = rsm::irand(key, 2 * j, W);
          const int y = rsm::irand(key, 2 * j + 1, H);
          img[j][0] = (float)(x * a.sub + half);
          img[j][1] = (float)(y * a.sub + half);
          const int p = x * H + y;
          obj[j][0] = sX[p];
          obj[j][1] = sY[p];
          obj[j][2] = sZ[p];
        }
        if (rsm::solve_pnp_p3p(obj, img, k, &cur)) {
          status = 2;
          double R[9];
          rsm::rodrigues(cur.r, R, nullptr);
          for (int j = 0; j < 4; j++) {
            double u, v;
            rsm::project(R, cur.t, k, obj[j][0], obj[j][1], obj[j][2], &u, &v, nullptr, nullptr, nullptr);
            const float dx = img[j][0] - (float)u, dy = img[j][1] - (float)v;
            if (sqrt((double)dx * dx + (double)dy * dy) < (double)a.thr) continue;
            status = 1;
            break;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 3; ++i) cur.r[i] = cur.t[i] = 0;
        }
      }
      const unsigned long long ok = __ballot(status == 2);
      if (!settled) {
        const unsigned g = (unsigned)(ok >> (slot * 8)) & 0xffu;
        int src = -1;
        if (g) src = __ffs((int)g) - 1;                                  // first accepted try of this batch
        else if (t0 + 8 >= a.max_tries) src = a.max_tries - 1 - t0;      // every try failed: the last try's pose stays (:157-220)
        if (src >= 0) {
          settled = true;
          if (tr == src) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              sHyp[h * 6 + i] = cur.r[i];
              sHyp[h * 6 + 3 + i] = cur.t[i];
            }
          }
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();   // the poses of this wave's hypotheses are written by its own lanes
  for (int h = wave; h < a.hyps; h += 4) {
    Pose res;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      res.r[i] = sHyp[h * 6 + i];
      res.t[i] = sHyp[h * 6 + 3 + i];
    }
    // score this hypothesis
    double R[9];
    rsm::rodrigues(res.r, R, nullptr);
    double acc = 0;
    for (int p = lane; p < N; p += 64) {
      const int x = div_h(p, H, magic), y = p - x * H;
      const float e = pixel_err(R, res.t, k, sX[p], sY[p], sZ[p], x * a.sub + half, y * a.sub + half, a.max_reproj);
      const float beta_e = inlierBeta * (e - a.thr);
      // beyond 40 the sigmoid is exactly 1 in fp64 (exp(-40) = 4e-18 < 2^-53, so 1 + exp(-x) == 1) and the pixel adds exactly +0:
      // far outliers -- nearly every pixel of a wrong hypothesis -- skip the exponential and the division
      if (beta_e > 40.f) continue;
      double softThreshold = beta_e;
      softThreshold = 1 / (1 + exp_dev(-softThreshold));
      acc += 1 - softThreshold;
    }
    double score = wave_tree(acc);
    score *= score_scale;
    if (lane == 0) {
      sScores[h] = score;
      double* hp = a.hyp_poses + ((size_t)frame * a.hyps + h) * 6;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        hp[i] = res.r[i];
        hp[3 + i] = res.t[i];
      }
      a.scores[(size_t)frame * a.hyps + h] = score;
    }
  }
  __syncthreads();

  // ---- select: softMax + draw(argmax)
  if (tid == 0) {
    double maxScore = 0;
    for (int i = 0; i < a.hyps; i++)
      if (i == 0 || sScores[i] > maxScore) maxScore = sScores[i];
    double sum = 0.0;
    for (int i = 0; i < a.hyps; i++) sum += detm::exp_(sScores[i] - maxScore);
    double maxProb = -1;
    int maxIdx = 0;
    for (int idx = 0; idx < a.hyps; idx++) {
      const double pr = detm::exp_(sScores[idx] - maxScore) / sum;
      if (pr < 0.00000001) continue;
      if (maxProb < 0 || pr > maxProb) {
        maxProb = pr;
        maxIdx = idx;
      }
    }
    sInt[0] = maxIdx;
    a.best[frame] = maxIdx;
  }
  __syncthreads();

  // ---- refine
  const int bestIdx = sInt[0];
  double param[6];
  {
    const double* hp = a.hyp_poses + ((size_t)frame * a.hyps + bestIdx) * 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) param[i] = hp[i];
  }
  const int rows = (N + 255) >> 8;  // thread tid looks at pixels p = tid + 256 i, i < rows (p < N)

  // bit i of the result: pixel tid + 256 i is an inlier of prm (getReproErrs + the threshold test of refineHyp, :536-560)
  auto classify = [&](const double* prm) -> uint64_t {
    double R[9];
    rsm::rodrigues(prm, R, nullptr);
    uint64_t flags = 0;
    for (int i = 0; i < rows; ++i) {
      const int p = tid + 256 * i;
      if (p >= N) break;
      const int x = div_h(p, H, magic), y = p - x * H;
      const float e = pixel_err(R, prm + 3, k, sX[p], sY[p], sZ[p], x * a.sub + half, y * a.sub + half, a.max_reproj);
      if (e < a.thr) flags |= 1ull << i;
    }
    return flags;
  };
  // the inliers in scan order (ascending p): counts per (row, wavefront) through LDS, positions from ballots. Returns their number.
  auto compact = [&](uint64_t flags) -> int {
    for (int i = 0; i < rows; ++i) {
      const unsigned long long m = __ballot((flags >> i) & 1ull);
      if (lane == 0) sCnt[i * 4 + wave] = __popcll(m);
    }
    __syncthreads();
    int run = 0;
    for (int i = 0; i < rows; ++i) {
      const bool f = (flags >> i) & 1ull;
      const unsigned long long m = __ballot(f);
      const int4 c = reinterpret_cast<const int4*>(sCnt)[i];
      const int base = run + (wave > 0 ? c.x : 0) + (wave > 1 ? c.y : 0) + (wave > 2 ? c.z : 0);
      const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      if (f) sList[base + below] = (uint16_t)(tid + 256 * i);
      run += (c.x + c.y) + (c.z + c.w);
    }
    __syncthreads();
    return run;
  };
  int red_parity = 0;
  // sums over the workgroup in the canonical order (oracle tree256); the 27 Jacobian sums are only read by wavefront 0, the one
  // that solves for the step; |e|^2 drives the control flow of every thread
  auto reduce28 = [&](double* acc, LMAccum* out) {
    double* buf = sRed + red_parity * (4 * 28);
    double r7[7];
    wave_tree28(acc, r7);
    if ((lane & 15) == 0) {
#pragma unroll
      for (int j = 0; j < 7; ++j) buf[wave * 28 + 7 * (lane >> 4) + j] = r7[j];
    }
    __syncthreads();
    if (wave == 0) {
      double s[27];
#pragma unroll
      for (int i = 0; i < 27; ++i) s[i] = ((buf[0 * 28 + i] + buf[1 * 28 + i]) + buf[2 * 28 + i]) + buf[3 * 28 + i];
      int q = 0;
#pragma unroll
      for (int aa = 0; aa < 6; ++aa)
#pragma unroll
        for (int bb = aa; bb < 6; ++bb) {
          out->JtJ[aa * 6 + bb] = s[q];
          out->JtJ[bb * 6 + aa] = s[q];
          ++q;
        }
#pragma unroll
      for (int aa = 0; aa < 6; ++aa) out->JtErr[aa] = s[21 + aa];
    }
    out->errsq = ((buf[0 * 28 + 27] + buf[1 * 28 + 27]) + buf[2 * 28 + 27]) + buf[3 * 28 + 27];
    red_parity ^= 1;
  };
  auto reduce1 = [&](double e) -> double {
    double* buf = sRed + red_parity * (4 * 28);
    const double v = wave_tree(e);
    if (lane == 0) buf[wave * 28 + 27] = v;
    __syncthreads();
    red_parity ^= 1;
    return ((buf[0 * 28 + 27] + buf[1 * 28 + 27]) + buf[2 * 28 + 27]) + buf[3 * 28 + 27];
  };
  // J^T J, J^T e and |e|^2 over the inlier list
  auto accumulate_J = [&](const double* prm, int cnt, LMAccum* out) {
    double R[9], dRdr[27];
    rsm::rodrigues(prm, R, dRdr);
    double acc[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = 0;
    for (int j = tid; j < cnt; j += 256) {
      const int p = sList[j];
      const int x = div_h(p, H, magic), y = p - x * H;
      double u, v, Ju[6], Jv[6];
      rsm::project(R, prm + 3, k, sX[p], sY[p], sZ[p], &u, &v, dRdr, Ju, Jv);
      const double eu = u - (double)(float)(x * a.sub + half), ev = v - (double)(float)(y * a.sub + half);
      int q = 0;
#pragma unroll
      for (int aa = 0; aa < 6; ++aa)
#pragma unroll
        for (int bb = aa; bb < 6; ++bb) {
          acc[q] = acc[q] + Ju[aa] * Ju[bb];
          acc[q] = acc[q] + Jv[aa] * Jv[bb];
          ++q;
        }
#pragma unroll
      for (int aa = 0; aa < 6; ++aa) {
        acc[21 + aa] = acc[21 + aa] + Ju[aa] * eu;
        acc[21 + aa] = acc[21 + aa] + Jv[aa] * ev;
      }
      acc[27] = acc[27] + eu * eu;
      acc[27] = acc[27] + ev * ev;
    }
    reduce28(acc, out);
  };
  // |e|^2 only (CvLevMarq's CHECK_ERR state)
  auto accumulate_E = [&](const double* prm, int cnt) -> double {
    double R[9];
    rsm::rodrigues(prm, R, nullptr);
    double acc = 0;
    for (int j = tid; j < cnt; j += 256) {
      const int p = sList[j];
      const int x = div_h(p, H, magic), y = p - x * H;
      double u, v;
      rsm::project(R, prm + 3, k, sX[p], sY[p], sZ[p], &u, &v, nullptr, nullptr, nullptr);
      const double eu = u - (double)(float)(x * a.sub + half), ev = v - (double)(float)(y * a.sub + half);
      acc = acc + eu * eu;
      acc = acc + ev * ev;
    }
    return reduce1(acc);
  };
  auto lm_step = [&](const LMAccum& acc, const double* prevParam, int lambdaLg10, double* prm) {
    // The 6x6 solve is serial fp64 work with identical inputs on every lane (Cholesky: ~2 us; its eigen fallback ~55 us): only
    // wavefront 0 runs it and hands the step over through LDS, which leaves the other three SIMDs to the second workgroup of
    // this CU (every lm_step is followed by an accumulation, whose block reduction orders the next overwrite of sStep).
    double* sStep = sRegion;   // the scores / sampled poses are dead after the selection
    if (wave == 0) {
      const double lambda = detm::pow10i(lambdaLg10);
      double A[36], x[6];
      for (int i = 0; i < 36; ++i) A[i] = acc.JtJ[i];
      for (int i = 0; i < 6; ++i) A[i * 6 + i] *= 1. + lambda;
      rsm::solve_normal6(A, acc.JtErr, x);
      if (lane == 0)
        for (int i = 0; i < 6; ++i) sStep[i] = x[i];
    }
    __syncthreads();
    for (int i = 0; i < 6; ++i) prm[i] = prevParam[i] - sStep[i];
  };

  uint64_t flags = classify(param), acc_flags = 0;
  unsigned bestInliers = 4;
  bool have_map = false;
  const int max_ref = a.max_ref_steps > 0 ? a.max_ref_steps : 100;
  for (int rStep = 0; rStep < max_ref; rStep++) {
    const int cnt = compact(flags);
    if ((unsigned)cnt <= bestInliers) break;
    bestInliers = (unsigned)cnt;

    // solvePnP(ITERATIVE, useExtrinsicGuess) on the inlier list
    {
      const int max_iter = 20;
      const double epsilon = 1.1920928955078125e-07;
      double prm[6], prevParam[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) prm[i] = param[i];
      int lambdaLg10 = -3, iters = 0;
      double prevErrNorm = 1.7976931348623157e308, errNorm;
      LMAccum acc;
      accumulate_J(prm, cnt, &acc);
      for (;;) {
        for (int i = 0; i < 6; ++i) prevParam[i] = prm[i];
        lm_step(acc, prevParam, lambdaLg10, prm);
        if (iters == 0) prevErrNorm = sqrt(acc.errsq);
        for (;;) {
          errNorm = sqrt(accumulate_E(prm, cnt));
          if (errNorm > prevErrNorm) {
            if (++lambdaLg10 <= 16) {
              lm_step(acc, prevParam, lambdaLg10, prm);
              continue;
            }
          }
          break;
        }
        lambdaLg10 = (lambdaLg10 - 1 > -16) ? lambdaLg10 - 1 : -16;
        double dn = 0, pn = 0;
        for (int i = 0; i < 6; ++i) {
          const double d = prm[i] - prevParam[i];
          dn += d * d;
          pn += prevParam[i] * prevParam[i];
        }
        const double rel = sqrt(dn) / (sqrt(pn) + 2.220446049250313e-16);
        if (++iters >= max_iter || rel < epsilon) break;
        prevErrNorm = errNorm;
        accumulate_J(prm, cnt, &acc);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) param[i] = prm[i];
    }
    acc_flags = flags;
    have_map = true;
    flags = classify(param);
  }

  // ---- outputs
  if (a.out_masks) {
    uint8_t* mk = a.out_masks + (size_t)frame * N;
    for (int i = 0; i < rows; ++i) {
      const int p = tid + 256 * i;
      if (p >= N) break;
      const int x = div_h(p, H, magic), y = p - x * H;
      mk[y * W + x] = (have_map && ((acc_flags >> i) & 1ull)) ? 1 : 0;
    }
  }
  if (tid == 0) {
    double R[9], T[16], Ti[16];
    rsm::rodrigues(param, R, nullptr);
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1. : 0.;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T[r * 4 + c] = R[r * 3 + c];
      T[r * 4 + 3] = param[3 + r];
    }
    if (!inv4x4(T, Ti))
      for (int i = 0; i < 16; ++i) Ti[i] = 0;
    for (int i = 0; i < 16; ++i) a.out_poses[(size_t)frame * 16 + i] = (float)Ti[i];
    for (int i = 0; i < 6; ++i) a.refined[(size_t)frame * 6 + i] = param[i];
    a.out_inliers[frame] = have_map ? (int)bestInliers : 0;
  }
}

}  // namespace

// ====================================================================================================
// C ABI
// ====================================================================================================
namespace {
constexpr int PARAM_SLOTS = 4;
// One pinned + device copy of the per-frame parameter block per call in flight: acez_register_rgb_device never waits for the
// stream it launches on (only for the call PARAM_SLOTS launches ago, whose kernel has long finished in any pipelined use).
struct ParamSlot {
  FrameParam* h = nullptr;  // pinned
  FrameParam* d = nullptr;
  hipEvent_t done = nullptr;
  bool in_flight = false;
};
}  // namespace

struct acez_ransac {
  int device = 0;
  int max_frames = 0, max_h = 0, max_w = 0, max_hyps = 0;
  ParamSlot slot[PARAM_SLOTS];
  int next_slot = 0;
  double* d_hyp_poses = nullptr;
  double* d_scores = nullptr;
  int* d_best = nullptr;
  double* d_refined = nullptr;
  float* d_big = nullptr;      // scan-order copies of frames that do not fit the LDS, allocated on first use
  size_t big_floats = 0;
  // staging for the host-buffer entry point
  float* d_sc = nullptr;
  float* d_pose = nullptr;
  int* d_inl = nullptr;
  uint8_t* d_mask = nullptr;
  int last_hyps = 0;
};

static int ensure_hyps(acez_ransac* ctx, int hyps) {
  if (hyps <= ctx->max_hyps) return ACEZ_OK;
  ACEZ_HIP_CHECK(hipDeviceSynchronize());   // earlier launches may still write the old buffers
  if (ctx->d_hyp_poses) (void)hipFree(ctx->d_hyp_poses);
  if (ctx->d_scores) (void)hipFree(ctx->d_scores);
  ctx->d_hyp_poses = nullptr;
  ctx->d_scores = nullptr;
  ACEZ_HIP_CHECK(hipMalloc((void**)&ctx->d_hyp_poses, (size_t)ctx->max_frames * hyps * 6 * sizeof(double)));
  ACEZ_HIP_CHECK(hipMalloc((void**)&ctx->d_scores, (size_t)ctx->max_frames * hyps * sizeof(double)));
  ctx->max_hyps = hyps;
  return ACEZ_OK;
}

extern "C" void acez_ransac_destroy(acez_ransac* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (ParamSlot& s : ctx->slot) {
    if (s.d) (void)hipFree(s.d);
    if (s.h) (void)hipHostFree(s.h);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  if (ctx->d_hyp_poses) (void)hipFree(ctx->d_hyp_poses);
  if (ctx->d_scores) (void)hipFree(ctx->d_scores);
  if (ctx->d_best) (void)hipFree(ctx->d_best);
  if (ctx->d_refined) (void)hipFree(ctx->d_refined);
  if (ctx->d_big) (void)hipFree(ctx->d_big);
  if (ctx->d_sc) (void)hipFree(ctx->d_sc);
  if (ctx->d_pose) (void)hipFree(ctx->d_pose);
  if (ctx->d_inl) (void)hipFree(ctx->d_inl);
  if (ctx->d_mask) (void)hipFree(ctx->d_mask);
  delete ctx;
}

extern "C" int acez_ransac_create(acez_ransac** out, int max_frames, int max_h, int max_w, int device) {
  ACEZ_REQUIRE(out, "null pointer");
  ACEZ_REQUIRE(max_frames > 0 && max_h > 0 && max_w > 0, "sizes must be positive");
  if ((int64_t)max_h * max_w > 256 * MAX_ROWS) {
    acez::set_error("%d x %d = %lld scene coordinates per frame: the DSAC* kernel handles at most %d (e.g. 96 x 170, a 768 x 1360 image "
                    "at output stride 8); lower the image resolution (ace_zero.py --image_resolution, dataset.py:40) -- the reference's "
                    "CPU loop has no such limit", max_h, max_w, (long long)max_h * max_w, 256 * MAX_ROWS);
    return ACEZ_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    acez::set_error("no HIP device visible: the DSAC* kernels need a gfx950 GPU (there is no CPU fallback)");
    return ACEZ_ERR_NODEVICE;
  }
  if (device >= 0) ACEZ_HIP_CHECK(hipSetDevice(device));
  acez_ransac* ctx = new (std::nothrow) acez_ransac();
  ACEZ_REQUIRE(ctx, "out of host memory");
  ACEZ_HIP_CHECK(hipGetDevice(&ctx->device));
  ctx->max_frames = max_frames;
  ctx->max_h = max_h;
  ctx->max_w = max_w;
  int rc = ACEZ_OK;
  auto A = [&](void** p, size_t bytes) {
    if (rc == ACEZ_OK && hipMalloc(p, bytes) != hipSuccess) {
      acez::set_error("hipMalloc(%zu) failed", bytes);
      rc = ACEZ_ERR_HIP;
    }
  };
  for (ParamSlot& s : ctx->slot) {
    A((void**)&s.d, (size_t)max_frames * sizeof(FrameParam));
    if (rc == ACEZ_OK && (hipHostMalloc((void**)&s.h, (size_t)max_frames * sizeof(FrameParam)) != hipSuccess ||
                          hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess)) {
      acez::set_error("hipHostMalloc / hipEventCreate failed");
      rc = ACEZ_ERR_HIP;
    }
  }
  A((void**)&ctx->d_best, (size_t)max_frames * sizeof(int));
  A((void**)&ctx->d_refined, (size_t)max_frames * 6 * sizeof(double));
  A((void**)&ctx->d_sc, (size_t)3 * max_h * max_w * sizeof(float));
  A((void**)&ctx->d_pose, 16 * sizeof(float));
  A((void**)&ctx->d_inl, sizeof(int));
  A((void**)&ctx->d_mask, (size_t)max_h * max_w);
  if (rc == ACEZ_OK) rc = ensure_hyps(ctx, 64);
  if (rc != ACEZ_OK) {
    acez_ransac_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return ACEZ_OK;
}

extern "C" int acez_register_rgb_device(acez_ransac* ctx, const float* d_scene_coords, int n_frames, int h, int w,
                                        const acez_ransac_params* params, const acez_intrinsics* h_intrinsics, uint64_t seed,
                                        const uint64_t* h_frame_ids, float* d_out_poses, int32_t* d_out_inliers,
                                        uint8_t* d_out_masks, void* stream) {
  ACEZ_REQUIRE(ctx && d_scene_coords && params && h_intrinsics && d_out_poses && d_out_inliers, "null pointer");
  ACEZ_REQUIRE(n_frames > 0 && n_frames <= ctx->max_frames, "n_frames exceeds the context's max_frames");
  ACEZ_REQUIRE(h > 0 && w > 0 && h <= ctx->max_h && w <= ctx->max_w, "frame larger than the context was created for");
  ACEZ_REQUIRE((int64_t)h * w <= 256 * MAX_ROWS, "at most 16384 scene coordinates per frame");
  ACEZ_REQUIRE(params->hypotheses > 0 && params->max_tries > 0, "hypotheses and max_tries must be positive");
  ACEZ_REQUIRE(params->subsampling > 0 && params->inlier_threshold > 0.f, "subsampling and inlier_threshold must be positive");
  ACEZ_HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  int rc = ensure_hyps(ctx, params->hypotheses);
  if (rc != ACEZ_OK) return rc;
  const int N = h * w;
  const int Npad = (N + 3) & ~3;
  const size_t lds_full = lds_bytes(N, params->hypotheses, false);
  const bool gc = lds_full > 160 * 1024;
  const size_t lds = gc ? lds_bytes(N, params->hypotheses, true) : lds_full;
  ACEZ_REQUIRE(lds <= 160 * 1024, "too many hypotheses for the 160 KB LDS of a CU");
  if (gc && ctx->big_floats < (size_t)ctx->max_frames * 3 * Npad) {
    ACEZ_HIP_CHECK(hipDeviceSynchronize());
    if (ctx->d_big) (void)hipFree(ctx->d_big);
    ctx->d_big = nullptr;
    ctx->big_floats = 0;
    ACEZ_HIP_CHECK(hipMalloc((void**)&ctx->d_big, (size_t)ctx->max_frames * 3 * Npad * sizeof(float)));
    ctx->big_floats = (size_t)ctx->max_frames * 3 * Npad;
  }
  ParamSlot& slot = ctx->slot[ctx->next_slot];
  ctx->next_slot = (ctx->next_slot + 1) % PARAM_SLOTS;
  if (slot.in_flight) ACEZ_HIP_CHECK(hipEventSynchronize(slot.done));
  for (int i = 0; i < n_frames; ++i) {
    slot.h[i].focal = h_intrinsics[i].focal;
    slot.h[i].ppx = h_intrinsics[i].ppx;
    slot.h[i].ppy = h_intrinsics[i].ppy;
    slot.h[i].pad = 0.f;
    slot.h[i].frame_id = h_frame_ids ? h_frame_ids[i] : (uint64_t)i;
  }
  ACEZ_HIP_CHECK(hipMemcpyAsync(slot.d, slot.h, (size_t)n_frames * sizeof(FrameParam), hipMemcpyHostToDevice, s));
  RansacArgs a;
  a.sc = d_scene_coords; a.fp = slot.d; a.big = ctx->d_big; a.H = h; a.W = w; a.N = N; a.hyps = params->hypotheses;
  a.h_magic = h > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)h - 1) / (uint64_t)h) : 0u;
  a.max_tries = params->max_tries; a.sub = params->subsampling; a.max_ref_steps = params->max_ref_steps;
  a.thr = params->inlier_threshold; a.alpha = params->inlier_alpha; a.max_reproj = params->max_reproj; a.seed = seed;
  a.hyp_poses = ctx->d_hyp_poses; a.scores = ctx->d_scores; a.best = ctx->d_best; a.refined = ctx->d_refined;
  a.out_poses = d_out_poses; a.out_inliers = d_out_inliers; a.out_masks = d_out_masks;
  if (gc) {
    ACEZ_HIP_CHECK(hipFuncSetAttribute((const void*)ransac_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ransac_kernel<true>, dim3(n_frames), dim3(256), lds, s, a);
  } else {
    ACEZ_HIP_CHECK(hipFuncSetAttribute((const void*)ransac_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ransac_kernel<false>, dim3(n_frames), dim3(256), lds, s, a);
  }
  ACEZ_HIP_CHECK(hipGetLastError());
  ACEZ_HIP_CHECK(hipEventRecord(slot.done, s));
  slot.in_flight = true;
  ctx->last_hyps = params->hypotheses;
  return ACEZ_OK;
}

extern "C" int acez_register_rgb_host(acez_ransac* ctx, const float* h_scene_coords, int64_t stride_c, int64_t stride_h,
                                      int64_t stride_w, int h, int w, const acez_ransac_params* params,
                                      const acez_intrinsics* intr, uint64_t seed, uint64_t frame_id, float* h_out_pose16,
                                      int32_t* out_inliers, uint8_t* h_out_mask) {
  ACEZ_REQUIRE(ctx && h_scene_coords && params && intr && h_out_pose16 && out_inliers, "null pointer");
  ACEZ_REQUIRE(h > 0 && w > 0 && h <= ctx->max_h && w <= ctx->max_w, "frame larger than the context was created for");
  ACEZ_HIP_CHECK(hipSetDevice(ctx->device));
  // honour the accessor strides of the caller's tensor (dsacstar.cpp:83-84) while packing to [3][h][w]
  std::vector<float> packed((size_t)3 * h * w);
  for (int c = 0; c < 3; ++c)
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) packed[((size_t)c * h + y) * w + x] = h_scene_coords[c * stride_c + y * stride_h + x * stride_w];
  ACEZ_HIP_CHECK(hipMemcpy(ctx->d_sc, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
  int rc = acez_register_rgb_device(ctx, ctx->d_sc, 1, h, w, params, intr, seed, &frame_id, ctx->d_pose, ctx->d_inl,
                                    h_out_mask ? ctx->d_mask : nullptr, nullptr);
  if (rc != ACEZ_OK) return rc;
  ACEZ_HIP_CHECK(hipMemcpy(h_out_pose16, ctx->d_pose, 16 * sizeof(float), hipMemcpyDeviceToHost));
  ACEZ_HIP_CHECK(hipMemcpy(out_inliers, ctx->d_inl, sizeof(int), hipMemcpyDeviceToHost));
  if (h_out_mask) ACEZ_HIP_CHECK(hipMemcpy(h_out_mask, ctx->d_mask, (size_t)h * w, hipMemcpyDeviceToHost));
  return ACEZ_OK;
}

extern "C" int acez_ransac_debug_fetch(acez_ransac* ctx, int n_frames, int hypotheses, double* h_hyp_poses, double* h_scores,
                                       int32_t* h_best, double* h_refined) {
  ACEZ_REQUIRE(ctx, "null context");
  ACEZ_REQUIRE(n_frames > 0 && n_frames <= ctx->max_frames && hypotheses == ctx->last_hyps, "shape does not match the last call");
  ACEZ_HIP_CHECK(hipSetDevice(ctx->device));
  ACEZ_HIP_CHECK(hipDeviceSynchronize());
  if (h_hyp_poses) ACEZ_HIP_CHECK(hipMemcpy(h_hyp_poses, ctx->d_hyp_poses, (size_t)n_frames * hypotheses * 6 * sizeof(double), hipMemcpyDeviceToHost));
  if (h_scores) ACEZ_HIP_CHECK(hipMemcpy(h_scores, ctx->d_scores, (size_t)n_frames * hypotheses * sizeof(double), hipMemcpyDeviceToHost));
  if (h_best) ACEZ_HIP_CHECK(hipMemcpy(h_best, ctx->d_best, (size_t)n_frames * sizeof(int), hipMemcpyDeviceToHost));
  if (h_refined) ACEZ_HIP_CHECK(hipMemcpy(h_refined, ctx->d_refined, (size_t)n_frames * 6 * sizeof(double), hipMemcpyDeviceToHost));
  return ACEZ_OK;
}
