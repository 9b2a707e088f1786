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
//               hypothesis from LDS, lane-strided fp64 partial sums, __shfl_xor butterfly (order 32,16,..,1)
//   select      softMax + draw(argmax) (:684-752) by one lane
//   refine      refineHyp (:522-597): all 256 threads; Levenberg-Marquardt PnP (cvFindExtrinsicCameraParams2 +
//               CvLevMarq) with J^T J / J^T e / |e|^2 accumulated per thread over its pixels, reduced by shuffle
//               butterflies inside each wavefront and a fixed-order sum over the 4 wavefronts through LDS; the
//               6x6 solve runs redundantly on every lane (no divergence, no broadcast)
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
  int H, W, N, hyps, max_tries, sub, max_ref_steps;
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

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src); }
__device__ __forceinline__ double wave_tree(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
  return v;
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

constexpr int MAX_PIX_PER_THREAD = 32;  // N <= 8192

__global__ __launch_bounds__(256, 2) void ransac_kernel(RansacArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = a.N, H = a.H, W = a.W;
  const int Npad = (N + 3) & ~3;
  float* sX = reinterpret_cast<float*>(smem_raw);
  float* sY = sX + Npad;
  float* sZ = sY + Npad;
  float* sErr = sZ + Npad;
  // one fp64 region of max(7 * hyps, 232) doubles, used twice: during sampling / scoring / selection it holds the scores [hyps] and
  // the sampled poses [hyps][6]; both are dead once the best hypothesis is chosen, and the refinement reuses the region for the LM
  // step hand-over (6 doubles, padded to 8) and the double-buffered block-reduction scratch [2][4][28]. Keeping the two apart cost
  // 1.8 KB more: 64 hypotheses then exceeded half of the 160 KiB LDS by 576 bytes and ran ONE workgroup per CU (20 ms vs 12 ms).
  double* sRegion = reinterpret_cast<double*>(sErr + Npad);
  double* sScores = sRegion;                              // [hyps]
  double* sHyp = sScores + a.hyps;                        // [hyps][6] sampled poses
  double* sRed = sRegion + 8;                             // [2][4][28] (refinement only)
  const int region = (7 * a.hyps > 232) ? 7 * a.hyps : 232;
  int* sInt = reinterpret_cast<int*>(sRegion + region);   // [8]: best, counts[4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frame = blockIdx.x;
  const FrameParam fp = a.fp[frame];
  const Cam k{(double)fp.focal, (double)fp.focal, (double)fp.ppx, (double)fp.ppy};
  const float* sc = a.sc + (size_t)frame * 3 * N;

  // ---- load: memory order i = y*W + x  ->  LDS order p = x*H + y
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
          img[j][0] = (float)(x * a.sub + a.sub / 2);
          img[j][1] = (float)(y * a.sub + a.sub / 2);
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
      const int x = p / H, y = p - x * H;
      const float e = pixel_err(R, res.t, k, sX[p], sY[p], sZ[p], x * a.sub + a.sub / 2, y * a.sub + a.sub / 2, a.max_reproj);
      double softThreshold = inlierBeta * (e - a.thr);
      softThreshold = 1 / (1 + detm::exp_(-softThreshold));
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
  const int npix = (N - tid + 255) / 256;  // pixels p = tid + 256 i owned by this thread

  auto recompute_errs = [&](const double* prm) {
    double R[9];
    rsm::rodrigues(prm, R, nullptr);
    for (int i = 0; i < npix; ++i) {
      const int p = tid + 256 * i;
      const int x = p / H, y = p - x * H;
      sErr[p] = pixel_err(R, prm + 3, k, sX[p], sY[p], sZ[p], x * a.sub + a.sub / 2, y * a.sub + a.sub / 2, a.max_reproj);
    }
  };
  int red_parity = 0;
  // block reduction of 28 doubles in the canonical order; every thread receives the result
  auto block_reduce28 = [&](double* acc) {
    double* buf = sRed + red_parity * (4 * 28);
#pragma unroll
    for (int i = 0; i < 28; ++i) {
      const double v = wave_tree(acc[i]);
      if (lane == 0) buf[wave * 28 + i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = ((buf[0 * 28 + i] + buf[1 * 28 + i]) + buf[2 * 28 + i]) + buf[3 * 28 + i];
    red_parity ^= 1;
  };
  auto lm_accumulate = [&](const double* prm, unsigned flags, bool withJ, LMAccum* out) {
    double R[9], dRdr[27];
    rsm::rodrigues(prm, R, withJ ? dRdr : nullptr);
    double acc[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = 0;
    for (int i = 0; i < npix; ++i) {
      if (!((flags >> i) & 1u)) continue;
      const int p = tid + 256 * i;
      const int x = p / H, y = p - x * H;
      double u, v, Ju[6], Jv[6];
      rsm::project(R, prm + 3, k, sX[p], sY[p], sZ[p], &u, &v, withJ ? dRdr : nullptr, Ju, Jv);
      const double eu = u - (double)(float)(x * a.sub + a.sub / 2), ev = v - (double)(float)(y * a.sub + a.sub / 2);
      if (withJ) {
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
      }
      acc[27] = acc[27] + eu * eu;
      acc[27] = acc[27] + ev * ev;
    }
    block_reduce28(acc);
    if (withJ) {
      int q = 0;
#pragma unroll
      for (int aa = 0; aa < 6; ++aa)
#pragma unroll
        for (int bb = aa; bb < 6; ++bb) {
          out->JtJ[aa * 6 + bb] = acc[q];
          out->JtJ[bb * 6 + aa] = acc[q];
          ++q;
        }
#pragma unroll
      for (int aa = 0; aa < 6; ++aa) out->JtErr[aa] = acc[21 + aa];
    }
    out->errsq = acc[27];
  };
  auto lm_step = [&](const LMAccum& acc, const double* prevParam, int lambdaLg10, double* prm) {
    // The 6x6 solve is serial fp64 work with identical inputs on every lane (Cholesky: ~2 us; its eigen fallback ~55 us): only wavefront 0 runs it and
    // hands the step over through LDS, which leaves the other three SIMDs to the second workgroup of this CU
    // (every lm_step is followed by an lm_accumulate, whose block reduction orders the next overwrite of sStep).
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

  recompute_errs(param);
  unsigned bestInliers = 4, acc_flags = 0;
  bool have_map = false;
  const int max_ref = a.max_ref_steps > 0 ? a.max_ref_steps : 100;
  for (int rStep = 0; rStep < max_ref; rStep++) {
    unsigned flags = 0;
    int cnt = 0;
    for (int i = 0; i < npix; ++i)
      if (sErr[tid + 256 * i] < a.thr) {
        flags |= 1u << i;
        cnt++;
      }
    // workgroup-wide inlier count (integer: order-free)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    __syncthreads();  // previous readers of sInt[1..4] are done
    if (lane == 0) sInt[1 + wave] = cnt;
    __syncthreads();
    const unsigned total = (unsigned)(sInt[1] + sInt[2] + sInt[3] + sInt[4]);
    if (total <= bestInliers) break;
    bestInliers = total;

    // solvePnP(ITERATIVE, useExtrinsicGuess) on the flagged pixels
    {
      const int max_iter = 20;
      const double epsilon = 1.1920928955078125e-07;
      double prm[6], prevParam[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) prm[i] = param[i];
      int lambdaLg10 = -3, iters = 0;
      double prevErrNorm = 1.7976931348623157e308, errNorm;
      LMAccum acc, tmp;
      lm_accumulate(prm, flags, true, &acc);
      for (;;) {
        for (int i = 0; i < 6; ++i) prevParam[i] = prm[i];
        lm_step(acc, prevParam, lambdaLg10, prm);
        if (iters == 0) prevErrNorm = sqrt(acc.errsq);
        for (;;) {
          lm_accumulate(prm, flags, false, &tmp);
          errNorm = sqrt(tmp.errsq);
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
        lm_accumulate(prm, flags, true, &acc);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) param[i] = prm[i];
    }
    acc_flags = flags;
    have_map = true;
    recompute_errs(param);
  }

  // ---- outputs
  if (a.out_masks) {
    uint8_t* mk = a.out_masks + (size_t)frame * N;
    for (int i = 0; i < npix; ++i) {
      const int p = tid + 256 * i;
      const int x = p / H, y = p - x * H;
      mk[y * W + x] = (have_map && ((acc_flags >> i) & 1u)) ? 1 : 0;
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
struct acez_ransac {
  int device = 0;
  int max_frames = 0, max_h = 0, max_w = 0, max_hyps = 0;
  FrameParam* d_fp = nullptr;
  FrameParam* h_fp = nullptr;  // pinned
  double* d_hyp_poses = nullptr;
  double* d_scores = nullptr;
  int* d_best = nullptr;
  double* d_refined = nullptr;
  // staging for the host-buffer entry point
  float* d_sc = nullptr;
  float* d_pose = nullptr;
  int* d_inl = nullptr;
  uint8_t* d_mask = nullptr;
  int last_hyps = 0;
};

static int ensure_hyps(acez_ransac* ctx, int hyps) {
  if (hyps <= ctx->max_hyps) return ACEZ_OK;
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
  if (ctx->d_fp) (void)hipFree(ctx->d_fp);
  if (ctx->h_fp) (void)hipHostFree(ctx->h_fp);
  if (ctx->d_hyp_poses) (void)hipFree(ctx->d_hyp_poses);
  if (ctx->d_scores) (void)hipFree(ctx->d_scores);
  if (ctx->d_best) (void)hipFree(ctx->d_best);
  if (ctx->d_refined) (void)hipFree(ctx->d_refined);
  if (ctx->d_sc) (void)hipFree(ctx->d_sc);
  if (ctx->d_pose) (void)hipFree(ctx->d_pose);
  if (ctx->d_inl) (void)hipFree(ctx->d_inl);
  if (ctx->d_mask) (void)hipFree(ctx->d_mask);
  delete ctx;
}

extern "C" int acez_ransac_create(acez_ransac** out, int max_frames, int max_h, int max_w, int device) {
  ACEZ_REQUIRE(out, "null pointer");
  ACEZ_REQUIRE(max_frames > 0 && max_h > 0 && max_w > 0, "sizes must be positive");
  ACEZ_REQUIRE((int64_t)max_h * max_w <= 256 * MAX_PIX_PER_THREAD, "at most 8192 scene coordinates per frame");
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
  A((void**)&ctx->d_fp, (size_t)max_frames * sizeof(FrameParam));
  A((void**)&ctx->d_best, (size_t)max_frames * sizeof(int));
  A((void**)&ctx->d_refined, (size_t)max_frames * 6 * sizeof(double));
  A((void**)&ctx->d_sc, (size_t)3 * max_h * max_w * sizeof(float));
  A((void**)&ctx->d_pose, 16 * sizeof(float));
  A((void**)&ctx->d_inl, sizeof(int));
  A((void**)&ctx->d_mask, (size_t)max_h * max_w);
  if (rc == ACEZ_OK && hipHostMalloc((void**)&ctx->h_fp, (size_t)max_frames * sizeof(FrameParam)) != hipSuccess) {
    acez::set_error("hipHostMalloc failed");
    rc = ACEZ_ERR_HIP;
  }
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
  ACEZ_REQUIRE(params->hypotheses > 0 && params->max_tries > 0, "hypotheses and max_tries must be positive");
  ACEZ_REQUIRE(params->subsampling > 0 && params->inlier_threshold > 0.f, "subsampling and inlier_threshold must be positive");
  ACEZ_HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  int rc = ensure_hyps(ctx, params->hypotheses);
  if (rc != ACEZ_OK) return rc;
  // the pinned parameter block is reused by every call: wait for earlier work that may still read it
  ACEZ_HIP_CHECK(hipStreamSynchronize(s));
  for (int i = 0; i < n_frames; ++i) {
    ctx->h_fp[i].focal = h_intrinsics[i].focal;
    ctx->h_fp[i].ppx = h_intrinsics[i].ppx;
    ctx->h_fp[i].ppy = h_intrinsics[i].ppy;
    ctx->h_fp[i].pad = 0.f;
    ctx->h_fp[i].frame_id = h_frame_ids ? h_frame_ids[i] : (uint64_t)i;
  }
  ACEZ_HIP_CHECK(hipMemcpyAsync(ctx->d_fp, ctx->h_fp, (size_t)n_frames * sizeof(FrameParam), hipMemcpyHostToDevice, s));
  RansacArgs a;
  a.sc = d_scene_coords; a.fp = ctx->d_fp; a.H = h; a.W = w; a.N = h * w; a.hyps = params->hypotheses;
  a.max_tries = params->max_tries; a.sub = params->subsampling; a.max_ref_steps = params->max_ref_steps;
  a.thr = params->inlier_threshold; a.alpha = params->inlier_alpha; a.max_reproj = params->max_reproj; a.seed = seed;
  a.hyp_poses = ctx->d_hyp_poses; a.scores = ctx->d_scores; a.best = ctx->d_best; a.refined = ctx->d_refined;
  a.out_poses = d_out_poses; a.out_inliers = d_out_inliers; a.out_masks = d_out_masks;
  const int Npad = (a.N + 3) & ~3;
  const int region = 7 * params->hypotheses > 232 ? 7 * params->hypotheses : 232;   // see the layout comment in ransac_kernel
  const size_t lds = (size_t)4 * Npad * sizeof(float) + (size_t)region * sizeof(double) + 8 * sizeof(int);
  ACEZ_REQUIRE(lds <= 160 * 1024, "frame + hypotheses do not fit the 160 KB LDS of a CU");
  ACEZ_HIP_CHECK(hipFuncSetAttribute((const void*)ransac_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(ransac_kernel, dim3(n_frames), dim3(256), lds, s, a);
  ACEZ_HIP_CHECK(hipGetLastError());
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
