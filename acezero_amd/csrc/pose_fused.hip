// pose_fused.hip -- the pose-refinement network (pose_kernels.hip; refine_poses.py:152-176,212-244) folded into launches the
// training step makes anyway (--pose_refinement mlp: every non-seed mapping round of ace_zero.py, ace_zero.py:86).
//
// As its own launches -- even on a second stream -- the network cost 72 us on top of a 166 us step (BENCH_r02: 237.7 vs 165.7 us):
// six small dependent kernels of 12-18 us each (63 workgroups at 1000 images) whose latency nothing hid, because the head's GEMM
// workgroups own the LDS of every CU while they run and a second hardware queue makes the kernel boundaries of both queues dearer.
// The dependency structure allows better:
//
//   pose forward  (S3)  needs the pose parameters of the previous step, feeds the loss kernel      -> beside the batch gather
//   reduce + backward chain (S1) needs the loss kernel's per-row pose gradients                     -> beside the head's AdamW
//   weight gradients + AdamW (S2) needs S1 of ALL image tiles (a grid-wide dependency)              -> one launch after it
//
// "Beside" = extra workgroups at the FRONT of the same launch (lowest block indices: they are dispatched first and are the
// long pole). Only launches whose own workgroups are small qualify -- the gather (no LDS) and the optimiser (8 KiB, HBM-bound): a
// kernel's LDS and register allocation is the same for all of its workgroups, so the GEMM launches (112-154 KiB) cannot host them.
// The pose workgroups are unchanged bodies of pose_kernels.hip; the per-image results are the same sums in the same order.
#include "pose_small.hip"

namespace acez {

// T = images per pose workgroup: 16 = the v_mfma_f32_16x16x4_f32 bodies of pose_kernels.hip (256 threads), 8 / 4 = the small tiles of
// pose_small.hip (T / 16 of the matrix time per layer, 16 / T times as many CUs, 512 threads; the default is 4). A launch's workgroups
// all have pose_threads<T>() threads: in the launches shared with the optimiser its workgroups use the first 256 (the other waves
// return at once), the gather's rows are spread over the waves there are.
template <int T>
constexpr int pose_threads() { return 256; }              // the launches that hold S1 (PN4_W_BWD = 4 waves)
template <int T>
constexpr int pose_fwd_threads() { return T == 16 ? 256 : 64 * PN4_W_FWD; }
template <int T>
constexpr int pose_fwd_smem_floats() { return T == 16 ? 4 : pose4_fwd_smem_floats<(T == 16 ? 8 : T)>(); }
template <int T>
constexpr int pose_s1_smem_bytes() { return T == 16 ? PS1_SMEM_BYTES : pose4_s1_smem_bytes<(T == 16 ? 8 : T)>(); }

// step_begin + S3: blocks [0, np) = pose forward tiles; then the gather blocks; the last block = the schedule bookkeeping that
// closes the previous iteration (do_post) -- exactly step_begin_kernel. The pose forward does not look at st->active (the schedule
// block of this very launch is rewriting it): refined poses computed for a step that turns out to be inactive are never used.
template <int T>
__global__ __launch_bounds__(pose_fwd_threads<T>(), T == 16 ? 2 : 1) void step_begin_pose_kernel(const uint16_t* __restrict__ feat, const int64_t* __restrict__ idx,
                                                                 uint16_t* __restrict__ out, int n, PostArgs p, int do_post, PoseNetArgs a, int np,
                                                                 GatherMeta meta) {
  __shared__ __attribute__((aligned(16))) float smem[pose_fwd_smem_floats<T>()];
  if ((int)blockIdx.x < np) {
    if constexpr (T == 16) pose_mlp_fwd_body(a, blockIdx.x);
    else pose4_fwd_body<T>(a, blockIdx.x, smem);
    return;
  }
  const int gb = (int)blockIdx.x - np, ngb = (int)gridDim.x - np - 1;
  if (gb == ngb) {
    if (do_post && threadIdx.x < 64) sched_post_wave(p.src, p.st, p.c, p.grad_stats, p.inv_global_batch, p.log_loss, p.log_inl, p.log_cap, p.fault, p.stat_partials, p.n_loss_blocks);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = (gb * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (ngb * blockDim.x) >> 6;
  gather_rows(feat, idx, out, n, wave, nwaves, lane, meta);
}

// S1 of one tile (reduction of the per-row pose gradients + compose backward + input-gradient chain)
template <int T>
__device__ __forceinline__ void pose_s1_tile(const PoseNetArgs& pn, const float* row_dT, const int* row_image, int n, int tile, char* smem) {
  if constexpr (T == 16) pose_s1_body(pn, row_dT, row_image, n, tile, smem);
  else pose4_s1_body<T>(pn, row_dT, row_image, n, tile, smem);
}
template <int T>
__global__ __launch_bounds__(pose_threads<T>()) void pose_s1t_kernel(PoseNetArgs a, const float* row_dT, const int* row_image, int n) {
  if (a.active && !*a.active) return;
  __shared__ __attribute__((aligned(16))) char smem[pose_s1_smem_bytes<T>()];
  pose_s1_tile<T>(a, row_dT, row_image, n, blockIdx.x, smem);
}
// stand-alone forward on small tiles (acez_trainer_get_poses, the split flow)
template <int T>
__global__ __launch_bounds__(pose_fwd_threads<T>()) void pose_fwd_t_kernel(PoseNetArgs a) {
  if (a.active && !*a.active) return;
  __shared__ __attribute__((aligned(16))) float smem[pose_fwd_smem_floats<T>()];
  if constexpr (T == 16) pose_mlp_fwd_body(a, blockIdx.x);
  else pose4_fwd_body<T>(a, blockIdx.x, smem);
}

// AdamW of the head + S1: blocks [0, np) = reduce + backward chain of one image tile each, the rest = adamw_kernel's blocks.
// Register budget: 3 waves per SIMD (168 VGPRs, one spilled dword in S1) = three 256-thread workgroups per CU for the launch's 859; as
// compiled for 2 (170 VGPRs) it was 21.8 us, for 3 it is 20.2 (same box, two alternations; the step 164.5 -> 163.8 us); for 4 (128 VGPRs)
// S1 spills 120 dwords and the launch takes 52 us.
#ifndef ACEZ_APK_WAVES
#define ACEZ_APK_WAVES 3
#endif
template <int T>
__global__ __launch_bounds__(256, ACEZ_APK_WAVES) void adamw_pose_kernel(AdamArgs a, PoseNetArgs pn, const float* row_dT, const int* row_image, int n, int np) {
  constexpr int SMEM = pose_s1_smem_bytes<T>() > 64 * 66 * 2 ? pose_s1_smem_bytes<T>() : 64 * 66 * 2;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  if ((int)blockIdx.x < np) {
    if (pn.active && !*pn.active) return;
    pose_s1_tile<T>(pn, row_dT, row_image, n, blockIdx.x, smem);
    return;
  }
  adamw_body(a, (int)blockIdx.x - np, reinterpret_cast<uint16_t (*)[66]>(smem));
}

// ---- the split flow (backward / all-reduce / update: a data-parallel rank) with pose refinement. As separate launches its tail was
// wgrad -> grad_reduce -> S1 -> S2 -> [all-reduce] -> adamw -> adamw_small -> (next step) pose_transpose: five small dependent launches
// where the fused step has two. S1 only needs the loss kernel's per-row pose gradients and grad_reduce has no LDS of its own, so S1 rides
// at the front of ITS launch; the pose parameters' AdamW and the refresh of the four transposed 128 x 128 copies ride at the front of the
// head's optimiser launch (adamw_small_kernel's arithmetic element by element, pose_transpose_kernel's copies: the same bits).
template <int T>
__global__ __launch_bounds__(256, 3) void grad_reduce_pose_kernel(GradReduceArgs a, PoseNetArgs pn, const float* row_dT, const int* row_image, int n, int np) {
  __shared__ __attribute__((aligned(16))) char smem[pose_s1_smem_bytes<T>()];
  if ((int)blockIdx.x < np) {
    if (pn.active && !*pn.active) return;
    pose_s1_tile<T>(pn, row_dT, row_image, n, blockIdx.x, smem);
    return;
  }
  grad_reduce_body(a, (int)blockIdx.x - np);
}

// blocks: the workgroups (if any) that gather the NEXT batch's rows (acez_train_update_next: the following backward's launch then carries
// the pose forward and the schedule wave only), npb of 256 pose parameters each, n_adam = adamw_kernel's. Wt: the transposed copies
// (null: none kept)
__global__ __launch_bounds__(256, 4) void adamw_split_pose_kernel(AdamArgs a, int npb, float* p, float* m, float* v, const float* g, int64_t n,
                                                               const AdamScalars* sc, const int* enable, const int* active, const int* fault, float* Wt,
                                                               int n_adam, const uint16_t* __restrict__ feat, const int64_t* __restrict__ idx_next,
                                                               uint16_t* __restrict__ out, int n_next, GatherMeta meta) {
  __shared__ uint16_t tileT[64][66];
  // grid order: gather workgroups first (their three dependent load levels start with the launch: adamw_next_kernel), pose parameters, optimiser
  const int ngb = (int)gridDim.x - npb - n_adam;
  int b = (int)blockIdx.x;
  if (b < ngb) {
    gather_rows(feat, idx_next, out, n_next, (b * (int)blockDim.x + (int)threadIdx.x) >> 6, (ngb * (int)blockDim.x) >> 6, threadIdx.x & 63, meta);
    return;
  }
  b -= ngb;
  if (b >= npb) {
    adamw_body(a, b - npb, tileT);
    return;
  }
  if (active && !*active) return;
  if (fault && *fault) return;   // abandoned step (rowseq fault, head_kernels.hip)
  if (enable && !*enable) return;
  const int64_t i = (int64_t)b * 256 + threadIdx.x;
  if (i >= n) return;
  const AdamScalars s = *sc;
  float gv = 0.f;
  gv += g[i];                    // (adamw_small_kernel with one partial vector: 0 + g, the same bits)
  float mv = m[i], vv = v[i];
  const float pn = adamw_small_one(p[i], gv, mv, vv, s);
  p[i] = pn;
  m[i] = mv;
  v[i] = vv;
  if (Wt) {
    const int64_t off[4] = {PN_C2_W, PN_C3_W, PN_F1_W, PN_F2_W};
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int64_t r = i - off[l];
      if (r >= 0 && r < 16384) Wt[(size_t)l * 16384 + (r & 127) * 128 + (r >> 7)] = pn;   // Wt[l][k][n] = W_l[n][k]
    }
  }
}

}  // namespace acez
