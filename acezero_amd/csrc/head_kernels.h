// head_kernels.h -- argument structs shared by the head kernels and their host-side launcher.
#pragma once
#include <stdint.h>

#include "gemm_common.h"   // ACEZ_DBG

namespace acez {

constexpr int MAX_LAYERS = 20;

enum { AUX_NONE = 0, AUX_RESIDUAL = 1, AUX_UNMASKED = 2 };
enum { LOSS_TANH = 0, LOSS_DYNTANH = 1, LOSS_L1 = 2, LOSS_L1_SQRT = 3, LOSS_L1_LOGL1 = 4 };
enum { SCHED_CONSTANT = 0, SCHED_1CYCLEPOLY = 1, SCHED_CIRCLE = 2 };

struct AdamScalars {
  float decay, one_minus_beta1, beta2, one_minus_beta2, bc2_sqrt, eps, step_size;
};

// Device-resident schedule / optimiser state: the training loop never synchronises the host.
struct TrainState {
  int active;          // this step runs (iteration < max_iterations and no NaN)
  int iteration;       // TrainerACE.iteration
  int max_iterations;  // ScheduleACE.max_iterations
  int in_cooldown;
  int warmup_epoch;    // last_epoch of the warm-up LinearLR / of OneCycleLR
  int cooldown_epoch;  // last_epoch of the cool-down LinearLR
  int nan_flag;
  int opt_steps;       // AdamW state['step']
  int crit_count;      // entries of the cool-down criterion ring in use (<= 100)
  int crit_pos;        // next slot of the ring to overwrite
  int calib_steps;
  float loss_weight;   // soft clamp of this iteration
  float last_loss, last_inliers;
  float crit_buf[100];  // cooldown_criterium_buffer (ace_schedule.py:44,121-125) as a ring: only its minimum is ever used
  double lr;            // param_group['lr']
  double calib_g, calib_m, calib_v;
  double beta1_pow, beta2_pow;  // beta^opt_steps
  AdamScalars adam;
  // pose-refinement network (refine_poses.py): own AdamW
  int pose_enable, pose_opt_steps;
  double pose_b1pow, pose_b2pow;
  AdamScalars pose_adam;
  // fp16 operands: the gradient chain is propagated scaled by grad_scale (a power of two, so that scaling commutes with fp16 rounding
  // wherever nothing under- or overflows) and un-scaled where it meets the fp32 optimiser. Chosen by the schedule wave from the largest
  // |d loss / d fc3 output| of the step before (what torch.cuda.amp.GradScaler does with inf checks, ace_schedule.py:70,107-113). bf16: 1.
  float grad_scale, inv_grad_scale;
  // bit patterns of the largest propagated gradient magnitudes of the running step, scaled units (absmax_publish): 64 words on 64
  // different 128-byte lines, workgroup b folds into word (b & 63) * 32. ONE word would serialise every wavefront's atomic of a launch
  // on one L2 line (11-13 ns each: measured +51 us on the input-gradient chain, +13 us on the loss kernel); readers take the maximum of
  // the 64 (absmax_all). Last member: the host's state read stops before it.
  uint32_t dz_absmax_slots[64 * 32];
};

// Per-row metadata of the gathered batch, written where the batch is gathered and read by the loss kernel: without it the loss kernel
// starts with a chain of three dependent loads (idx -> view_idx -> view_image) in front of its first computation, because vmcnt
// completes in order. {view, image, target u, target v (float bits)} of row r; dst == null: not written.
struct GatherMeta {
  int4* dst;
  const int32_t* view_idx;
  const int32_t* view_image;
  const float* target_px;
  const int* hold;   // null, or the trainer's sticky fault word: while it is set the gather stores nothing (the rows of the faulted step must
                     // survive until the host has finished that step, head_api.hip wgo_recover)
};

struct SchedConfig {
  int schedule, iterations, warmup_iterations, cooldown_iterations;
  int loss_type, circle_schedule, refine_calibration;
  float soft_clamp, soft_clamp_min;
  double lr_min, lr_max, warmup_lr, cooldown_trigger_percent;
  double beta1, beta2, eps, weight_decay, calib_lr;
  int pose_refinement, pose_wait;
  double pose_lr;
  int f16;   // fp16 operands (dynamic gradient scale) instead of bf16
};

struct RowGemmArgs {
  const uint16_t* In;    // [M][K] bf16
  const uint16_t* W;     // [N][K] bf16
  const float* bias;     // [N] or null
  const uint16_t* add;   // [M][N] or null: added before the activation
  // ReLU masks as bits (round 6; the input-gradient layers used to re-read the forward layer's whole 16-bit output tile through LDS only to
  // test `> 0`: 36.7 MB per step, 20 KiB of LDS-DMA per tile and layer). The forward layer and the input-gradient layer that needs its mask
  // run on the same 80 x 128 tiles with the same accumulator layout, so the mask is LANE-PRIVATE: every multiplier lane packs the 40 bits
  // of its own outputs -- fragment f = 2 j + i: bits 9 - f / 25 - f of word x (word y) = the two (next two) of its four outputs are > 0
  // after the 16-bit rounding -- and the backward lane loads the same 8 bytes: [(row tile * 4 + column tile) * 4 + wave][lane] uint2,
  // 2 KiB per tile instead of 20.
  uint2* mask_out;        // forward (bias + ReLU) launches: where the bits go, or null (inference)
  const uint2* mask_in;   // input-gradient launches: out_main is zeroed where the bit is clear; null = not a masked launch
  const uint16_t* res;   // [M][N] residual input for AUX_RESIDUAL
  uint16_t* out_main;
  uint16_t* out_aux;
  float* bias_partials;  // [row tiles * 2][512] column sums of out_main (masked launches) or null
  int M, N, K, relu, aux_mode;
  const TrainState* st;
  int dbg;  // ablation switches for tools/ablate_rowgemm.hip (0 in production): 1 = no epilogue, 2 = no MFMA, 4 = no loads
  uint32_t* absmax;  // fp16 gradient launches: TrainState::dz_absmax_slots (else null)
};

struct WgradArgs {
  const uint16_t* dZ[MAX_LAYERS];
  const uint16_t* In[MAX_LAYERS];
  int64_t w_off[MAX_LAYERS], b_off[MAX_LAYERS];
  float* slabs;
  int64_t slab_stride;
  int M, nslabs, n_layers;
  const TrainState* st;
  int dbg;                // ablation switches (tools/ablate_rowgemm.hip): 1 = no stores, 2 = no MFMA, 4 = no loads
  const uint16_t* zeros;  // >= 256 bytes of zeros (source of the DMA for rows past the end of a slab)
};

struct LossArgs {
  const uint16_t* act;  // [n][512] output of fc2
  const uint16_t* W3;   // [no][512] bf16
  const float* b3;      // [no] fp32
  int n, no, use_homogeneous;
  float mean[3], max_inv_scale, min_inv_scale, h_beta;
  // training only (idx == null -> inference)
  const int64_t* idx;
  const int4* meta;   // GatherMeta::dst of the batch `idx` (null: the chain idx -> view_idx -> view_image is followed here)
  const float* target_px;
  const float* target_crds;      // [patches][3] ground-truth scene coordinates (zeros = none) or null: use_depth mode
  const int32_t* view_idx;
  const float* view_aug_inv;
  const float* view_K;
  const float* view_Kinv;
  const int32_t* view_image;
  const float* image_pose_inv;   // [images][16] world->cam poses used by this step (refined ones under pose refinement)
  float* row_dT;                 // [n][12] d(loss)/d(pose rows 0..2) per batch row, or null
  int* row_image;                // [n] image of each batch row, or null
  int loss_type, refine_calibration;
  float hard_clamp, depth_min, depth_max, depth_target, inlier_px, inv_batch, focal_init;
  const TrainState* st;
  // outputs
  float* out_xyz;        // [n][3] or null; with planar_hw > 0: [frames][3][planar_hw] maps (row m = frame * planar_hw + pixel)
  int planar_hw, row_offset;   // row_offset: index of this launch's row 0 in the whole batch (chunked inference)
  uint16_t* dZ;          // [n][512] gradient wrt the fc2 pre-activation
  float* fc3_partials;   // [blocks][fc3_stride]
  int64_t fc3_stride;
  float* stat_partials;  // [blocks][4]
  float* bias_partials;  // [blocks][512] column sums of dZ
  int dbg;               // ablation (tools/ablate_rowgemm.hip): 1 = stop after phase A, 2 = stop after phase B
  uint32_t* absmax;      // fp16 training: TrainState::dz_absmax_slots (else null)
  const int* fault;      // training: the trainer's sticky fault word (null: none). While it is set the launch is a no-op: the buffers of the
                         // faulted step stay as they are until the host's fall-back has finished that step
};

struct GradReduceArgs {
  const float* slabs;
  int64_t slab_stride;
  int nslabs;
  const float* fc3_partials;
  int64_t fc3_stride;
  const float* stat_partials;
  int n_loss_blocks;
  const float* bias_partials;  // [n_layers][bias_layer_stride]
  int64_t bias_layer_stride;
  int bias_count[MAX_LAYERS];  // partial rows per layer
  int n_layers;
  float* grad;
  int64_t n_wide, n_params;
  const TrainState* st;
  int skip_wide;   // fused single-GPU step: the slabs are summed by adamw_kernel itself, only the tail is reduced here
  const int* fault;   // the trainer's rowseq fault word (null: none); written to statistics slot 3
};

struct AdamArgs {
  float* params;
  float* m;
  float* v;
  const float* grad;
  uint16_t* Wb;
  uint16_t* WbT;
  uint16_t* W3b;
  int64_t w_off[MAX_LAYERS], b_off[MAX_LAYERS];
  int64_t fc3_off, n_fc3, n_params;
  int n_layers, no;
  const TrainState* st;
  // fused single-GPU step (acez_train_step): wide-layer gradients are read as the fixed-order sum of the wgrad slabs
  // (the same additions grad_reduce_kernel performs) instead of from `grad`; null in the backward / all-reduce / update flow
  const float* slabs;
  int nslabs;
  int64_t slab_stride;
  GradReduceArgs tail;   // with slabs != null: the partials of the small parameters and statistics (reduced here as well)
  int* fault;            // the trainer's rowseq fault word: a faulted step updates nothing
  int f16;               // the compute copies W / W^T / W3 are fp16 (else bf16)
  int layer_lo, layer_hi;   // wide layers whose WEIGHT tiles this launch updates (sharded data-parallel update: the rank's own
                            // layers); biases, fc3 and the statistics are always handled. 0 .. n_layers = everything
};

// wgrad_opt_kernel (single-GPU fused step): the weight-gradient launch finishes the optimiser step of the wide layers itself. The two
// row slabs of a 128 x 128 gradient tile run on two CUs of ONE XCD; each sends the half of its partial tile the other one owns through
// that XCD's L2 (`xch`) and bumps the owner's counter (`flags`, the hand-off of rowseq_kernel: L2-local atomic, bounded sc1 poll, no
// fence), then applies AdamW to its own half: slab 0 + slab 1 in that order, the additions grad_reduce_kernel / adamw_kernel perform.
// What a loader wave of wgrad_opt_kernel whose exchange poll expired leaves behind, so that the host's fall-back can FINISH the step
// (head_api.hip wgo_recover: wgrad_kernel on the step's untouched buffers, then wgo_recover_kernel on exactly the regions whose wave
// timed out, with the step's own optimiser scalars): after it the parameters are bitwise those of the two-launch flow's step.
struct WgoFaultRec {
  AdamScalars s;          // st->adam of the faulted step (the schedule wave of that launch has already moved the state on)
  float inv_scale;        // st->inv_grad_scale of the faulted step
  int M;                  // its batch rows
  uint32_t epoch;         // WgradOptArgs::epoch of the faulted launch (0: no wgrad_opt fault is pending)
  uint32_t pad;
  const uint16_t* in0;    // WgradArgs::In[0] of the faulted launch (the trainer swaps its two input buffers every step)
};

struct WgradOptArgs {
  AdamArgs ad;           // the optimiser's arguments (slabs unused; tail = the partial buffers of this step: NaN / overflow guards)
  float* xch;            // [n_layers * 16 tiles][2 owners][64][128] fp32
  uint32_t* flags;       // [n_layers * 16 tiles][2 owners][32]: one counter per receiving workgroup on its own 128-byte line
  uint32_t target;       // value of a counter once both sending waves of this launch have stored (2 x launches so far)
  uint32_t spin_limit;   // poll budget, as RowSeqArgs::spin_limit
  int nsmall;            // workgroups 0 .. nsmall - 1 also run one small-parameter block of the optimiser each (adamw_small_columns); 0: none
  int do_post;           // the last workgroup also runs the schedule wave that closes the step (sched_post_wave)
  unsigned long long* trace;   // diagnostics build (ACEZ_WGO_TRACE=1, tools/wgo_trace.py): [256 workgroups][12 waves][8] s_memtime stamps; else null
  uint32_t epoch;        // launches so far, this one included (> 0)
  uint32_t* status;      // [workgroups][8 loader waves]: epoch * 4 + 3 = "this wave's poll expired in launch `epoch`, its rows of the half tile
                         // were NOT updated" (written by timed-out waves only: nothing is stored on the normal path)
  WgoFaultRec* rec;      // written by the timed-out waves of the FIRST faulting launch (later launches see the fault word and skip by guard)
  int fault_mod;         // diagnostics build: workgroups with b % fault_mod == 1 wait for a count that never comes (a partial fault); 0: none
};

}  // namespace acez
