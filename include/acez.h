/*
 * acez.h -- C ABI of libacez.so, the MI355X (gfx950) implementation of ACE Zero's hot path.
 *
 * Two groups of entry points, one per hot-path row group of SURVEY.md section 8:
 *
 *   (R) DSAC* RANSAC-PnP registration     replaces  dsacstar/dsacstar.cpp:66-186 (dsacstar_rgb_forward),
 *                                                   bound at dsacstar/dsacstar.cpp:898-899 as
 *                                                   dsacstar.forward_rgb and called from
 *                                                   register_mapping.py:229-242.
 *   (T) scene-coordinate head training    replaces  ace_trainer.py:454-497 (run_epoch gathers) and
 *                                                   ace_trainer.py:499-679 (training_step), i.e.
 *                                                   ace_network.py:120-149 (Head.forward),
 *                                                   ace_loss.py:39-91 (ReproLoss.compute),
 *                                                   ace_schedule.py:72-126 (ScheduleACE).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / pybind types cross this boundary.
 *   - every "d_" pointer is a DEVICE pointer (HBM), every "h_" pointer a host pointer.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream). All device work is enqueued on it;
 *     functions documented as asynchronous return before the work completes.
 *   - return value: ACEZ_OK (0) or a negative error code; nothing throws across the boundary.
 *     "PnP failed" is NOT an error (dsacstar_util.h:104-117,185-196): the call succeeds with a zero pose.
 *   - the library owns only its contexts and their workspaces; all in/out buffers are caller-owned.
 */
#ifndef ACEZ_H
#define ACEZ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACEZ_OK 0
#define ACEZ_ERR_INVALID (-1)   /* bad argument (null pointer, bad shape, unsupported option)          */
#define ACEZ_ERR_HIP (-2)       /* a HIP runtime call failed; acez_last_error() has the message         */
#define ACEZ_ERR_NODEVICE (-3)  /* no gfx950 device visible                                              */
#define ACEZ_ERR_NAN (-4)       /* NaN loss detected (ace_trainer.py:615-617 aborts here)                */

/* Message of the last error on this thread (never NULL). */
const char* acez_last_error(void);
/* Library version string, e.g. "acez 0.1 (gfx950)". */
const char* acez_version(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int acez_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * (R) DSAC* registration
 * ---------------------------------------------------------------------------------------------- */

/* Arguments of dsacstar.forward_rgb (dsacstar.cpp:66-78) that are shared by all frames of a batch. */
typedef struct acez_ransac_params {
  int32_t hypotheses;        /* ransacHypotheses    (register_mapping.py:64, 64; ace_zero.py:140, 32)   */
  int32_t max_tries;         /* max_hypotheses_tries (register_mapping.py:67 1e6; ace_zero.py:233, 16)  */
  float inlier_threshold;    /* inlierThreshold, px (register_mapping.py:70, 10)                        */
  float inlier_alpha;        /* inlierAlpha         (register_mapping.py:73, 100)                       */
  float max_reproj;          /* maxReproj, px       (register_mapping.py:76, 100)                       */
  int32_t subsampling;       /* subSampling = network.OUTPUT_SUBSAMPLE = 8 (ace_network.py:159)         */
  int32_t max_ref_steps;     /* MAX_REF_STEPS = 100 (dsacstar.cpp:47); <=0 selects 100                  */
  int32_t reserved;
} acez_ransac_params;

/* Per-frame pinhole intrinsics: focalLength, ppointX, ppointY of dsacstar.cpp:70-72. */
typedef struct acez_intrinsics {
  float focal, ppx, ppy;
} acez_intrinsics;

typedef struct acez_ransac acez_ransac; /* opaque context: stream-ordered workspaces for up to max_frames */

/* Create a registration context able to process batches of up to `max_frames` frames of at most
 * `max_h` x `max_w` scene coordinates (60x80 for 480x640 input). device < 0 selects the current device. */
int acez_ransac_create(acez_ransac** out, int max_frames, int max_h, int max_w, int device);
void acez_ransac_destroy(acez_ransac* ctx);

/* Batched device variant -- scene coordinates never leave HBM.
 *   d_scene_coords  float32 [n_frames][3][h][w], the layout Regressor.forward produces (ace_network.py:265-270)
 *   h_intrinsics    n_frames entries (host)
 *   seed            randomSeed of dsacstar.cpp:77
 *   h_frame_ids     n_frames entries (host) or NULL (= 0..n-1). The random stream is keyed by
 *                   (seed, frame_id, hypothesis, try, draw), see DESIGN.md "RNG"; this replaces the
 *                   call-order dependent ThreadRand (thread_rand.cpp:13-42).
 *   d_out_poses     float32 [n_frames][4][4] row-major cam->world pose (dsacstar.cpp:177-182)
 *   d_out_inliers   int32   [n_frames] return value of forward_rgb (dsacstar.cpp:185)
 *   d_out_masks     uint8   [n_frames][h][w] inlier map that produced the pose, or NULL
 * Asynchronous on `stream`. */
int acez_register_rgb_device(acez_ransac* ctx, const float* d_scene_coords, int n_frames, int h, int w,
                             const acez_ransac_params* params, const acez_intrinsics* h_intrinsics,
                             uint64_t seed, const uint64_t* h_frame_ids, float* d_out_poses,
                             int32_t* d_out_inliers, uint8_t* d_out_masks, void* stream);

/* Host-buffer variant with the exact argument meaning of dsacstar.forward_rgb for ONE frame
 * (what the reference's pybind wrapper would call). Strides are in elements, as an ATen accessor
 * reads them (dsacstar.cpp:83-84). Synchronous. Returns ACEZ_OK and writes *out_inliers. */
int acez_register_rgb_host(acez_ransac* ctx, const float* h_scene_coords, int64_t stride_c,
                           int64_t stride_h, int64_t stride_w, int h, int w,
                           const acez_ransac_params* params, const acez_intrinsics* intr, uint64_t seed,
                           uint64_t frame_id, float* h_out_pose16, int32_t* out_inliers,
                           uint8_t* h_out_mask /* nullable, h*w */);

/* Diagnostics for the parity tests: copies the per-hypothesis results of the LAST device call.
 *   h_hyp_poses  float64 [n_frames][hypotheses][6]  (rvec, tvec) after sampling
 *   h_scores     float64 [n_frames][hypotheses]     soft inlier scores
 *   h_best       int32   [n_frames]                 selected hypothesis
 *   h_refined    float64 [n_frames][6]              (rvec, tvec) after refinement
 * Any pointer may be NULL. Synchronous. */
int acez_ransac_debug_fetch(acez_ransac* ctx, int n_frames, int hypotheses, double* h_hyp_poses,
                            double* h_scores, int32_t* h_best, double* h_refined);

/* ------------------------------------------------------------------------------------------------
 * (T) head training / inference
 * ---------------------------------------------------------------------------------------------- */

#define ACEZ_HEAD_CHANNELS 512 /* ace_network.py:82, hard-coded in the reference too */

/* learning-rate schedules of ace_schedule.py:17-67 */
#define ACEZ_SCHED_CONSTANT 0
#define ACEZ_SCHED_1CYCLEPOLY 1
#define ACEZ_SCHED_CIRCLE 2
/* loss types of ace_loss.py:39-91 */
#define ACEZ_LOSS_TANH 0
#define ACEZ_LOSS_DYNTANH 1
#define ACEZ_LOSS_L1 2
#define ACEZ_LOSS_L1_SQRT 3
#define ACEZ_LOSS_L1_LOGL1 4

typedef struct acez_head_desc {
  int32_t num_head_blocks;   /* train_ace.py:83, default 1 -> 8 layers of 512x512 + fc3                */
  int32_t use_homogeneous;   /* train_ace.py:89, default 1 -> fc3 has 4 outputs                         */
  float mean[3];             /* Head.mean (ace_network.py:118)                                          */
  float max_inv_scale;       /* 1/homogeneous_max_scale = 0.25 (ace_network.py:112)                     */
  float min_inv_scale;       /* 1/homogeneous_min_scale = 100  (ace_network.py:114)                     */
  float h_beta;              /* ln2/(1-max_inv_scale)          (ace_network.py:113)                     */
} acez_head_desc;

typedef struct acez_train_config {
  acez_head_desc head;
  int32_t max_batch;           /* upper bound of rows per step on this rank (train_ace.py:137, 5120)     */
  int32_t global_batch;        /* loss normaliser B (ace_trainer.py:613); = max_batch on one GPU         */
  /* loss (ace_trainer.py:158-166, ace_loss.py) */
  int32_t loss_type;
  float soft_clamp;            /* repro_loss_soft_clamp 50                                               */
  float soft_clamp_min;        /* repro_loss_soft_clamp_min 1                                            */
  int32_t circle_schedule;     /* ace_loss.py:62                                                         */
  float hard_clamp;            /* repro_loss_hard_clamp 1000                                             */
  float depth_min, depth_max, depth_target; /* 0.1, 1000, 10 (train_ace.py:166-176)                      */
  float inlier_px_threshold;   /* learning_rate_cooldown_trigger_px_threshold 10                         */
  /* optimiser + schedule (ace_schedule.py) */
  int32_t schedule;
  int32_t iterations;          /* options.iterations                                                     */
  double lr_min, lr_max;
  int32_t warmup_iterations;   /* 1cyclepoly                                                             */
  double warmup_lr;
  int32_t cooldown_iterations;
  double cooldown_trigger_percent; /* 0.7                                                                */
  double beta1, beta2, eps, weight_decay; /* AdamW defaults 0.9 0.999 1e-8 1e-2                          */
  /* calibration refinement (refine_calibration.py): 0 = off */
  int32_t refine_calibration;
  float focal_init;            /* CalibrationRefiner.focal_length_init                                   */
  double calib_lr;
  /* pose refinement (refine_poses.py): 0 = none, 1 = naive (the 3x4 poses are the parameters, :224-234),
   * 2 = mlp (PoseNetwork(0,128), :152-176) */
  int32_t pose_refinement;
  int32_t pose_refinement_wait;   /* train_ace.py:220                                                     */
  double pose_refinement_lr;      /* train_ace.py:223, 1e-3                                               */
  float pose_refinement_weight;   /* train_ace.py:216, 0.1                                                */
  int32_t pose_refinement_ortho;  /* --refinement_ortho (train_ace.py:226): 0 = gram-schmidt, 1 = procrustes */
  /* 16-bit operand format of the head's GEMMs (features, activations, propagated gradients, weight copies; fp32 accumulation,
   * fp32 geometry / loss / optimiser either way). The reference runs the head under fp16 autocast (train_ace.py --use_half True,
   * ace_trainer.py:330,517-518): ACEZ_DTYPE_FP16 mirrors that (the gradient chain carries a power-of-two scale that the device-side
   * schedule adapts every step, the role torch.cuda.amp.GradScaler plays in ace_schedule.py:70,107-113); ACEZ_DTYPE_BF16 is BASELINE.json's north_star and the default. --use_half False (fp32) is
   * rejected, never silently replaced. d_features of acez_train_buffer / acez_head_forward are in this format. */
  int32_t compute_dtype;
  /* 1: a context for acez_head_forward / acez_head_forward_maps only (Regressor at registration time, register_mapping.py:201-242): no
   * gradient, weight-gradient, partial-sum or second input buffers are allocated (they are 60 % of a whole-frame context's memory) and
   * the training entry points return ACEZ_ERR_INVALID. 0: a trainer (which can also run inference). */
  int32_t inference_only;
} acez_train_config;
#define ACEZ_DTYPE_BF16 0
#define ACEZ_DTYPE_FP16 1
#define ACEZ_DTYPE_FP32 2

/* Caller-owned parameter storage, so the host side can expose the same state_dict keys as
 * ace_network.Head (ace_trainer.py:690-693) as views of one flat tensor.
 * Flat order = Head.named_parameters(): for each 512x512 layer {weight[out][in], bias[out]},
 * then fc3.weight[no][512], fc3.bias[no]  (2 103 300 floats for the default head). */
typedef struct acez_param_buffers {
  float* d_params;  /* fp32 master weights                                   */
  float* d_adam_m;  /* AdamW exp_avg                                         */
  float* d_adam_v;  /* AdamW exp_avg_sq                                      */
  float* d_grad;    /* flat fp32 gradient + 4 trailing stats floats {loss_sum, inlier_count, focal_grad, nan_flag};
                       this is the bucket a data-parallel host all-reduces between
                       acez_train_backward and acez_train_update */
  int64_t n_params; /* must equal acez_head_num_params(head)                 */
  /* pose-refinement network, flat in PoseNetwork.named_parameters() order (head_skip, conv1..3, fc1..3; 70 924
   * floats); only read when cfg.pose_refinement == 2. Its gradient is appended to d_grad after the 4 statistics,
   * so d_grad holds n_params + 4 + n_pose_params floats and one all-reduce still covers everything. */
  float* d_pose_params;
  float* d_pose_m;
  float* d_pose_v;
  int64_t n_pose_params; /* mlp: ACEZ_POSE_MLP_PARAMS; naive: 12 * n_images (d_pose_params = the [n_images][3][4] poses); none: 0 */
} acez_param_buffers;
#define ACEZ_POSE_MLP_PARAMS 70924

/* The training buffer of ace_trainer.py:330-340, with the per-image data stored once per view
 * (image x augmentation pass) instead of once per patch. */
typedef struct acez_train_buffer {
  const void* d_features;      /* bf16 (fp16 with ACEZ_DTYPE_FP16) [n_patches][512]                    */
  const float* d_target_px;    /* f32  [n_patches][2]                                                  */
  const int32_t* d_view_idx;   /* i32  [n_patches]  -> view                                            */
  int64_t n_patches;
  const float* d_view_aug_inv; /* f32 [n_views][3][4]   aug_poses_inv (ace_trainer.py:331)             */
  const float* d_view_K;       /* f32 [n_views][3][3]   intrinsics                                     */
  const float* d_view_Kinv;    /* f32 [n_views][3][3]   intrinsics_inv                                 */
  const int32_t* d_view_image; /* i32 [n_views]  -> image (pose_idx of ace_trainer.py:339)             */
  int32_t n_views;
  const float* d_image_pose_inv; /* f32 [n_images][4][4] world->cam poses_inv (ace_trainer.py:332)      */
  int32_t n_images;
  const float* d_target_crds;  /* f32 [n_patches][3] ground-truth scene coordinates from depth, zeros where none
                                  (ace_trainer.py:338); non-NULL selects the use_depth variant of the step
                                  (ace_trainer.py:567-574,601-609), NULL the constant-depth proxy target */
} acez_train_buffer;

typedef struct acez_train_state {
  int32_t iteration;        /* TrainerACE.iteration                                                    */
  int32_t max_iterations;   /* ScheduleACE.max_iterations (rewritten by the cool-down trigger)         */
  int32_t in_cooldown;
  int32_t nan_flag;
  double lr;                /* lr the NEXT optimiser step will use                                     */
  float last_loss;          /* loss of the last completed step (already / global_batch)                */
  float last_batch_inliers; /* fraction, ace_trainer.py:586                                            */
  double focal_scale;       /* 1 + global_f (refine_calibration.py:28-32)                              */
  float grad_scale;         /* fp16: the power of two on the propagated gradients of the NEXT step (GradScaler's scale); bf16: 1 */
  int32_t opt_steps;        /* AdamW steps applied so far (< iteration when fp16 overflows skipped updates, as GradScaler.step does) */
} acez_train_state;

typedef struct acez_trainer acez_trainer;

int64_t acez_head_num_params(const acez_head_desc* head);
int acez_trainer_create(acez_trainer** out, const acez_train_config* cfg, const acez_param_buffers* params,
                        int device);
void acez_trainer_destroy(acez_trainer* tr);
int acez_trainer_set_buffer(acez_trainer* tr, const acez_train_buffer* buf);
/* Re-derive the bf16 compute copies (W and W^T) from the fp32 master weights; call after loading weights. */
int acez_trainer_sync_weights(acez_trainer* tr, void* stream);

/* One iteration of ace_trainer.py:499-679 on `n` buffer rows selected by d_indices (int64, device),
 * split in two so that a data-parallel host can all-reduce d_grad in between:
 *   backward: schedule bookkeeping, gather, head forward, loss, head backward -> d_grad
 *   update  : AdamW on the fp32 masters, bf16 recast, scheduler step
 * Both are asynchronous and never synchronise the host. A step issued after the schedule has ended
 * (iteration >= max_iterations, ace_trainer.py:509-510) is a device-side no-op. */
int acez_train_backward(acez_trainer* tr, const int64_t* d_indices, int n, void* stream);
int acez_train_update(acez_trainer* tr, void* stream);
/* acez_train_update with the NEXT acez_train_backward's indices announced (a data-parallel rank knows its rows of the next batch: every
 * rank draws the same epoch permutation, ace_trainer.py:466-494): the next batch is gathered, and this step's schedule bookkeeping
 * closed, inside the optimiser's launch, so the next acez_train_backward called with the same device pointer and count starts with the
 * forward chain instead of a gather launch. Same contract for the announced indices as acez_train_step_next; any other next call is
 * still correct. Bitwise the same parameters and state as acez_train_update. With the pose network (--pose_refinement mlp) the rows are
 * gathered in the optimiser's launch as well and the next backward's first launch keeps the pose forward and the schedule wave. With
 * d_indices_next == NULL or n_next == 0 (a rank whose shard holds no row of the next batch): exactly acez_train_update. */
int acez_train_update_next(acez_trainer* tr, const int64_t* d_indices_next, int n_next, void* stream);
/* Sharded data-parallel update (reduce-scatter of the weight gradients, each rank updates its own layers, all-gather
 * of the 16-bit compute copies; no reference counterpart -- the reference is single-GPU):
 *   acez_train_update_layers        AdamW on the weight matrices of wide layers [layer_lo, layer_hi) only (d_grad holds their reduced
 *                                   gradients) and on ALL small parameters (biases, fc3: their gradients are all-reduced, the update is
 *                                   replicated); closes the step's schedule bookkeeping like acez_train_update
 *   acez_trainer_export_weights16   16-bit W[out][in] of layers [layer_lo, layer_hi) -> d_dst  ((hi - lo) x 512 x 512 elements)
 *   acez_trainer_import_weights16   d_src -> the 16-bit W of those layers, and rebuilds their transposed copies
 *   acez_trainer_import_weights16_all  the receiving side of the all-gather in ONE launch: d_src_all = [L][512][512] 16-bit (every rank's
 *                                   layers); all layers OUTSIDE [own_lo, own_hi) are taken over (W and W^T), the rank's own stay as its
 *                                   optimiser wrote them
 * The fp32 masters / moments of layers a rank does not own go stale on it; the host copies the owners' values in before it reads them. */
int acez_train_update_layers(acez_trainer* tr, int layer_lo, int layer_hi, void* stream);
int acez_trainer_export_weights16(acez_trainer* tr, int layer_lo, int layer_hi, void* d_dst, void* stream);
int acez_trainer_import_weights16(acez_trainer* tr, int layer_lo, int layer_hi, const void* d_src, void* stream);
int acez_trainer_import_weights16_all(acez_trainer* tr, int own_lo, int own_hi, const void* d_src_all, void* stream);

/* Single-GPU step = backward + update, with the wide-layer weight gradients handed straight to the optimiser: bitwise the same
 * parameters as the two calls above; afterwards d_grad holds the bias / fc3 gradients and the statistics only (its wide-layer weight
 * part is not written). Without pose refinement, and while the one-launch chains are enabled (acez_trainer_seq_status), the weight-gradient
 * launch applies the whole optimiser step itself (wgrad_opt_kernel: the split-K halves of a tile are exchanged inside one XCD's L2 and
 * never reach memory) and closes the step's schedule; otherwise the slabs go through memory to a separate optimiser launch. Use
 * backward / all-reduce / update when ranks exchange d_grad. */
int acez_train_step(acez_trainer* tr, const int64_t* d_indices, int n, void* stream);
/* acez_train_step with the NEXT step's indices announced (run_epoch walks consecutive slices of one permutation, ace_trainer.py:466-494,
 * so the caller knows them): the next batch is gathered inside this step's launches (beside the loss kernel, into the second of the
 * trainer's two input buffers; with pose refinement: inside the optimiser launch, with the schedule bookkeeping). The next call must pass the same device pointer and count to profit; a different pointer or count is still correct (the batch
 * is then gathered again). The rows gathered ahead are recognised by (pointer, count) only: the n_next indices at d_indices_next must
 * not be rewritten, nor their memory freed and reused, between this call and the next step call -- acez_trainer_set_buffer and
 * acez_trainer_sync_weights drop the rows gathered ahead. NULL = acez_train_step. */
int acez_train_step_next(acez_trainer* tr, const int64_t* d_indices, int n, const int64_t* d_indices_next, int n_next, void* stream);
/* Synchronises `stream` and copies the schedule state. */
int acez_trainer_get_state(acez_trainer* tr, acez_train_state* h_out, void* stream);
/* Per-iteration log kept on the device: loss and batch_inliers of iterations [first, first+count). */
int acez_trainer_get_log(acez_trainer* tr, int first, int count, float* h_loss, float* h_inliers, void* stream);
/* Scene coordinates predicted in the last backward call: f32 [n][3] (diagnostics / tests). */
int acez_trainer_last_scene_coords(acez_trainer* tr, float* h_xyz, int n, void* stream);

/* Diagnostics for the roofline measurement (bench.py): bracket every kernel launch of the following steps with HIP
 * events on the launch stream; get_profile synchronises and returns summed milliseconds and launch counts for the
 * 8 kernel classes {sched, gather, gemm_fwd, loss, gemm_dgrad, wgrad, grad_reduce, adamw}, then clears the record. */
int acez_trainer_set_profiling(acez_trainer* tr, int enable);
int acez_trainer_get_profile(acez_trainer* tr, float* h_ms8, int32_t* h_counts8);

/* Status of the one-launch GEMM chains (rowseq_kernel): *enabled = 1 while the trainer uses them, *probe = result
 * of the placement probe run by acez_trainer_create (1 passed, 0 failed -> per-layer launches from the start, -1 not run), *faults =
 * number of times a bounded hand-off poll expired and the trainer fell back to per-layer launches (acez_trainer_get_state performs the
 * fall-back; the abandoned iterations are device-side no-ops and are not counted in acez_train_state.iteration). No reference
 * counterpart. Any pointer may be NULL.
 * A step is atomic under either kind of fault (ace_trainer.py:620-640): a poll that expires inside one of the GEMM chains abandons the
 * step before anything of it is applied (state = the step before); a poll that expires in the fused weight-gradient / optimiser launch
 * (wgrad_opt_kernel) is FINISHED by the same fall-back -- the launches issued after it were no-ops, the step's operands are intact, the
 * rows that were not updated are updated with that step's optimiser scalars -- so that the state is exactly the step after, bit for bit
 * what acez_train_backward + acez_train_update would have produced. */
int acez_trainer_seq_status(acez_trainer* tr, int* enabled, int* probe, int* faults);

/* Diagnostics for the tests: copy one intermediate device buffer of the last backward call to the host (synchronous).
 * kind 0: post-ReLU output of wide layer `index`, bf16 [n][512];  1: dZ of layer `index`, bf16 [n][512];
 * 2: residual stream `index` (0 = the gathered batch), bf16 [n][512];  3: weight-gradient slab `index`, f32 [n_wide];
 * 4: bias-gradient partial rows of layer `index`, f32 [max_batch/32][512];  5: s_memtime stamps of the chain kernel
 * (ACEZ_CHAIN_TRACE=1), u64 [2][256];  6: XCD placement record of the one-launch GEMM chains (ACEZ_SEQ_XCC=1), u32 [8 + 256]:
 * words 0..7 = OR of 1 << XCC_ID over the workgroups with blockIdx & 7 = word, word 8 + 4 mt + nt = XCC_ID that ran tile
 * (mt, nt) in the last launch.  No reference counterpart (autograd internals). */
int acez_trainer_debug_read(acez_trainer* tr, int kind, int index, void* h_out, int64_t bytes, void* stream);

/* Current refined world->cam poses of all images, f32 [n_images][3][4] (PoseRefiner.get_all_current_poses,
 * refine_poses.py:184-210); the original poses when pose refinement is off. Synchronous. */
int acez_trainer_get_poses(acez_trainer* tr, float* h_poses34, void* stream);

/* Head inference (Regressor.get_scene_coordinates, ace_network.py:262-263) on n feature rows:
 *   d_features bf16 [n][512]  ->  d_out_xyz f32 [n][3].
 * Asynchronous for passes of more than 5120 rows per max_batch chunk. A pass small enough for the one-launch chains (<= 5120 rows) ends
 * with a read of the trainer's fault word, i.e. it SYNCHRONISES `stream` before it returns: an expired hand-off poll would have left
 * garbage in d_out_xyz, and the call repeats the pass on per-layer launches in that case (it also performs a pending training
 * fall-back, see acez_trainer_seq_status). */
int acez_head_forward(acez_trainer* tr, const void* d_features, int n, float* d_out_xyz, void* stream);

/* Head.forward for whole frames, written as Regressor.forward's [B,3,H,W] maps (ace_network.py:265-270): the input of
 * acez_register_rgb_device, so that scene coordinates go encoder -> head -> RANSAC without leaving HBM
 * (register_mapping.py:209-213 copies them to the CPU instead).
 *   d_features  16-bit [n_frames * h * w][512] in the trainer's compute_dtype (bfloat16 or float16), rows in (frame, y, x) order
 *               (acez_encoder_forward's output for an encoder of the same compute_dtype)
 *   d_out_maps  float32  [n_frames][3][h][w] */
int acez_head_forward_maps(acez_trainer* tr, const void* d_features, int n_frames, int h, int w, float* d_out_maps,
                           void* stream);

/* =====================================================================================================
 * E. Feature encoder: ace_network.py:14-59 Encoder.forward
 *    11 convolutions, two residual blocks, output stride 8 (Regressor.OUTPUT_SUBSAMPLE, ace_network.py:159).
 * ===================================================================================================== */
#define ACEZ_ENCODER_LAYERS 11
/* Layer order of the weight/bias pointer arrays == Encoder.__init__ (ace_network.py:26-40):
 *   conv1 conv2 conv3 conv4 res1_conv1 res1_conv2 res1_conv3 res2_conv1 res2_conv2 res2_conv3 res2_skip */

typedef struct acez_encoder acez_encoder; /* opaque: 16-bit weight matrices + activation workspaces for max_frames frames */

/* h_weights[i]  float32 host pointer, torch Conv2d layout [c_out][c_in][k][k] (state_dict "<name>.weight")
 * h_biases[i]   float32 host pointer [c_out]                                   (state_dict "<name>.bias")
 * out_channels  Encoder(out_channels): c_out of res2_conv3 and res2_skip (512 in every shipped encoder)
 * max_frames / max_h / max_w   capacity of one internal pass; acez_encoder_forward chunks larger batches.
 * compute_dtype ACEZ_DTYPE_BF16 or ACEZ_DTYPE_FP16: the 16-bit format of the weights, of every activation and of the feature rows.
 *               The reference runs this network under fp16 autocast (ace_trainer.py:366-367 buffer creation, register_mapping.py:209-210
 *               registration; ace_network.py:41-59): ACEZ_DTYPE_FP16 is that arithmetic (fp16 operands, fp32 accumulation, one rounding
 *               per layer output). Must equal the compute_dtype of the head that consumes the rows. */
int acez_encoder_create(acez_encoder** out, const float* const* h_weights, const float* const* h_biases,
                        int out_channels, int max_frames, int max_h, int max_w, int compute_dtype, int device);
void acez_encoder_destroy(acez_encoder* enc);

/* Spatial size of the feature map for an h x w input: three stride-2, pad-1, 3x3 convolutions. */
int acez_encoder_output_size(int h, int w, int* out_h, int* out_w);

/* Encoder.forward (ace_network.py:42-59) for n_frames grayscale frames.
 *   d_images    float32 [n_frames][1][h][w], normalised as dataset.py:150-153 does
 *   d_features  16-bit (the context's compute_dtype) [n_frames * out_h * out_w][out_channels]: one row per feature-map pixel in
 *               (frame, y, x) order -- the row layout of acez_train_buffer.d_features and of acez_head_forward's input
 * 16-bit operands, fp32 accumulation (the reference runs this network under fp16 autocast, register_mapping.py:209).
 * Asynchronous on `stream`. */
int acez_encoder_forward(acez_encoder* enc, const float* d_images, int n_frames, int h, int w, void* d_features,
                         void* stream);

/* Training-buffer sampling for a batch of views whose features were just computed (ace_trainer.py:404-431): per view,
 * `samples_per_view` rows are drawn uniformly with replacement among the pixels whose mask byte is non-zero
 * (torch.multinomial(mask, n, replacement=True), ace_trainer.py:419-422) and written, view after view, to the output
 * arrays -- which are slices of acez_train_buffer's d_features / d_target_px / d_view_idx at the current fill offset.
 *   d_view_features  16-bit [n_views * map_h * map_w][channels]   (acez_encoder_forward's output; rows are copied, not interpreted)
 *   d_masks          uint8 [n_views][map_h][map_w] validity at feature resolution (the nearest-neighbour resize of
 *                    ace_trainer.py:373-374), or NULL = every pixel valid. Every view must have a valid pixel
 *                    (the reference skips empty views, ace_trainer.py:377-378; so must the caller).
 *   seed, first_view_id   the draw of sample s of view v is a counter-based stream keyed by (seed, first_view_id + v, s)
 *   view_index_base  value written to d_out_view_idx for view 0 (index into the caller's per-view tables)
 *   d_out_features   16-bit [n_views * samples_per_view][channels]
 *   d_out_target_px  float32  [n_views * samples_per_view][2] = 8 * (x + 0.5, y + 0.5)   (ace_util.py:7-13)
 *   d_out_view_idx   int32    [n_views * samples_per_view]
 *   d_out_pixel      int32    [n_views * samples_per_view] chosen feature-map pixel y * map_w + x, or NULL (diagnostics)
 * Asynchronous on `stream`. */
int acez_buffer_sample_views(const void* d_view_features, const uint8_t* d_masks, int n_views, int map_h, int map_w,
                             int channels, int samples_per_view, uint64_t seed, uint64_t first_view_id,
                             int32_t view_index_base, void* d_out_features, float* d_out_target_px,
                             int32_t* d_out_view_idx, int32_t* d_out_pixel, void* stream);

/* Augmented training views of resident frames (dataset.py:283-343: resize by a common factor, rotate about the centre, ColorJitter
 * brightness / contrast on the grey values, a validity mask that goes through the same warp with zero padding), the step in front of the
 * encoder when ace_trainer.py:293-452 fills the buffer with --use_aug True. One launch per batch of views of one canvas size (+ a
 * per-view reduction when jitter is on) instead of a dozen framework kernels and a 157 MB sampling grid per 64 views.
 *   d_images      float32 [n_images][H][W] normalised grey frames, resident on the device
 *   d_image_index int32 [n_views]: the frame each view is taken from
 *   d_theta       float32 [n_views][6]: the affine map of torch.nn.functional.affine_grid(align_corners=False) -- normalised output
 *                 coordinates (x_n, y_n, 1) -> normalised source coordinates -- row-major 2 x 3
 *   d_jitter      float32 [n_views][2] = (brightness, contrast) factors applied as torchvision's ColorJitter does on (v * 0.25 + 0.4)
 *                 clamped to [0, 1] (dataset.py:148), or NULL
 *   d_out_views   float32 [n_views][hs][ws]: bilinear, reflection padding (F.grid_sample(..., padding_mode="reflection", align_corners=False))
 *   d_out_mask    uint8 [n_views][map_h][map_w] or NULL: the zero-padded bilinear warp of an all-ones image, > 0, taken at the pixels the
 *                 nearest-neighbour resize to feature resolution reads (ace_trainer.py:373-374): what acez_buffer_sample_views expects
 *   d_scratch     float32 [n_views] (the per-view mean of the brightness-adjusted frame; used only with d_jitter)
 * Asynchronous on `stream`. */
int acez_buffer_warp_views(const float* d_images, int n_images, int H, int W, const int32_t* d_image_index, const float* d_theta,
                           const float* d_jitter, int n_views, int hs, int ws, float* d_out_views, uint8_t* d_out_mask, int map_h,
                           int map_w, float* d_scratch, void* stream);

/* =====================================================================================================
 * F. Point-cloud extraction
 * =====================================================================================================
 * Replaces the per-frame body of ace_vis_util.get_point_cloud_from_network (ace_vis_util.py:430-591; callers
 * export_point_cloud.py:87-94, ace_zero.py:379-400) from the point where the scene-coordinate maps of a batch of frames
 * exist on the device (acez_head_forward_maps' output): reprojection error against the pixel grid (:481-499),
 * scene-coordinate gradient with reflect padding (:501-510), the escalating gradient thresholds 0.1 / 0.5 / 1 / inf
 * (:443,512-516), depth filter and "keep all if nothing survives" (:518-526), reprojection threshold of 1 px (:451,528-530)
 * with the relaxed k-th-error branch (:535-544) and the random sub-sampling branch (:545-551), and the merge into one
 * point list in frame order with the OpenCV -> OpenGL flip (:574-587).  `torch.randperm`'s stream is replaced by a
 * counter-based draw keyed by (seed, first_frame_id + frame, rank of the surviving point); everything else is bit-exact
 * against the restatement in oracle/cloud_oracle.py, which is pinned on the reference function's own output.
 *   d_scene_coords        float32 [n_frames][3][map_h][map_w]
 *   d_poses_inv           float32 [n_frames][12]: rows of the 3x4 world -> camera transform (gt_inv_pose[:, :3], :479)
 *   d_intrinsics          float32 [n_frames][9]: row-major K
 *   filter_depth          metres (export_point_cloud.py:94 passes 100); dense_cloud as ace_vis_util.py:453-456
 *   points_per_image_min  int(100000 / len(data_loader)), points_per_image_max  int(1000000 / len(data_loader)) (:458-459)
 *   opengl_convention     non-zero: y and z negated as the reference returns them
 *   d_keep                uint8 [n_frames][map_h * map_w]   out: 1 = the pixel's point is part of the cloud
 *   d_counts              int32 [n_frames]                  out: points kept per frame
 *   d_offsets             int32 [n_frames + 1]              out: exclusive prefix of d_counts; [n_frames] = total N
 *   d_out_xyz             float32 [n_frames * map_h * map_w][3], first N rows written, frame after frame, pixel order
 *   d_out_source          int32 [same], frame * map_h * map_w + pixel of every point (colour lookup), or NULL
 * Asynchronous on `stream`. */
int acez_point_cloud_filter(const float* d_scene_coords, const float* d_poses_inv, const float* d_intrinsics, int n_frames,
                            int map_h, int map_w, float filter_depth, int dense_cloud, int points_per_image_min,
                            int points_per_image_max, uint64_t seed, uint64_t first_frame_id, int opengl_convention,
                            uint8_t* d_keep, int32_t* d_counts, int32_t* d_offsets, float* d_out_xyz, int32_t* d_out_source,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACEZ_H */
