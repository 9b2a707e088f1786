// head_api.hip -- C-ABI entry points of the head training / inference path (include/acez.h, group T).
// Host-side orchestration only: every entry point enqueues kernels of head_kernels.hip on the caller's stream.
#include "head_kernels.hip"
#include "pose_fused.hip"
#include "head_maps.hip"
#include "acez_common.h"
#include "conv_launch.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "acez_common.h"
#include <vector>
#include <new>

using namespace acez;

struct acez_trainer {
  char* arena_base = nullptr;   // current arena of dmalloc (all arenas are in `allocs`)
  size_t arena_size = 0, arena_cursor = 0;
  acez_train_config cfg;
  acez_param_buffers pb;
  acez_train_buffer buf;
  bool have_buf = false;
  int device = 0;
  int L = 0, nb = 0, no = 4;
  int64_t n_wide = 0, n_params = 0, fc3_off = 0;
  int64_t fc3_stride = 0;
  int nslabs = 1;
  int max_batch = 0;
  int last_n = 0;
  // device allocations
  uint16_t *Wb = nullptr, *WbT = nullptr, *W3b = nullptr;
  uint16_t* Wf = nullptr;       // fragment-ordered copy of Wb for head_maps_kernel (allocated at the first whole-frame pass)
  std::vector<uint16_t*> out;   // post-relu output of each wide layer
  std::vector<uint16_t*> R;     // residual stream, R[0] = gathered features
  std::vector<uint16_t*> dZ;    // gradient wrt each wide layer's pre-activation
  uint16_t* dR[2] = {nullptr, nullptr};
  uint2* maskbits = nullptr;    // [L][row tiles * 4 column tiles * 4 waves * 64 lanes] lane-private ReLU mask bits (RowGemmArgs::mask_out / mask_in)
  size_t mask_stride = 0;       // uint2 elements per layer
  float* slabs = nullptr;
  int4* batch_meta = nullptr;   // [max_batch] GatherMeta of the batch in R[0]
  float* fc3_partials = nullptr;
  float* stat_partials = nullptr;
  float* bias_partials = nullptr;
  int64_t bias_layer_stride = 0;
  float* xyz = nullptr;
  uint16_t* zeros = nullptr;
  float *log_loss = nullptr, *log_inl = nullptr;
  int log_cap = 0;
  TrainState* st = nullptr;         // = st_slot[st_cur]: the slot every launch made from now on reads
  TrainState* st_slot[2] = {nullptr, nullptr};   // sched_post_wave reads one slot and writes the other; the host flips after launching it
  int st_cur = 0;
  TrainState* st_infer = nullptr;   // schedule-free launches (inference) get this always-active state, so that a rowseq fault can switch them off too
  // pose refinement runs on its own stream, beside the head's GEMM chains (6 launches, ~65 us if serialised at 1000 images)
  hipStream_t pose_stream = nullptr;
  hipEvent_t ev_begin = nullptr, ev_pose_fwd = nullptr, ev_loss = nullptr, ev_pose_bwd = nullptr;
  GradReduceArgs last_reduce{};   // partial buffers of the last backward (input of the fused update)
  bool post_pending = false;   // acez_train_update has run; its schedule bookkeeping rides with the next step's gather (flush_post)
  // acez_train_step_next: the batch of the NEXT step was gathered (and this step's bookkeeping done) inside the optimiser's launch
  const int64_t* pre_idx = nullptr;
  int pre_n = 0;
  SchedConfig sc;
  std::vector<void*> allocs;
  // pose refinement (mlp): per-image activations and gradients, allocated by set_buffer (needs n_images)
  int pose_images = 0;
  float *pa1 = nullptr, *pa2 = nullptr, *pa3 = nullptr, *pr = nullptr, *pf1 = nullptr, *pf2 = nullptr, *pdlt = nullptr, *pose_cur = nullptr;
  float *pdT = nullptr, *pddelta = nullptr, *pdz2 = nullptr, *pdz1 = nullptr, *pdr = nullptr, *pdzc3 = nullptr, *pdzc2 = nullptr, *pdzc1 = nullptr;
  float* pose_wt = nullptr;     // [4][128][128] transposed pose-network weights (forward)
  // mlp refinement folded into the step's own launches (pose_fused.hip); ACEZ_POSE_FUSED=0 = the separate launches of round 2
  bool pose_fused = true;
  // images per pose workgroup of the fused path: 16 (pose_kernels.hip), 8 or 4 (pose_small.hip); ACEZ_POSE_TILE sets both, ACEZ_POSE_TILE_FWD
  // the forward alone (the global layouts do not depend on the tile size)
  int pose_tile = 4, pose_tile_fwd = 4;
  bool pose_wt_valid = false;   // pose_wt matches the parameters (kept up to date by the fused optimiser epilogue of pose_mlp_wgrad_kernel)
  float* row_dT = nullptr;
  int* row_image = nullptr;
  // optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline leg)
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  struct EvUse { int cls, i0, i1, launches; };
  std::vector<EvUse> ev_used;
  size_t ev_next = 0;
  int prof_launches = 0;  // launches inside the currently open scope
  // rowseq_kernel: the forward layers / the input-gradient layers as ONE launch each, kernel boundaries replaced by a same-XCD
  // hand-off (head_kernels.hip). Default when every workgroup can be resident (grid <= CUs); ACEZ_SEQ=0 = per-layer launches.
  bool seq = true;
  // 16-bit operand format of the GEMM chains (acez_train_config.compute_dtype): bf16, or fp16 with the gradient chain scaled by
  // grad_scale (fp16's smallest normal is 6e-5; the reference uses a GradScaler for the same reason, ace_schedule.py:70,107-113)
  bool f16 = false;
  int loss_rows = 4;   // rows per wavefront of loss_kernel (4 waves per workgroup): 4 = 16-row workgroups; ACEZ_LOSS_ROWS=8 (diagnostics build): 8
  int last_nblk = 0;   // loss workgroups of the last backward (the schedule wave scans their |ds| maxima in fp16 mode)
  int n_cus = 0;
  uint32_t* seq_flags = nullptr;  // [64 row tiles][32] hand-off counters, monotonically increasing
  uint32_t seq_base[64] = {};     // seams completed so far, per row tile
  uint32_t* seq_xcc = nullptr;    // ACEZ_SEQ_XCC=1: placement record of rowseq_kernel (acez_trainer_debug_read kind 6)
  // Safety net of the hand-off (the workgroup -> XCD mapping is not an API contract): a placement probe at creation decides whether
  // the one-launch chains are used at all; every poll is bounded and raises `seq_err` (device), which turns the optimiser /
  // schedule kernels of that step into no-ops; the next state read (seq_fault_check) resets the counters and switches this
  // trainer to per-layer launches for good. The abandoned iterations are not counted, so the training loop simply runs them again.
  int* seq_err = nullptr;              // = (int*)(seq_flags + 64 * 32): one allocation, so that the kernel needs no second pointer
  uint32_t seq_spin_limit = 40000;     // polls (>= ~0.5 us each: at least ~20 ms); ACEZ_SEQ_SPIN_US overrides (2 polls per us)
  int seq_faults = 0;                  // fall-backs taken so far
  int seq_probe = -1;                  // -1 not run, 0 failed (seq disabled), 1 passed
  long seq_launches = 0, seq_fault_at = -1;   // tests: ACEZ_SEQ_FAULT_AT=<n> makes the n-th launch time out
  // wgrad_opt_kernel (head_kernels.hip): the optimiser step of the wide layers inside the weight-gradient launch of the single-GPU fused
  // step. Same placement contract and safety net as the one-launch chains (tr->seq, the probe, the bounded poll, seq_err): usable only
  // while they are, and only when every workgroup of the launch is resident (wgrad_opt_usable).
  bool wgrad_opt = true;          // ACEZ_WGRAD_OPT=0 (diagnostics build): wgrad_kernel + the optimiser's tile workgroups, as before
  float* wg_xch = nullptr;        // [L * 16 tiles][2][64][128] fp32 exchange tiles
  uint32_t* wg_flags = nullptr;   // [L * 16 tiles][2][32] hand-off counters
  uint32_t wg_epoch = 0;          // wgrad_opt launches so far (a counter reaches 2 * epoch in each of them)
  bool wide_done = false;         // this step's backward has already applied the optimiser to the wide layers' weights AND the small parameters
  bool post_done = false;         // ... and run the schedule wave that closes the step (no pose refinement)
  // acez_train_step_next on this path: the next batch is gathered beside the loss kernel (loss_gather_kernel) into the OTHER input
  // buffer / metadata table (the weight-gradient launch of the running step still reads the current ones); swapped when the step ends
  uint16_t* R0_alt = nullptr;
  int4* batch_meta_alt = nullptr;
  bool next_gathered = false;
  const int64_t* next_idx = nullptr;
  int next_n = 0;
  long wgo_fault_at = -1;         // tests: ACEZ_WGO_FAULT_AT=<n> makes the n-th wgrad_opt launch time out
  int wgo_fault_mod = 0;          // tests: ACEZ_WGO_FAULT_MOD=<m>: ... only in the workgroups with b % m == 1 (a partially applied step)
  uint32_t* wg_status = nullptr;  // [workgroups][8 loader waves] WgradOptArgs::status
  WgoFaultRec* wg_rec = nullptr;  // WgradOptArgs::rec
  int wgo_recovered = 0;          // faulted wgrad_opt steps the fall-back has finished (wgo_recover)
  bool inference_only = false;    // acez_train_config.inference_only: no training buffers exist, the training entry points refuse
  bool sizing = false;            // acez_trainer_create's first pass: dmalloc only adds up
  size_t sized_total = 0;
  unsigned long long* pose_trace = nullptr;  // ACEZ_POSE_TRACE=1 (diagnostics build): [3][1024][16] stamps of the pose forward (S3) / S1 / S2 workgroups (debug_read kind 8)
  unsigned long long* wgo_trace = nullptr;   // ACEZ_WGO_TRACE=1 (diagnostics build): s_memtime stamps of wgrad_opt_kernel's last launch (debug_read kind 7)
};

enum { KC_SCHED = 0, KC_GATHER, KC_GEMM_FWD, KC_LOSS, KC_GEMM_DGRAD, KC_WGRAD, KC_REDUCE, KC_ADAMW, KC_COUNT };

struct ProfScope {
  acez_trainer* tr; hipStream_t s; int cls; int i0 = -1;
  ProfScope(acez_trainer* t, hipStream_t st, int c) : tr(t), s(st), cls(c) {
    if (!tr->profiling) return;
    if (tr->ev_next + 2 > tr->ev_pool.size()) {
      for (int i = 0; i < 64; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; tr->ev_pool.push_back(e); }
    }
    i0 = (int)tr->ev_next; tr->ev_next += 2;
    tr->prof_launches = 0;
    (void)hipEventRecord(tr->ev_pool[i0], s);
  }
  ~ProfScope() {
    if (i0 < 0) return;
    (void)hipEventRecord(tr->ev_pool[i0 + 1], s);
    tr->ev_used.push_back({cls, i0, i0 + 1, tr->prof_launches > 0 ? tr->prof_launches : 1});
  }
  ProfScope(const ProfScope&) = delete;
};

// Device memory of a trainer comes from a few large arenas (sub-allocated, 4 KiB aligned, freed with the trainer) instead of one
// hipMalloc per buffer. Measured (round 4, tools/ab_headline.sh, one box, alternating processes): with ~60 separate allocations the
// forward chain ran 40.7 us in five of eight processes and 43-44 us in the other three (round 3's unexplained "bimodal between
// processes"); skewing every base by a multiple of 4352 bytes made it 42.3-44.0 us always; 2 MiB-aligned buffers inside one arena
// 41.2-42.5; buffers PACKED into one arena 40.1-40.7 us in every process (and the input-gradient chain 44.2-44.3 instead of 44.8-45.4) --
// one contiguous virtual range maps with large page-table fragments, so the 256 CUs' concurrent row streams over a dozen 5 MiB buffers
// stop missing in the address-translation caches.
// acez_trainer_create sizes its arena exactly: a first pass over its allocation list (tr->sizing) adds the requests up, ONE hipMalloc of
// that sum follows, the second pass hands the pieces out (round 4 rounded every arena up to 256 MiB / eight times the largest request:
// 2.4 GB for an inference context with 300 MB activations, 256 MiB for the short-lived per-call heads of session.scene_coordinates).
// Allocations after creation (acez_trainer_set_buffer: the pose tables) open a chunk of their own, 32 MiB at least.
static int dmalloc(acez_trainer* tr, void** p, size_t bytes) {
  constexpr size_t A = 4096;
  if (tr->sizing) { tr->sized_total += (bytes + A - 1) / A * A; *p = nullptr; return ACEZ_OK; }
  size_t cur = (tr->arena_cursor + A - 1) / A * A;
  if (!tr->arena_base || cur + bytes > tr->arena_size) {
    size_t chunk = std::max(tr->sized_total, bytes) + A;
    tr->sized_total = 0;   // (only the first chunk is pre-sized)
    if (tr->arena_base) chunk = std::max((size_t)8 * bytes, (size_t)32 << 20);
    ACEZ_HIP_CHECK(hipMalloc((void**)&tr->arena_base, chunk));
    tr->allocs.push_back(tr->arena_base);
    tr->arena_size = chunk;
    cur = 0;
  }
  *p = tr->arena_base + cur;
  tr->arena_cursor = cur + bytes;
  return ACEZ_OK;
}

extern "C" int64_t acez_head_num_params(const acez_head_desc* head) {
  if (!head) return -1;
  const int64_t L = 3 + 3 * (int64_t)head->num_head_blocks + 2;
  const int64_t no = head->use_homogeneous ? 4 : 3;
  return L * (512 * 512 + 512) + no * 513;
}

extern "C" void acez_trainer_destroy(acez_trainer* tr) {
  if (!tr) return;
  for (void* p : tr->allocs) (void)hipFree(p);
  for (hipEvent_t e : tr->ev_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : {tr->ev_begin, tr->ev_pose_fwd, tr->ev_loss, tr->ev_pose_bwd})
    if (e) (void)hipEventDestroy(e);
  if (tr->pose_stream) (void)hipStreamDestroy(tr->pose_stream);
  delete tr;
}

// One launch with rowseq_kernel's geometry (grid of a full 64-row-tile batch, 512 threads, the same LDS) that records which XCD
// ran each tile. The one-launch chains are enabled only on a gfx950 whose eight XCDs take the workgroups of a launch round-robin,
// i.e. if the four column tiles of EVERY row tile report one XCD (ACEZ_SEQ_NOPROBE=1 skips the probe: tests of the fault path).
static int seq_placement_probe(acez_trainer* tr) {
  hipDeviceProp_t prop;
  ACEZ_HIP_CHECK(hipGetDeviceProperties(&prop, tr->device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 || tr->n_cus < 256) { tr->seq = false; tr->seq_probe = 0; return ACEZ_OK; }
  if (ACEZ_DIAG_ENV("ACEZ_SEQ_NOPROBE")) return ACEZ_OK;
  uint32_t* d_rec = nullptr;
  ACEZ_HIP_CHECK(hipMalloc((void**)&d_rec, 256 * sizeof(uint32_t)));
  uint32_t h[256];
  bool ok = true;
  for (int rep = 0; rep < 2 && ok; ++rep) {   // the full grid, and a ragged one (41 row tiles: 6 per XCD, the last XCDs short)
    const int mtiles = rep == 0 ? 64 : 41;
    (void)hipMemset(d_rec, 0xff, 256 * sizeof(uint32_t));
    hipLaunchKernelGGL(seq_probe_kernel, dim3(32 * ((mtiles + 7) / 8)), dim3(512), 0, 0, d_rec, mtiles);
    if (hipGetLastError() != hipSuccess || hipMemcpy(h, d_rec, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
    for (int mt = 0; mt < mtiles && ok; ++mt)
      for (int nt = 0; nt < 4; ++nt) {
        const uint32_t x = h[mt * 4 + nt] >> 16, x0 = h[mt * 4] >> 16;
        if (h[mt * 4 + nt] == 0xffffffffu || x > 7 || x != x0) ok = false;
      }
  }
  (void)hipFree(d_rec);
  tr->seq_probe = ok ? 1 : 0;
  if (!ok) tr->seq = false;   // per-layer launches: correct on any placement
  return ACEZ_OK;
}

static void fill_adam_args(acez_trainer* tr, AdamArgs& a);
static void fill_wgrad_args(acez_trainer* tr, WgradArgs& a, int n, const TrainState* st);

// A hand-off poll of wgrad_opt_kernel expired (WgoFaultRec::epoch != 0): unlike a rowseq fault -- whose step is a device-side no-op -- that
// step WAS applied in part: the small parameters, the schedule wave (both read the fault word long before it is raised) and every half
// tile whose exchange completed. The fall-back finishes it instead of leaving the trainer between two steps: since the fault word went up
// every launch that writes a step's buffers has returned at entry (rowseq_kernel, loss_kernel, the gather beside it), so the faulted
// step's activations and gradients are still in place; wgrad_kernel recomputes the slabs from them and wgo_recover_kernel applies AdamW
// to exactly the rows whose wave gave up, with that step's scalars. Result: bitwise the two-launch flow's step (tests/test_wgrad_opt_gpu.py).
static void wgo_recover(acez_trainer* tr, hipStream_t s) {
  if (!tr->wg_rec) return;
  WgoFaultRec rec{};
  if (hipMemcpyAsync(&rec, tr->wg_rec, sizeof(rec), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess || !rec.epoch) return;
  WgradArgs a{};
  fill_wgrad_args(tr, a, rec.M, nullptr);   // (st = null: the fault handler has cleared `active`)
  a.In[0] = rec.in0;
  const int groups = tr->L * tr->nslabs;
  if (tr->f16) hipLaunchKernelGGL(wgrad_kernel<EltF16>, dim3(128 * ((groups + 7) / 8)), dim3(WGRAD_THREADS), 0, s, a);
  else hipLaunchKernelGGL(wgrad_kernel<EltBf16>, dim3(128 * ((groups + 7) / 8)), dim3(WGRAD_THREADS), 0, s, a);
  AdamArgs ad;
  fill_adam_args(tr, ad);
  const dim3 grid(256 * ((tr->L + 7) / 8));
  if (tr->f16) hipLaunchKernelGGL(wgo_recover_kernel<EltF16>, grid, dim3(512), 0, s, ad, (const float*)tr->slabs, tr->n_wide, (const uint32_t*)tr->wg_status, (const WgoFaultRec*)tr->wg_rec, tr->L);
  else hipLaunchKernelGGL(wgo_recover_kernel<EltBf16>, grid, dim3(512), 0, s, ad, (const float*)tr->slabs, tr->n_wide, (const uint32_t*)tr->wg_status, (const WgoFaultRec*)tr->wg_rec, tr->L);
  (void)hipMemsetAsync(tr->wg_rec, 0, sizeof(WgoFaultRec), s);
  ++tr->wgo_recovered;
}

// Read the fault word (the caller has synchronised the stream or is about to): on a fault, reset the hand-off state and switch the
// trainer to per-layer launches. Returns 1 if a fall-back was taken.
static int seq_fault_check(acez_trainer* tr, hipStream_t s) {
  int err = 0;
  if (hipMemcpyAsync(&err, tr->seq_err, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 0;
  if (!err) return 0;
  wgo_recover(tr, s);   // (before the fault word comes down: nothing else may run on this trainer's buffers in between)
  tr->pre_idx = nullptr; tr->pre_n = 0; tr->next_gathered = false;   // batches "gathered ahead" since the fault were held back
  (void)hipMemsetAsync(tr->seq_flags, 0, (64 * 32 + 1) * sizeof(uint32_t), s);   // counters + fault word (the poll budget behind them stays)
  hipLaunchKernelGGL(sched_reactivate_kernel, dim3(1), dim3(64), 0, s, tr->st);
  const int one = 1;
  (void)hipMemcpyAsync(&tr->st_infer->active, &one, sizeof(int), hipMemcpyHostToDevice, s);
  (void)hipStreamSynchronize(s);
  for (int mt = 0; mt < 64; ++mt) tr->seq_base[mt] = 0;
  tr->seq = false;
  ++tr->seq_faults;
  return 1;
}

extern "C" int acez_trainer_seq_status(acez_trainer* tr, int* enabled, int* probe, int* faults) {
  ACEZ_REQUIRE(tr, "null trainer");
  if (enabled) *enabled = tr->seq ? 1 : 0;
  if (probe) *probe = tr->seq_probe;
  if (faults) *faults = tr->seq_faults;
  return ACEZ_OK;
}

extern "C" int acez_trainer_create(acez_trainer** out, const acez_train_config* cfg, const acez_param_buffers* params,
                                   int device) {
  ACEZ_REQUIRE(out && cfg && params, "null pointer");
  ACEZ_REQUIRE(cfg->head.num_head_blocks >= 0 && 3 + 3 * cfg->head.num_head_blocks + 2 <= MAX_LAYERS, "num_head_blocks out of range");
  ACEZ_REQUIRE(cfg->max_batch > 0 && cfg->global_batch > 0, "batch sizes must be positive");
  ACEZ_REQUIRE(params->d_params && params->d_adam_m && params->d_adam_v && params->d_grad, "null parameter buffer");
  ACEZ_REQUIRE(params->n_params == acez_head_num_params(&cfg->head), "n_params does not match the head description");
  ACEZ_REQUIRE(cfg->schedule >= 0 && cfg->schedule <= 2, "unknown schedule");
  ACEZ_REQUIRE(cfg->loss_type >= 0 && cfg->loss_type <= 4, "unknown loss type");
  ACEZ_REQUIRE(cfg->pose_refinement >= 0 && cfg->pose_refinement <= 2, "pose_refinement must be 0 (none), 1 (naive) or 2 (mlp)");
  ACEZ_REQUIRE(cfg->pose_refinement_ortho == 0 || cfg->pose_refinement_ortho == 1, "pose_refinement_ortho must be 0 (gram-schmidt) or 1 (procrustes)");
  ACEZ_REQUIRE(cfg->pose_refinement == 0 || (params->d_pose_params && params->d_pose_m && params->d_pose_v), "pose refinement needs the pose parameter buffers");
  ACEZ_REQUIRE(cfg->pose_refinement != 2 || params->n_pose_params == ACEZ_POSE_MLP_PARAMS, "mlp: n_pose_params must be ACEZ_POSE_MLP_PARAMS");
  ACEZ_REQUIRE(cfg->pose_refinement != 1 || (params->n_pose_params > 0 && params->n_pose_params % 12 == 0), "naive: n_pose_params must be 12 * n_images");
  ACEZ_REQUIRE(cfg->compute_dtype != ACEZ_DTYPE_FP32, "compute_dtype fp32 (train_ace.py --use_half False) is not implemented: choose ACEZ_DTYPE_BF16 or ACEZ_DTYPE_FP16");
  ACEZ_REQUIRE(cfg->compute_dtype == ACEZ_DTYPE_BF16 || cfg->compute_dtype == ACEZ_DTYPE_FP16, "unknown compute_dtype");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    set_error("no HIP device visible: the head kernels need a gfx950 GPU (there is no CPU fallback)");
    return ACEZ_ERR_NODEVICE;
  }
  if (device >= 0) ACEZ_HIP_CHECK(hipSetDevice(device));
  acez_trainer* tr = new (std::nothrow) acez_trainer();
  ACEZ_REQUIRE(tr, "out of host memory");
  ACEZ_HIP_CHECK(hipGetDevice(&tr->device));
  tr->cfg = *cfg;
  tr->f16 = cfg->compute_dtype == ACEZ_DTYPE_FP16;
  if (const char* e = ACEZ_DIAG_ENV("ACEZ_LOSS_ROWS")) tr->loss_rows = (atoi(e) == 8) ? 8 : 4;
  if (const char* e = ACEZ_DIAG_ENV("ACEZ_POSE_FUSED")) tr->pose_fused = atoi(e) != 0;
  if (const char* e = ACEZ_DIAG_ENV("ACEZ_POSE_TILE")) { const int v = atoi(e); tr->pose_tile = tr->pose_tile_fwd = (v == 4 || v == 16) ? v : 8; }
  if (const char* e = ACEZ_DIAG_ENV("ACEZ_POSE_TILE_FWD")) { const int v = atoi(e); tr->pose_tile_fwd = (v == 4 || v == 16) ? v : 8; }
  if (cfg->pose_refinement != 2) tr->pose_fused = false;
  if (cfg->pose_refinement != 0 && !tr->pose_fused && !(ACEZ_DIAG_ENV("ACEZ_POSE_STREAM") && atoi(ACEZ_DIAG_ENV("ACEZ_POSE_STREAM")) == 0)) {
    ACEZ_HIP_CHECK(hipStreamCreateWithFlags(&tr->pose_stream, hipStreamNonBlocking));
    for (hipEvent_t* e : {&tr->ev_begin, &tr->ev_pose_fwd, &tr->ev_loss, &tr->ev_pose_bwd}) ACEZ_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  tr->pb = *params;
  tr->nb = cfg->head.num_head_blocks;
  tr->L = 3 + 3 * tr->nb + 2;
  tr->no = cfg->head.use_homogeneous ? 4 : 3;
  tr->n_wide = (int64_t)tr->L * (512 * 512 + 512);
  tr->fc3_off = tr->n_wide;
  tr->n_params = params->n_params;
  tr->fc3_stride = ((int64_t)tr->no * 513 + 3) & ~3LL;
  tr->max_batch = cfg->max_batch;
  if (const char* e = getenv("ACEZ_SEQ")) tr->seq = atoi(e) != 0;
  {
    hipDeviceProp_t prop;
    ACEZ_HIP_CHECK(hipGetDeviceProperties(&prop, tr->device));
    tr->n_cus = prop.multiProcessorCount;
  }
  // wgrad_kernel: 16 tiles per layer; as many row slabs as fill the 256 CUs
  tr->nslabs = 256 / (16 * tr->L);
  if (tr->nslabs < 1) tr->nslabs = 1;

  int rc = ACEZ_OK;
  const size_t act_bytes = (size_t)tr->max_batch * 512 * sizeof(uint16_t);
  auto A = [&](void** p, size_t bytes) { if (rc == ACEZ_OK) rc = dmalloc(tr, p, bytes); };
  // buffers only a trainer that trains needs (acez_train_config.inference_only = 1: Regressor, session.scene_coordinates): gradients, weight-
  // gradient slabs, partial sums, the second input buffer / metadata table of the pre-gathered next batch and wgrad_opt_kernel's exchange
  // tiles, counters and fault record -- 7.5 of the 14.5 GB of a 128-frame inference context
  const bool trains = cfg->inference_only == 0;
  tr->inference_only = !trains;
  const int wgo_grid = 256 * ((tr->L + 7) / 8);
  for (int pass = 0; pass < 2; ++pass) {   // pass 0 adds the requests up (dmalloc, tr->sizing), pass 1 hands out pieces of ONE exact arena
  tr->sizing = pass == 0;
  A((void**)&tr->Wb, (size_t)tr->L * 262144 * 2);
  A((void**)&tr->WbT, (size_t)tr->L * 262144 * 2);
  A((void**)&tr->W3b, (size_t)tr->no * 512 * 2);
  tr->out.resize(tr->L, nullptr);
  tr->dZ.resize(tr->L, nullptr);
  tr->R.resize(tr->nb + 2, nullptr);
  for (int l = 0; l < tr->L; ++l) { A((void**)&tr->out[l], act_bytes); if (trains) A((void**)&tr->dZ[l], act_bytes); }
  for (int b = 0; b < tr->nb + 2; ++b) A((void**)&tr->R[b], act_bytes);
  const int max_loss_blocks = (tr->max_batch + 4 * tr->loss_rows - 1) / (4 * tr->loss_rows);
  tr->bias_layer_stride = (int64_t)max_loss_blocks * 512;
  if (trains) {
    A((void**)&tr->dR[0], act_bytes);
    A((void**)&tr->dR[1], act_bytes);
    tr->mask_stride = (size_t)((tr->max_batch + 79) / 80) * 4 * 4 * 64;
    A((void**)&tr->maskbits, (size_t)tr->L * tr->mask_stride * sizeof(uint2));
    A((void**)&tr->slabs, (size_t)tr->nslabs * tr->n_wide * sizeof(float));
    A((void**)&tr->fc3_partials, (size_t)max_loss_blocks * tr->fc3_stride * sizeof(float));
    A((void**)&tr->stat_partials, (size_t)max_loss_blocks * 4 * sizeof(float));
    A((void**)&tr->bias_partials, (size_t)tr->L * tr->bias_layer_stride * sizeof(float));
    A((void**)&tr->xyz, (size_t)tr->max_batch * 3 * sizeof(float));
    A((void**)&tr->batch_meta, (size_t)tr->max_batch * sizeof(int4));
  }
  A((void**)&tr->seq_flags, (64 * 32 + 32) * sizeof(uint32_t));
  if (ACEZ_DIAG_ENV("ACEZ_SEQ_XCC")) A((void**)&tr->seq_xcc, (8 + 256) * sizeof(uint32_t));
  if (ACEZ_DIAG_ENV("ACEZ_POSE_TRACE")) A((void**)&tr->pose_trace, 3 * 1024 * 16 * sizeof(unsigned long long));
  if (ACEZ_DIAG_ENV("ACEZ_WGO_TRACE")) A((void**)&tr->wgo_trace, 256 * 12 * 8 * sizeof(unsigned long long));
  if (trains) {
    A((void**)&tr->R0_alt, act_bytes);
    A((void**)&tr->batch_meta_alt, (size_t)tr->max_batch * sizeof(int4));
    A((void**)&tr->wg_xch, (size_t)tr->L * 16 * 2 * 8192 * sizeof(float));
    A((void**)&tr->wg_flags, (size_t)tr->L * 16 * 2 * 32 * sizeof(uint32_t));
    A((void**)&tr->wg_status, (size_t)wgo_grid * WGRAD_LOADERS * sizeof(uint32_t));
    A((void**)&tr->wg_rec, sizeof(WgoFaultRec));
  }
  A((void**)&tr->zeros, 1024);
  tr->log_cap = cfg->iterations + 8;
  A((void**)&tr->log_loss, (size_t)tr->log_cap * sizeof(float));
  A((void**)&tr->log_inl, (size_t)tr->log_cap * sizeof(float));
  A((void**)&tr->st_slot[0], sizeof(TrainState));
  A((void**)&tr->st_slot[1], sizeof(TrainState));
  tr->st = tr->st_slot[0];
  A((void**)&tr->st_infer, sizeof(TrainState));
  }   // (allocation passes)
  if (rc != ACEZ_OK) { acez_trainer_destroy(tr); return rc; }
  ACEZ_HIP_CHECK(hipMemset(tr->seq_flags, 0, (64 * 32 + 32) * sizeof(uint32_t)));
  tr->seq_err = reinterpret_cast<int*>(tr->seq_flags + 64 * 32);
  if (const char* e = getenv("ACEZ_SEQ_SPIN_US")) tr->seq_spin_limit = (uint32_t)std::max(1L, atol(e)) * 2u;
  ACEZ_HIP_CHECK(hipMemcpy(tr->seq_flags + 64 * 32 + 1, &tr->seq_spin_limit, sizeof(uint32_t), hipMemcpyHostToDevice));
  if (const char* e = ACEZ_DIAG_ENV("ACEZ_SEQ_FAULT_AT")) tr->seq_fault_at = atol(e);
  if (tr->seq_xcc) ACEZ_HIP_CHECK(hipMemset(tr->seq_xcc, 0, (8 + 256) * sizeof(uint32_t)));
  if (const char* e = ACEZ_DIAG_ENV("ACEZ_WGRAD_OPT")) tr->wgrad_opt = atoi(e) != 0;
  if (const char* e = ACEZ_DIAG_ENV("ACEZ_WGO_FAULT_AT")) tr->wgo_fault_at = atol(e);
  if (const char* e = ACEZ_DIAG_ENV("ACEZ_WGO_FAULT_MOD")) tr->wgo_fault_mod = atoi(e);
  if (tr->pose_trace) ACEZ_HIP_CHECK(hipMemset(tr->pose_trace, 0, 3 * 1024 * 16 * sizeof(unsigned long long)));
  if (tr->wgo_trace) ACEZ_HIP_CHECK(hipMemset(tr->wgo_trace, 0, 256 * 12 * 8 * sizeof(unsigned long long)));
  if (trains) {
    ACEZ_HIP_CHECK(hipMemset(tr->wg_flags, 0, (size_t)tr->L * 16 * 2 * 32 * sizeof(uint32_t)));
    ACEZ_HIP_CHECK(hipMemset(tr->wg_status, 0, (size_t)wgo_grid * WGRAD_LOADERS * sizeof(uint32_t)));
    ACEZ_HIP_CHECK(hipMemset(tr->wg_rec, 0, sizeof(WgoFaultRec)));
  } else {
    tr->wgrad_opt = false;   // (no exchange tiles: an inference-only head never runs the fused step)
  }

  SchedConfig& sc = tr->sc;
  sc.schedule = cfg->schedule; sc.iterations = cfg->iterations; sc.warmup_iterations = cfg->warmup_iterations;
  sc.cooldown_iterations = cfg->cooldown_iterations; sc.loss_type = cfg->loss_type;
  sc.circle_schedule = cfg->circle_schedule; sc.refine_calibration = cfg->refine_calibration;
  sc.soft_clamp = cfg->soft_clamp; sc.soft_clamp_min = cfg->soft_clamp_min;
  sc.lr_min = cfg->lr_min; sc.lr_max = cfg->lr_max; sc.warmup_lr = cfg->warmup_lr;
  sc.cooldown_trigger_percent = cfg->cooldown_trigger_percent;
  sc.beta1 = cfg->beta1; sc.beta2 = cfg->beta2; sc.eps = cfg->eps; sc.weight_decay = cfg->weight_decay;
  sc.calib_lr = cfg->calib_lr;
  sc.pose_refinement = cfg->pose_refinement; sc.pose_wait = cfg->pose_refinement_wait; sc.pose_lr = cfg->pose_refinement_lr;
  sc.f16 = tr->f16 ? 1 : 0;
  hipLaunchKernelGGL(sched_init_kernel, dim3(1), dim3(64), 0, 0, tr->st, sc);
  ACEZ_HIP_CHECK(hipGetLastError());
  ACEZ_HIP_CHECK(hipMemset(tr->zeros, 0, 1024));
  { TrainState one{}; one.active = 1; ACEZ_HIP_CHECK(hipMemcpy(tr->st_infer, &one, sizeof(TrainState), hipMemcpyHostToDevice)); }
  ACEZ_HIP_CHECK(hipMemset(tr->log_loss, 0, (size_t)tr->log_cap * sizeof(float)));
  ACEZ_HIP_CHECK(hipMemset(tr->log_inl, 0, (size_t)tr->log_cap * sizeof(float)));
  ACEZ_HIP_CHECK(hipDeviceSynchronize());
  if (tr->seq) {
    const int prc = seq_placement_probe(tr);
    if (prc != ACEZ_OK) { acez_trainer_destroy(tr); return prc; }
  }
  *out = tr;
  return ACEZ_OK;
}

extern "C" int acez_trainer_set_buffer(acez_trainer* tr, const acez_train_buffer* buf) {
  ACEZ_REQUIRE(tr && buf, "null pointer");
  ACEZ_REQUIRE(!tr->inference_only, "an inference-only context (acez_train_config.inference_only) takes no training buffer");
  ACEZ_REQUIRE(buf->d_features && buf->d_target_px && buf->d_view_idx && buf->n_patches > 0, "empty patch buffer");
  ACEZ_REQUIRE(buf->d_view_aug_inv && buf->d_view_K && buf->d_view_Kinv && buf->d_view_image && buf->n_views > 0, "empty view table");
  ACEZ_REQUIRE(buf->d_image_pose_inv && buf->n_images > 0, "empty pose table");
  tr->buf = *buf;
  tr->have_buf = true;
  tr->pre_idx = nullptr; tr->pre_n = 0;   // rows gathered ahead (acez_train_step_next) came from the buffer before
  if (tr->cfg.pose_refinement == 1) {
    ACEZ_REQUIRE(tr->pb.n_pose_params == (int64_t)buf->n_images * 12, "naive pose refinement: n_pose_params != 12 * n_images");
    if (tr->pose_images < buf->n_images) {
      ACEZ_HIP_CHECK(hipSetDevice(tr->device));
      int rc = ACEZ_OK;
      auto A = [&](void** p, size_t bytes) { if (rc == ACEZ_OK) rc = dmalloc(tr, p, bytes); };
      A((void**)&tr->pdT, (size_t)buf->n_images * 12 * sizeof(float));
      A((void**)&tr->pose_cur, (size_t)buf->n_images * 16 * sizeof(float));
      A((void**)&tr->pa1, (size_t)buf->n_images * 16 * sizeof(float));   // zero table standing in for T0
      if (!tr->row_dT) {
        A((void**)&tr->row_dT, (size_t)tr->max_batch * 12 * sizeof(float));
        A((void**)&tr->row_image, (size_t)tr->max_batch * sizeof(int));
      }
      if (rc != ACEZ_OK) return rc;
      ACEZ_HIP_CHECK(hipMemset(tr->pa1, 0, (size_t)buf->n_images * 16 * sizeof(float)));
      tr->pose_images = buf->n_images;
    }
  }
  if (tr->cfg.pose_refinement == 2 && tr->pose_images < buf->n_images) {
    ACEZ_HIP_CHECK(hipSetDevice(tr->device));
    const int I = buf->n_images;
    int rc = ACEZ_OK;
    auto A = [&](void** p, size_t bytes) { if (rc == ACEZ_OK) rc = dmalloc(tr, p, bytes); };
    float** wide[] = {&tr->pa1, &tr->pa2, &tr->pa3, &tr->pr, &tr->pf1, &tr->pf2, &tr->pdz2, &tr->pdz1, &tr->pdr, &tr->pdzc3, &tr->pdzc2, &tr->pdzc1};
    for (float** p : wide) A((void**)p, (size_t)I * 128 * sizeof(float));
    A((void**)&tr->pdlt, (size_t)I * 12 * sizeof(float));
    A((void**)&tr->pdT, (size_t)I * 12 * sizeof(float));
    A((void**)&tr->pddelta, (size_t)I * 12 * sizeof(float));
    A((void**)&tr->pose_cur, (size_t)I * 16 * sizeof(float));
    A((void**)&tr->pose_wt, (size_t)4 * 128 * 128 * sizeof(float));
    if (!tr->row_dT) {
      A((void**)&tr->row_dT, (size_t)tr->max_batch * 12 * sizeof(float));
      A((void**)&tr->row_image, (size_t)tr->max_batch * sizeof(int));
    }
    if (rc != ACEZ_OK) return rc;
    tr->pose_images = I;
  }
  return ACEZ_OK;
}

static void fill_adam_args(acez_trainer* tr, AdamArgs& a) {
  a.params = tr->pb.d_params; a.m = tr->pb.d_adam_m; a.v = tr->pb.d_adam_v; a.grad = tr->pb.d_grad;
  a.Wb = tr->Wb; a.WbT = tr->WbT; a.W3b = tr->W3b;
  for (int l = 0; l < tr->L; ++l) { a.w_off[l] = (int64_t)l * 262656; a.b_off[l] = a.w_off[l] + 262144; }
  a.fc3_off = tr->fc3_off; a.n_fc3 = (int64_t)tr->no * 513; a.n_params = tr->n_params;
  a.n_layers = tr->L; a.no = tr->no; a.st = tr->st;
  a.slabs = nullptr; a.nslabs = 0; a.slab_stride = 0;
  a.fault = tr->seq_err; a.f16 = tr->f16 ? 1 : 0;
  a.layer_lo = 0; a.layer_hi = tr->L;
}

extern "C" int acez_trainer_sync_weights(acez_trainer* tr, void* stream) {
  ACEZ_REQUIRE(tr, "null trainer");
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  AdamArgs a;
  fill_adam_args(tr, a);
  const int nblk = tr->L * 64 + (tr->no * 512 + 255) / 256;
  hipLaunchKernelGGL(recast_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
  ACEZ_HIP_CHECK(hipGetLastError());
  tr->pose_wt_valid = false;   // the caller may have rewritten the pose parameters as well
  tr->pre_idx = nullptr; tr->pre_n = 0;   // a restart point: the next step gathers its own batch
  return ACEZ_OK;
}

// forward chain on n rows whose input features are in `in0`; returns the fc2 output buffer
// rowseq_kernel is usable when its workgroups wait for each other safely: all of them resident at once
static bool seq_usable(const acez_trainer* tr, int n) {
  const int mtiles = (n + 79) / 80;
  return tr->seq && mtiles <= 64 && 32 * ((mtiles + 7) / 8) <= tr->n_cus;
}

// wgrad_opt_kernel's workgroups wait for each other too: same conditions, its own grid (both slabs of a layer on one XCD: 32 workgroups
// per layer, layers round-robin over the XCDs)
// Without pose refinement only: with it the optimiser launch also carries the pose network's per-image reduction and backward chain
// (adamw_pose_kernel), 18 us that the weight tiles used to run beside for free -- measured with wgrad_opt on that path: 191.6 us per
// step against 176.7 (profiles/r04_*_trace_mlp.csv history in DESIGN.md section 3).
static bool wgrad_opt_usable(const acez_trainer* tr) {
  return tr->wgrad_opt && tr->seq && tr->nslabs == 2 &&
         tr->cfg.pose_refinement == 0 && 256 * ((tr->L + 7) / 8) <= tr->n_cus;
}

template <bool BWD>
static void launch_rowseq(acez_trainer* tr, const std::vector<SeqLayer>& layers, int n, const TrainState* st, hipStream_t s) {
  const int mtiles = (n + 79) / 80;
  for (size_t i0 = 0; i0 < layers.size(); i0 += SEQ_MAX_LAYERS) {
    RowSeqArgs a{};
    const int cnt = (int)std::min<size_t>(SEQ_MAX_LAYERS, layers.size() - i0);
    for (int i = 0; i < cnt; ++i) a.layer[i] = layers[i0 + i];
    a.n_layers = cnt; a.M = n; a.st = st ? st : tr->st_infer; a.flags = tr->seq_flags; a.xcc_dbg = tr->seq_xcc; a.spin_limit = tr->seq_spin_limit;
    for (int mt = 0; mt < 64; ++mt) a.base[mt] = tr->seq_base[mt];
    // tests (ACEZ_SEQ_FAULT_AT): this launch is told that a million seams have completed before it -- every hand-off then waits
    // for a count that never comes, which is what a sibling on a foreign XCD looks like
    if (cnt > 1 && tr->seq_launches == tr->seq_fault_at)
      for (int mt = 0; mt < 64; ++mt) a.base[mt] += 1u << 20;
    ++tr->seq_launches;
    if (tr->f16) hipLaunchKernelGGL((rowseq_kernel<BWD, EltF16>), dim3(32 * ((mtiles + 7) / 8)), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(rowseq_kernel<BWD>, dim3(32 * ((mtiles + 7) / 8)), dim3(512), 0, s, a);
    for (int mt = 0; mt < mtiles; ++mt) tr->seq_base[mt] += (uint32_t)(cnt - 1);
    tr->prof_launches += cnt;   // accounted as layer GEMMs so that the per-layer average stays comparable
  }
}

static uint16_t* launch_forward(acez_trainer* tr, const uint16_t* in0, int n, const TrainState* st, hipStream_t s) {
  ProfScope chain_scope(tr, s, KC_GEMM_FWD);  // one event pair around the whole chain of dependent GEMM launches
  const float* P = tr->pb.d_params;
  const bool seq = seq_usable(tr, n);
  std::vector<SeqLayer> sq;
  // a training forward (st != null) leaves every layer's ReLU mask as bits for the input-gradient chain; fc2's mask is applied by the loss kernel
  auto mask_out = [&](int l) -> uint2* { return (st && tr->maskbits && l != 3 * (tr->nb + 1) + 1) ? tr->maskbits + (size_t)l * tr->mask_stride : nullptr; };
  auto gemm = [&](int l, const uint16_t* in, uint16_t* out_main, const uint16_t* res, uint16_t* out_aux) {
    if (seq) {
      SeqLayer y{};
      y.In = in; y.W = tr->Wb + (size_t)l * 262144; y.bias = P + (int64_t)l * 262656 + 262144; y.res = res; y.out_main = out_main;
      y.out_aux = out_aux; y.aux_mode = res ? AUX_RESIDUAL : AUX_NONE;
      y.mask_out = mask_out(l);
      sq.push_back(y);
      return;
    }
    RowGemmArgs g{};
    g.In = in; g.W = tr->Wb + (size_t)l * 262144; g.bias = P + (int64_t)l * 262656 + 262144;
    g.add = nullptr; g.mask_out = mask_out(l); g.mask_in = nullptr; g.res = res; g.out_main = out_main; g.out_aux = out_aux;
    g.M = n; g.N = 512; g.K = 512; g.relu = 1; g.aux_mode = res ? AUX_RESIDUAL : AUX_NONE; g.st = st; g.dbg = 0; g.bias_partials = nullptr;
    launch_rowgemm(g, s, tr->f16);
    ++tr->prof_launches;
  };
  const uint16_t* r = in0;
  for (int b = 0; b <= tr->nb; ++b) {
    gemm(3 * b, r, tr->out[3 * b], nullptr, nullptr);
    gemm(3 * b + 1, tr->out[3 * b], tr->out[3 * b + 1], nullptr, nullptr);
    // the block's last activation is only ever read again as a ReLU mask (its sum with the residual stream is the next input, its
    // weight-gradient operand is the layer before): a training forward leaves the mask bits and does not store the tile (5 MB per block)
    gemm(3 * b + 2, tr->out[3 * b + 1], mask_out(3 * b + 2) ? nullptr : tr->out[3 * b + 2], r, tr->R[b + 1]);
    r = tr->R[b + 1];
  }
  const int f1 = 3 * (tr->nb + 1), f2 = f1 + 1;
  gemm(f1, r, tr->out[f1], nullptr, nullptr);
  gemm(f2, tr->out[f1], tr->out[f2], nullptr, nullptr);
  if (seq) launch_rowseq<false>(tr, sq, n, st, s);
  return tr->out[f2];
}

static void fill_loss_head(acez_trainer* tr, LossArgs& a) {
  a.W3 = tr->W3b; a.b3 = tr->pb.d_params + tr->fc3_off + (int64_t)tr->no * 512;
  a.no = tr->no; a.use_homogeneous = tr->cfg.head.use_homogeneous;
  for (int i = 0; i < 3; ++i) a.mean[i] = tr->cfg.head.mean[i];
  a.max_inv_scale = tr->cfg.head.max_inv_scale; a.min_inv_scale = tr->cfg.head.min_inv_scale; a.h_beta = tr->cfg.head.h_beta;
}


static void launch_loss(acez_trainer* tr, int nblk, hipStream_t s, const LossArgs& a) {
  if (tr->loss_rows == 4) {
    if (tr->f16) hipLaunchKernelGGL((loss_kernel<EltF16, 4>), dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((loss_kernel<EltBf16, 4>), dim3(nblk), dim3(256), 0, s, a);
  } else {
    if (tr->f16) hipLaunchKernelGGL((loss_kernel<EltF16, 8>), dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((loss_kernel<EltBf16, 8>), dim3(nblk), dim3(256), 0, s, a);
  }
}

// training-mode arguments of the loss kernel
// where the gather launches of a batch leave the per-row metadata for the loss kernel
static GatherMeta gather_meta(acez_trainer* tr) {
  GatherMeta m{};
  m.dst = tr->batch_meta; m.view_idx = tr->buf.d_view_idx; m.view_image = tr->buf.d_view_image; m.target_px = tr->buf.d_target_px;
  // EVERY training gather holds while the sticky fault word is up: a faulted wgrad_opt step is finished by the host from the
  // rows still in R[0] (wgo_recover; rec.in0), and a step queued behind the fault -- announced or not -- must not overwrite them. The
  // faulted launch's schedule wave has already activated the next state slot, so gather_kernel's `active` test alone does not hold them.
  m.hold = tr->seq_err;
  return m;
}

static void fill_loss_train(acez_trainer* tr, LossArgs& a, const uint16_t* act, const int64_t* d_indices, int n, bool pose_tables) {
  const int f2 = 3 * (tr->nb + 1) + 1;
  fill_loss_head(tr, a);
  a.act = act; a.n = n;
  a.idx = d_indices; a.meta = tr->batch_meta; a.target_px = tr->buf.d_target_px; a.target_crds = tr->buf.d_target_crds; a.view_idx = tr->buf.d_view_idx;
  a.view_aug_inv = tr->buf.d_view_aug_inv; a.view_K = tr->buf.d_view_K; a.view_Kinv = tr->buf.d_view_Kinv;
  a.view_image = tr->buf.d_view_image; a.image_pose_inv = pose_tables ? tr->pose_cur : tr->buf.d_image_pose_inv;
  a.row_dT = pose_tables ? tr->row_dT : nullptr; a.row_image = pose_tables ? tr->row_image : nullptr;
  a.loss_type = tr->cfg.loss_type; a.refine_calibration = tr->cfg.refine_calibration;
  a.hard_clamp = tr->cfg.hard_clamp; a.depth_min = tr->cfg.depth_min; a.depth_max = tr->cfg.depth_max;
  a.depth_target = tr->cfg.depth_target; a.inlier_px = tr->cfg.inlier_px_threshold;
  a.inv_batch = 1.0f / (float)tr->cfg.global_batch; a.focal_init = tr->cfg.focal_init;
  a.st = tr->st; a.out_xyz = tr->xyz; a.dZ = tr->dZ[f2];
  a.fc3_partials = tr->fc3_partials; a.fc3_stride = tr->fc3_stride; a.stat_partials = tr->stat_partials;
  a.bias_partials = tr->bias_partials + (size_t)f2 * tr->bias_layer_stride; a.dbg = 0;
  a.absmax = tr->f16 ? tr->st->dz_absmax_slots : nullptr;
  a.fault = tr->seq_err;
  if (const char* e = ACEZ_DIAG_ENV("ACEZ_LOSS_DBG")) a.dbg = atoi(e);   // ablation: 1 = stop after phase A, 2 = after phase B (timing only)
}

// ---- pose refinement (the flat parameter offsets in PoseNetwork.named_parameters() order are PN_* in pose_kernels.hip)
static PoseNetArgs pose_net_args(acez_trainer* tr, const int* active, int trace_slot = 0) {
  PoseNetArgs a{};
  if (tr->pose_trace && (tr->buf.n_images + 3) / 4 <= 1024) a.trace = tr->pose_trace + (size_t)trace_slot * 1024 * 16;
  a.P = tr->pb.d_pose_params; a.T0 = tr->buf.d_image_pose_inv; a.I = tr->buf.n_images; a.w = tr->cfg.pose_refinement_weight;
  a.a1 = tr->pa1; a.a2 = tr->pa2; a.a3 = tr->pa3; a.r = tr->pr; a.f1 = tr->pf1; a.f2 = tr->pf2; a.delta = tr->pdlt; a.pose_cur = tr->pose_cur;
  a.dT = tr->pdT; a.ddelta = tr->pddelta; a.dz2 = tr->pdz2; a.dz1 = tr->pdz1; a.dr = tr->pdr; a.dzc3 = tr->pdzc3; a.dzc2 = tr->pdzc2;
  a.dzc1 = tr->pdzc1; a.active = active; a.Wt = tr->pose_wt; a.ortho = tr->cfg.pose_refinement_ortho;
  return a;
}

// refined poses of all images (PoseRefiner._predict_pose_updates, refine_poses.py:152-176): one fused launch
static void pose_forward(acez_trainer* tr, const int* active, hipStream_t s) {
  const PoseNetArgs a = pose_net_args(tr, active);
  hipLaunchKernelGGL(pose_transpose_kernel, dim3(4, 4, 4), dim3(256), 0, s, a.P, tr->pose_wt, active);
  hipLaunchKernelGGL(pose_mlp_fwd_kernel, dim3((a.I + PN_IMG - 1) / PN_IMG), dim3(256), 0, s, a);
}

// the layer tables of pose_mlp_wgrad_kernel; fuse: AdamW (+ refreshed transposed copies) in the tile epilogue (single-GPU step)
static void launch_pose_wgrad(acez_trainer* tr, const int* active, bool fuse, hipStream_t s) {
  const int I = tr->buf.n_images;
  PoseWgradArgs w{};
  const float* T0 = tr->buf.d_image_pose_inv;
  const float* dY[7] = {tr->pddelta, tr->pdz2, tr->pdz1, tr->pdr, tr->pdzc3, tr->pdzc2, tr->pdzc1};
  const float* X[7] = {tr->pf2, tr->pf1, tr->pr, T0, tr->pa2, tr->pa1, T0};
  const int O[7] = {12, 128, 128, 128, 128, 128, 128}, K[7] = {128, 128, 128, 12, 128, 128, 12}, XP[7] = {128, 128, 128, 16, 128, 128, 16};
  const int64_t OW[7] = {PN_F3_W, PN_F2_W, PN_F1_W, PN_SKIP_W, PN_C3_W, PN_C2_W, PN_C1_W};
  const int64_t OB[7] = {PN_F3_B, PN_F2_B, PN_F1_B, PN_SKIP_B, PN_C3_B, PN_C2_B, PN_C1_B};
  const int WT[7] = {-1, 3, 2, -1, 1, 0, -1};   // pose_wt holds conv2, conv3, fc1, fc2 (pose_transpose_kernel)
  int jobs = 0;
  for (int l = 0; l < 7; ++l) {
    w.dY[l] = dY[l]; w.X[l] = X[l]; w.O[l] = O[l]; w.K[l] = K[l]; w.xpitch[l] = XP[l]; w.offW[l] = OW[l]; w.offB[l] = OB[l];
    w.wt_slot[l] = WT[l];
    w.job_start[l] = jobs;
    jobs += ((O[l] + 15) / 16) * ((K[l] + 15) / 16);
  }
  w.job_start[7] = jobs;
  w.I = I; w.grad = tr->pb.d_grad + tr->n_params + 4; w.active = active;
  w.fuse = fuse ? 1 : 0; w.p = tr->pb.d_pose_params; w.m = tr->pb.d_pose_m; w.v = tr->pb.d_pose_v; w.Wt = tr->pose_wt;
  w.sc = &tr->st->pose_adam; w.enable = &tr->st->pose_enable; w.fault = tr->seq_err;
  if (tr->pose_trace && jobs <= 1024) w.trace = tr->pose_trace + (size_t)2 * 1024 * 16;
  static const int wb = ACEZ_DIAG_ENV("ACEZ_POSE_WB") ? atoi(ACEZ_DIAG_ENV("ACEZ_POSE_WB")) : 16;   // operand steps requested per round trip
  static const int ww = ACEZ_DIAG_ENV("ACEZ_POSE_WW") ? atoi(ACEZ_DIAG_ENV("ACEZ_POSE_WW")) : 8;    // waves per workgroup
  if (wb == 16 && ww == 4) hipLaunchKernelGGL((pose_mlp_wgrad_kernel<16, 4>), dim3(jobs), dim3(256), 0, s, w);
  else if (wb == 16) hipLaunchKernelGGL((pose_mlp_wgrad_kernel<16, 8>), dim3(jobs), dim3(512), 0, s, w);
  else if (ww == 4) hipLaunchKernelGGL((pose_mlp_wgrad_kernel<32, 4>), dim3(jobs), dim3(256), 0, s, w);
  else hipLaunchKernelGGL((pose_mlp_wgrad_kernel<32, 8>), dim3(jobs), dim3(512), 0, s, w);
}

static void launch_pose_grad_reduce(acez_trainer* tr, int n, const int* active, hipStream_t s) {
  const int I = tr->buf.n_images;
  hipLaunchKernelGGL(pose_grad_reduce2_kernel, dim3((I + 15) / 16), dim3(256), 0, s, (const float*)tr->row_dT, (const int*)tr->row_image, n,
                     tr->pdT, I, active);
}

// gradient of the pose network from the per-row pose gradients of the loss kernel -> d_grad tail (4 launches)
static void pose_backward(acez_trainer* tr, int n, const int* active, hipStream_t s) {
  const int I = tr->buf.n_images;
  launch_pose_grad_reduce(tr, n, active, s);
  const PoseNetArgs a = pose_net_args(tr, active);
  hipLaunchKernelGGL(pose_mlp_bwd_kernel, dim3((I + PN_IMG - 1) / PN_IMG), dim3(256), 0, s, a);
  launch_pose_wgrad(tr, active, false, s);
}

// after a launch that ran the schedule bookkeeping (it wrote the other slot): every later launch reads that slot
static void st_flip(acez_trainer* tr) {
  tr->st_cur ^= 1;
  tr->st = tr->st_slot[tr->st_cur];
}

static PostArgs post_args(acez_trainer* tr) {
  PostArgs p;
  p.src = tr->st; p.st = tr->st_slot[tr->st_cur ^ 1]; p.c = tr->sc; p.grad_stats = (const float*)(tr->pb.d_grad + tr->n_params);
  p.inv_global_batch = 1.0f / (float)tr->cfg.global_batch; p.log_loss = tr->log_loss; p.log_inl = tr->log_inl; p.log_cap = tr->log_cap;
  p.fault = tr->seq_err; p.stat_partials = tr->stat_partials; p.n_loss_blocks = tr->last_nblk;
  return p;
}

// The schedule bookkeeping that closes a step (sched_post) is deferred: normally it is executed by an extra workgroup of
// the next step's gather launch. Every entry point that reads the schedule state or the log flushes it first.
static void flush_post(acez_trainer* tr, hipStream_t s) {
  if (!tr->post_pending) return;
  tr->post_pending = false;
  ProfScope ps(tr, s, KC_SCHED);
  const PostArgs p = post_args(tr);
  hipLaunchKernelGGL(sched_post_kernel, dim3(1), dim3(64), 0, s, p.src, p.st, p.c, p.grad_stats, p.inv_global_batch, p.log_loss, p.log_inl, p.log_cap, p.fault,
                     p.stat_partials, p.n_loss_blocks);
  st_flip(tr);
}

// operands of the weight-gradient launch of a step on n rows (wgrad_kernel / wgrad_opt_kernel; wgo_recover runs it again on a faulted step)
static void fill_wgrad_args(acez_trainer* tr, WgradArgs& a, int n, const TrainState* st) {
  const int f1 = 3 * (tr->nb + 1), f2 = f1 + 1;
  for (int l = 0; l < tr->L; ++l) {
    a.dZ[l] = tr->dZ[l];
    a.w_off[l] = (int64_t)l * 262656; a.b_off[l] = a.w_off[l] + 262144;
  }
  for (int b = 0; b <= tr->nb; ++b) {
    a.In[3 * b] = tr->R[b]; a.In[3 * b + 1] = tr->out[3 * b]; a.In[3 * b + 2] = tr->out[3 * b + 1];
  }
  a.In[f1] = tr->R[tr->nb + 1]; a.In[f2] = tr->out[f1];
  a.slabs = tr->slabs; a.slab_stride = tr->n_wide; a.M = n; a.nslabs = tr->nslabs; a.n_layers = tr->L; a.st = st; a.zeros = tr->zeros; a.dbg = 0;
}

static int train_backward_impl(acez_trainer* tr, const int64_t* d_indices, int n, void* stream, bool fused, const int64_t* d_next = nullptr,
                               int n_next = 0) {
  ACEZ_REQUIRE(tr && d_indices, "null pointer");
  ACEZ_REQUIRE(!tr->inference_only, "an inference-only context (acez_train_config.inference_only) cannot train");
  ACEZ_REQUIRE(tr->have_buf, "acez_trainer_set_buffer has not been called");
  ACEZ_REQUIRE(n > 0 && n <= tr->max_batch, "n must be in [1, max_batch]");
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  hipStream_t s = (hipStream_t)stream;
  const TrainState* st = tr->st;   // (re-read below: the launch that closes the previous step moves the state to the other slot)
  tr->last_n = n;

  const bool pose_naive = tr->cfg.pose_refinement == 1;
  const bool pose_mlp = tr->cfg.pose_refinement == 2 || pose_naive;   // both need the refined-pose table and per-row pose gradients
  // Pose refinement on its own stream: the refined poses are needed by the loss phase only, so their launches run beside the
  // head's forward chain; the pose-gradient launches run beside the input-gradient chain and wgrad.
  hipStream_t ps = (pose_mlp && tr->pose_stream) ? tr->pose_stream : s;
  // mlp refinement folded into the step's own launches (pose_fused.hip): forward beside the gather, backward beside / behind the optimiser
  const bool pf = tr->pose_fused;
  const int f1 = 3 * (tr->nb + 1), f2 = f1 + 1;
  const int nblk = (n + 4 * tr->loss_rows - 1) / (4 * tr->loss_rows);
  auto pose_fwd_launches = [&](hipStream_t q) {
    if (tr->cfg.pose_refinement == 2) pose_forward(tr, &tr->st->active, q);
    if (pose_naive)   // refine_poses.py:224-234: the poses themselves are the parameters; P = 0 + 1 * params, then Gram-Schmidt
      hipLaunchKernelGGL(pose_compose_kernel, dim3((tr->buf.n_images + 255) / 256), dim3(256), 0, q, (const float*)tr->pa1,
                         (const float*)tr->pb.d_pose_params, 1.0f, tr->pose_cur, tr->buf.n_images, (const int*)&tr->st->active, tr->cfg.pose_refinement_ortho);
  };
  auto pose_bwd_launches = [&](hipStream_t q) {
    if (tr->cfg.pose_refinement == 2) pose_backward(tr, n, &tr->st->active, q);
    if (pose_naive) {
      const int I = tr->buf.n_images;
      launch_pose_grad_reduce(tr, n, (const int*)&tr->st->active, q);
      hipLaunchKernelGGL(pose_compose_bwd_kernel, dim3((I + 255) / 256), dim3(256), 0, q, (const float*)tr->pa1, (const float*)tr->pb.d_pose_params, 1.0f,
                         (const float*)tr->pdT, tr->pb.d_grad + tr->n_params + 4, I, (const int*)&tr->st->active, tr->cfg.pose_refinement_ortho);
    }
  };

  uint16_t* act = nullptr;
  {
  // (pose refinement folded into the step's launches: the batch may have been gathered ahead too -- beside the loss kernel of the step
  // before, round 5 -- but the launch below still runs: the pose network's forward and the schedule wave ride in it, with no gather blocks)
  const bool pre_ok = tr->pre_idx == d_indices && tr->pre_n == n;
  const bool pregathered = !pf && pre_ok && !tr->post_pending;
  tr->pre_idx = nullptr; tr->pre_n = 0;
  if (pregathered) {
    // acez_train_step_next of the step before has gathered exactly this batch into R[0] and closed that step's bookkeeping
  } else {
  if (pf && !tr->pose_wt_valid) {   // first step / after acez_trainer_sync_weights / after a split (backward + update) step
    hipLaunchKernelGGL(pose_transpose_kernel, dim3(4, 4, 4), dim3(256), 0, s, (const float*)tr->pb.d_pose_params, tr->pose_wt, (const int*)nullptr);
    tr->pose_wt_valid = true;
  }
  ProfScope* psg = new ProfScope(tr, s, KC_GATHER);
  const int gblocks = (pf && pre_ok) ? 0 : ((n + 3) / 4 < 1024 ? (n + 3) / 4 : 1024);
  if (pf) {   // + the pose network's forward for all images, as the first workgroups of the same launch
    const int T = tr->pose_tile_fwd, np = (tr->buf.n_images + T - 1) / T;
    const int do_post = tr->post_pending ? 1 : 0;
    tr->post_pending = false;
#define ACEZ_SBP(TT) hipLaunchKernelGGL(step_begin_pose_kernel<TT>, dim3(np + gblocks + 1), dim3(pose_fwd_threads<TT>()), 0, s, (const uint16_t*)tr->buf.d_features, \
                                        d_indices, tr->R[0], n, post_args(tr), do_post, pose_net_args(tr, nullptr), np, gather_meta(tr))
    if (T == 16) ACEZ_SBP(16); else if (T == 4) ACEZ_SBP(4); else ACEZ_SBP(8);
#undef ACEZ_SBP
    if (do_post) st_flip(tr);
  } else if (tr->post_pending) {   // gather of this step + the schedule bookkeeping of the previous one, in one launch
    tr->post_pending = false;
    hipLaunchKernelGGL(step_begin_kernel, dim3(gblocks + 1), dim3(256), 0, s, (const uint16_t*)tr->buf.d_features, d_indices, tr->R[0], n,
                       post_args(tr), gather_meta(tr));
    st_flip(tr);
  } else {
    hipLaunchKernelGGL(gather_kernel, dim3(gblocks), dim3(256), 0, s, (const uint16_t*)tr->buf.d_features, d_indices, tr->R[0], n, st, gather_meta(tr));
  }
  delete psg;
  }
  }
  st = tr->st;   // the slot the bookkeeping (if any rode with the gather) has just written
  if (ps != s) {   // after the schedule bookkeeping of step_begin (the pose kernels read st->active / pose_enable)
    ACEZ_HIP_CHECK(hipEventRecord(tr->ev_begin, s));
    ACEZ_HIP_CHECK(hipStreamWaitEvent(ps, tr->ev_begin, 0));
  }
  if (!pf) pose_fwd_launches(ps);
  if (ps != s) ACEZ_HIP_CHECK(hipEventRecord(tr->ev_pose_fwd, ps));
  act = launch_forward(tr, tr->R[0], n, st, s);
  if (ps != s) ACEZ_HIP_CHECK(hipStreamWaitEvent(s, tr->ev_pose_fwd, 0));   // the loss kernel projects with the refined poses

  {
    LossArgs a{};
    fill_loss_train(tr, a, act, d_indices, n, pose_mlp);
    tr->last_nblk = nblk;   // (after this step's step_begin, whose schedule wave closed the step BEFORE with that step's count)
    ProfScope ps(tr, s, KC_LOSS);
    tr->next_gathered = false;
    if (fused && d_next && n_next > 0 && (wgrad_opt_usable(tr) || pf) && tr->loss_rows == 4 && tr->R0_alt) {
      // the next batch's gather as extra workgroups of the loss launch (this path has no optimiser launch to carry it): 32 rows per
      // workgroup and pass, as many workgroups as the loss kernel leaves free (two of these workgroups fit a CU)
      const int want = (n_next + 31) / 32, room = std::max(32, 2 * tr->n_cus - nblk);
      const int gblocks = std::min(want, room);
      GatherMeta gm = gather_meta(tr);
      gm.dst = tr->batch_meta_alt;
      if (tr->f16) hipLaunchKernelGGL((loss_gather_kernel<EltF16, 4>), dim3(nblk + gblocks), dim3(256), 0, s, a, nblk, (const uint16_t*)tr->buf.d_features, d_next, tr->R0_alt, n_next, gm);
      else hipLaunchKernelGGL((loss_gather_kernel<EltBf16, 4>), dim3(nblk + gblocks), dim3(256), 0, s, a, nblk, (const uint16_t*)tr->buf.d_features, d_next, tr->R0_alt, n_next, gm);
      tr->next_gathered = true; tr->next_idx = d_next; tr->next_n = n_next;
    } else {
      launch_loss(tr, nblk, s, a);
    }
  }

  if (ps != s) {   // the pose gradients start from the per-row pose gradients the loss kernel has just written
    ACEZ_HIP_CHECK(hipEventRecord(tr->ev_loss, s));
    ACEZ_HIP_CHECK(hipStreamWaitEvent(ps, tr->ev_loss, 0));
  }
  if (!pf) pose_bwd_launches(ps);
  if (ps != s) ACEZ_HIP_CHECK(hipEventRecord(tr->ev_pose_bwd, ps));

  // input-gradient chain
  const bool seq = seq_usable(tr, n);
  std::vector<SeqLayer> sq;
  auto dgrad = [&](int l, int l_out, const uint16_t* add, uint16_t* out_main, uint16_t* out_aux) {
    const uint2* mask = tr->maskbits + (size_t)l_out * tr->mask_stride;   // the bits forward layer l_out left (RowGemmArgs::mask_in)
    if (seq) {
      if (add && !out_aux) abort();   // rowseq_kernel<true> has the three epilogue shapes of this chain only (an `add` comes with a residual-gradient output)
      SeqLayer y{};
      y.In = tr->dZ[l]; y.W = tr->WbT + (size_t)l * 262144; y.add = add; y.mask_in = mask; y.out_main = out_main; y.out_aux = out_aux;
      y.bias_partials = tr->bias_partials + (size_t)l_out * tr->bias_layer_stride; y.aux_mode = out_aux ? AUX_UNMASKED : AUX_NONE;
      sq.push_back(y);
      return;
    }
    RowGemmArgs g{};
    g.bias_partials = tr->bias_partials + (size_t)l_out * tr->bias_layer_stride;
    g.In = tr->dZ[l]; g.W = tr->WbT + (size_t)l * 262144; g.bias = nullptr; g.add = add; g.mask_in = mask; g.mask_out = nullptr; g.res = nullptr;
    g.out_main = out_main; g.out_aux = out_aux; g.M = n; g.N = 512; g.K = 512; g.relu = 0;
    g.aux_mode = out_aux ? AUX_UNMASKED : AUX_NONE; g.st = st; g.dbg = 0;
    g.absmax = tr->f16 ? tr->st->dz_absmax_slots : nullptr;
    launch_rowgemm(g, s, tr->f16);
    ++tr->prof_launches;
  };
  ProfScope* dchain = new ProfScope(tr, s, KC_GEMM_DGRAD);
  dgrad(f2, f1, nullptr, tr->dZ[f1], nullptr);
  int cur = 0;
  dgrad(f1, 3 * tr->nb + 2, nullptr, tr->dZ[3 * tr->nb + 2], tr->dR[cur]);
  for (int b = tr->nb; b >= 0; --b) {
    dgrad(3 * b + 2, 3 * b + 1, nullptr, tr->dZ[3 * b + 1], nullptr);
    dgrad(3 * b + 1, 3 * b, nullptr, tr->dZ[3 * b], nullptr);
    if (b > 0) {
      dgrad(3 * b, 3 * (b - 1) + 2, tr->dR[cur], tr->dZ[3 * (b - 1) + 2], tr->dR[cur ^ 1]);
      cur ^= 1;
    }
  }
  if (seq) launch_rowseq<true>(tr, sq, n, st, s);

  delete dchain;
  // the partial buffers of this step (reduced by grad_reduce_kernel in the split flow, by the optimiser launches in the fused step)
  {
    GradReduceArgs a{};
    a.slabs = tr->slabs; a.slab_stride = tr->n_wide; a.nslabs = tr->nslabs; a.fc3_partials = tr->fc3_partials;
    a.fc3_stride = tr->fc3_stride; a.stat_partials = tr->stat_partials; a.n_loss_blocks = nblk; a.grad = tr->pb.d_grad;
    a.n_wide = tr->n_wide; a.n_params = tr->n_params; a.st = st;
    a.bias_partials = tr->bias_partials; a.bias_layer_stride = tr->bias_layer_stride; a.n_layers = tr->L;
    a.skip_wide = fused ? 1 : 0;
    a.fault = tr->seq_err;
    // partial rows per layer: one per loss workgroup for fc2, one per 80-row tile from the input-gradient GEMMs
    for (int l = 0; l < tr->L; ++l) a.bias_count[l] = (l == f2) ? nblk : (n + 79) / 80;
    tr->last_reduce = a;   // the fused update reduces the partials itself
  }
  // weight gradients of all wide layers in one launch
  tr->wide_done = false;
  {
    WgradArgs a{};
    fill_wgrad_args(tr, a, n, st);
    if (const char* e = ACEZ_DIAG_ENV("ACEZ_WGO_DBG")) a.dbg = atoi(e);   // timing experiments (wgrad_opt_kernel's ablation bits)
    ProfScope ps(tr, s, KC_WGRAD);
    const int groups = tr->L * tr->nslabs;
    if (fused && wgrad_opt_usable(tr)) {
      // single-GPU fused step: the launch applies the optimiser to the wide layers' weights itself (wgrad_opt_kernel); the update that
      // follows (train_update_impl) only has the small parameters, the statistics and the schedule left
      WgradOptArgs o{};
      fill_adam_args(tr, o.ad);
      o.ad.tail = tr->last_reduce;
      o.xch = tr->wg_xch; o.flags = tr->wg_flags; o.spin_limit = tr->seq_spin_limit;
      o.trace = tr->wgo_trace;
      o.target = 2u * ++tr->wg_epoch;
      o.epoch = tr->wg_epoch; o.status = tr->wg_status; o.rec = tr->wg_rec;
      if ((long)tr->wg_epoch - 1 == tr->wgo_fault_at) {   // tests: a partner that never arrives (in every workgroup, or in every fault_mod-th)
        if (tr->wgo_fault_mod > 0) o.fault_mod = tr->wgo_fault_mod;
        else o.target += 1u << 20;
      }
      // the small parameters ride in the multiplier waves of the first workgroups, the schedule wave that closes the step in the last one
      o.nsmall = small_cols_blocks(tr->L, (int64_t)tr->no * 513, 8);
      o.do_post = 1;
      if (const char* e = ACEZ_DIAG_ENV("ACEZ_WGO_POST")) o.do_post = o.do_post && atoi(e) != 0;   // timing experiments: the schedule wave as its own launch
      const dim3 grid(256 * ((tr->L + 7) / 8));
      if ((int)grid.x < o.nsmall) abort();
      const PostArgs post = post_args(tr);
      if (tr->f16) hipLaunchKernelGGL(wgrad_opt_kernel<EltF16>, grid, dim3(WGRAD_THREADS), 0, s, a, o, post);
      else hipLaunchKernelGGL(wgrad_opt_kernel<EltBf16>, grid, dim3(WGRAD_THREADS), 0, s, a, o, post);
      tr->wide_done = true;
          tr->post_done = o.do_post != 0;
    }
    else if (tr->f16) hipLaunchKernelGGL(wgrad_kernel<EltF16>, dim3(128 * ((groups + 7) / 8)), dim3(WGRAD_THREADS), 0, s, a);
    else hipLaunchKernelGGL(wgrad_kernel<EltBf16>, dim3(128 * ((groups + 7) / 8)), dim3(WGRAD_THREADS), 0, s, a);
  }
  if (!fused) {
    const int64_t wide_blocks = (tr->n_wide / 4 + 255) / 256;
    const int64_t tail_blocks = grad_reduce_tail_blocks(tr->L, tr->n_params - tr->n_wide);
    ProfScope ps(tr, s, KC_REDUCE);
    if (pf) {
      // split flow with the pose network folded in (a data-parallel host all-reduces d_grad next, so the pose gradients must be complete
      // now): the reduce + backward chain (S1) rides at the front of the gradient-reduction launch, the weight gradients (S2) follow
      const PoseNetArgs a = pose_net_args(tr, &tr->st->active);
      const int T = tr->pose_tile, np = (tr->buf.n_images + T - 1) / T;
#define ACEZ_GRP(TT) hipLaunchKernelGGL(grad_reduce_pose_kernel<TT>, dim3((unsigned)(np + wide_blocks + tail_blocks)), dim3(256), 0, s, tr->last_reduce, a, \
                                        (const float*)tr->row_dT, (const int*)tr->row_image, n, np)
      if (T == 16) ACEZ_GRP(16); else if (T == 4) ACEZ_GRP(4); else ACEZ_GRP(8);
#undef ACEZ_GRP
    } else {
      hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)(wide_blocks + tail_blocks)), dim3(256), 0, s, tr->last_reduce);
    }
  }
  if (ps != s) ACEZ_HIP_CHECK(hipStreamWaitEvent(s, tr->ev_pose_bwd, 0));   // d_grad's pose tail: read by the all-reduce and by the pose AdamW
  if (pf && !fused) launch_pose_wgrad(tr, &tr->st->active, false, s);
  ACEZ_HIP_CHECK(hipGetLastError());
  return ACEZ_OK;
}

extern "C" int acez_train_backward(acez_trainer* tr, const int64_t* d_indices, int n, void* stream) {
  return train_backward_impl(tr, d_indices, n, stream, false);
}

static int train_update_impl(acez_trainer* tr, void* stream, bool fused, int layer_lo = 0, int layer_hi = -1, const int64_t* d_next = nullptr,
                             int n_next = 0) {
  ACEZ_REQUIRE(tr, "null trainer");
  ACEZ_REQUIRE(!tr->inference_only, "an inference-only context (acez_train_config.inference_only) cannot train");
  if (layer_hi < 0) layer_hi = tr->L;
  ACEZ_REQUIRE(layer_lo >= 0 && layer_lo <= layer_hi && layer_hi <= tr->L, "layer range out of bounds");
  ACEZ_REQUIRE(!fused || (layer_lo == 0 && layer_hi == tr->L), "the fused step updates every layer");
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  hipStream_t s = (hipStream_t)stream;
  // two updates without a backward in between (a data-parallel rank whose shard holds no row of a batch zeroes its gradient and
  // only takes part in the all-reduce): the schedule bookkeeping of the previous step must not be lost
  flush_post(tr, s);
  AdamArgs a;
  fill_adam_args(tr, a);
  if (fused) { a.slabs = tr->slabs; a.nslabs = tr->nslabs; a.slab_stride = tr->n_wide; a.tail = tr->last_reduce; }
  const int nsmall = adamw_small_blocks(tr->L, (int64_t)tr->no * 513, fused);   // small-parameter workgroups come first in the grid
  // wgrad_opt_kernel of this step's backward has already updated the wide layers' weights and the small parameters: no optimiser
  // workgroups for the head are left
  const bool wide_done = fused && tr->wide_done;
  tr->wide_done = false;
  if (wide_done) {
    // ... and it has closed the step; the next batch (acez_train_step_next) was gathered beside the loss kernel into the other buffers
    if (tr->post_done) { st_flip(tr); tr->post_pending = false; } else { tr->post_pending = true; }
    tr->post_done = false;
    if (tr->next_gathered) {
      std::swap(tr->R[0], tr->R0_alt);
      std::swap(tr->batch_meta, tr->batch_meta_alt);
      tr->pre_idx = tr->next_idx; tr->pre_n = tr->next_n;
      tr->next_gathered = false;
    }
    return ACEZ_OK;
  }
  const bool pf = tr->pose_fused;
  if (pf && fused) {
    // the head's AdamW with the pose network's reduce + backward chain (S1) as the first workgroups of the same launch, then the
    // pose weight gradients with AdamW in their epilogue (S2: needs S1 of every image tile)
    const int T = tr->pose_tile, np = (tr->buf.n_images + T - 1) / T;
    { ProfScope ps(tr, s, KC_ADAMW);
#define ACEZ_AP(TT) hipLaunchKernelGGL(adamw_pose_kernel<TT>, dim3(np + tr->L * 64 + nsmall), dim3(pose_threads<TT>()), 0, s, a, pose_net_args(tr, &tr->st->active, 1), \
                                       (const float*)tr->row_dT, (const int*)tr->row_image, tr->last_n, np)
      if (T == 16) ACEZ_AP(16); else if (T == 4) ACEZ_AP(4); else ACEZ_AP(8);
#undef ACEZ_AP
    }
    launch_pose_wgrad(tr, &tr->st->active, true, s);
    tr->post_pending = true;
    if (tr->next_gathered) {   // the next batch was gathered beside this step's loss kernel into the other input buffer (this step's wgrad read the current one)
      std::swap(tr->R[0], tr->R0_alt);
      std::swap(tr->batch_meta, tr->batch_meta_alt);
      tr->pre_idx = tr->next_idx; tr->pre_n = tr->next_n;
      tr->next_gathered = false;
    }
    ACEZ_HIP_CHECK(hipGetLastError());
    return ACEZ_OK;
  }
  a.layer_lo = layer_lo; a.layer_hi = layer_hi;
  if (d_next && n_next > 0 && tr->cfg.pose_refinement == 0 && tr->have_buf && layer_lo == 0 && layer_hi == tr->L) {
    // the next batch is known: its gather and this step's schedule bookkeeping ride in the optimiser's launch (adamw_next_kernel).
    // fused: the optimiser sums the weight-gradient slabs itself; else (acez_train_update_next: a data-parallel rank) it reads the
    // all-reduced bucket like adamw_kernel, and the schedule wave takes the statistics from the bucket too (a.tail is null)
    int n_adam = tr->L * 64 + nsmall;
    // timing experiments (diagnostics build; results wrong by construction): 1 = no optimiser workgroups at all, 2 = no gather
    const int tail_abl = ACEZ_DIAG_ENV("ACEZ_TAIL_ABL") ? atoi(ACEZ_DIAG_ENV("ACEZ_TAIL_ABL")) : 0;
    if (tail_abl & 1) n_adam = 0;
    if (tail_abl & 2) n_next = 0;
    // every workgroup of the launch resident at once (4 of these 256-thread workgroups per CU): gather workgroups that had to wait for a
    // free slot started when the optimiser's tiles were done and ran their three load levels as the launch's tail. ACEZ_NEXT_GBLOCKS
    // overrides the count (timing experiments).
    int gcap = 4 * tr->n_cus - n_adam - 1;
    if (gcap < 64) gcap = 64;
    if (const char* e = ACEZ_DIAG_ENV("ACEZ_NEXT_GBLOCKS")) gcap = std::max(1, atoi(e));
    const int gwant = (n_next + 3) / 4 < 1024 ? (n_next + 3) / 4 : 1024;
    const int gblocks = gwant < gcap ? gwant : gcap;
    { ProfScope ps(tr, s, KC_ADAMW);
      // (+ 1: the launch's last workgroup is the schedule wave, as in step_begin_kernel -- gblocks gather workgroups remain)
      hipLaunchKernelGGL(adamw_next_kernel, dim3(n_adam + gblocks + 1), dim3(256), 0, s, a, n_adam, (const uint16_t*)tr->buf.d_features, d_next, tr->R[0], n_next,
                         post_args(tr), gather_meta(tr)); }
    st_flip(tr);
    tr->pre_idx = d_next; tr->pre_n = n_next;
    tr->post_pending = false;
    ACEZ_HIP_CHECK(hipGetLastError());
    return ACEZ_OK;
  }
  if (tr->cfg.pose_refinement != 0) {
    // the pose parameters' AdamW (and, for the network, the refresh of its four transposed copies) at the front of the head's optimiser launch
    float* wt = (tr->cfg.pose_refinement == 2 && tr->pose_fused) ? tr->pose_wt : nullptr;
    const int npb = (int)((tr->pb.n_pose_params + 255) / 256);
    const int n_adam = (layer_hi - layer_lo) * 64 + nsmall;
    // the next batch announced (acez_train_update_next) and the pose network folded into the step's launches: its rows are gathered here,
    // the next backward's first launch keeps the pose forward and the schedule wave (step_begin_pose_kernel without gather workgroups)
    const bool ahead = d_next && n_next > 0 && tr->pose_fused && tr->have_buf && layer_lo == 0 && layer_hi == tr->L;
    int gblocks = 0;
    if (ahead) {
      const int gwant = (n_next + 3) / 4 < 1024 ? (n_next + 3) / 4 : 1024;
      gblocks = std::max(64, std::min(gwant, 4 * tr->n_cus - n_adam - npb));
    }
    { ProfScope ps(tr, s, KC_ADAMW);
      hipLaunchKernelGGL(adamw_split_pose_kernel, dim3(npb + n_adam + gblocks), dim3(256), 0, s, a, npb, tr->pb.d_pose_params, tr->pb.d_pose_m,
                         tr->pb.d_pose_v, (const float*)(tr->pb.d_grad + tr->n_params + 4), (int64_t)tr->pb.n_pose_params,
                         (const AdamScalars*)&tr->st->pose_adam, (const int*)&tr->st->pose_enable, (const int*)&tr->st->active, (const int*)tr->seq_err, wt,
                         n_adam, (const uint16_t*)tr->buf.d_features, d_next, tr->R[0], ahead ? n_next : 0, gather_meta(tr)); }
    if (ahead) { tr->pre_idx = d_next; tr->pre_n = n_next; }
    if (!wt) tr->pose_wt_valid = false;   // no transposed copies kept here: the next folded forward rebuilds them
  } else {
    ProfScope ps(tr, s, KC_ADAMW);
    hipLaunchKernelGGL(adamw_kernel, dim3((layer_hi - layer_lo) * 64 + nsmall), dim3(256), 0, s, a);
  }
  tr->post_pending = true;   // sched_post: with the next step's gather, or at the next state read-out (flush_post)
  ACEZ_HIP_CHECK(hipGetLastError());
  return ACEZ_OK;
}

extern "C" int acez_train_update(acez_trainer* tr, void* stream) { return train_update_impl(tr, stream, false); }
extern "C" int acez_train_update_next(acez_trainer* tr, const int64_t* d_indices_next, int n_next, void* stream) {
  ACEZ_REQUIRE(n_next >= 0 && (!tr || n_next <= tr->max_batch), "n_next must be in [0, max_batch]");
  return train_update_impl(tr, stream, false, 0, -1, d_indices_next, n_next);
}

// Sharded data-parallel update (ZeRO-1 by layer): this rank applies AdamW to the weight matrices of wide layers [layer_lo, layer_hi)
// only -- d_grad must hold their reduced gradients -- and to ALL small parameters (biases, fc3; their gradients are all-reduced and
// the update is replicated), then closes the step's schedule bookkeeping like acez_train_update. The fp32 masters / AdamW moments of
// the other layers' weights go stale on this rank until the owners' values are copied in (host: HeadTrainer.gather_masters).
extern "C" int acez_train_update_layers(acez_trainer* tr, int layer_lo, int layer_hi, void* stream) {
  return train_update_impl(tr, stream, false, layer_lo, layer_hi);
}

// 16-bit compute copies W[out][in] of wide layers [layer_lo, layer_hi), 512 x 512 each, to / from a caller buffer (the all-gather
// of the sharded update). Import also rebuilds the transposed copies of those layers.
extern "C" int acez_trainer_export_weights16(acez_trainer* tr, int layer_lo, int layer_hi, void* d_dst, void* stream) {
  ACEZ_REQUIRE(tr && d_dst, "null pointer");
  ACEZ_REQUIRE(layer_lo >= 0 && layer_lo <= layer_hi && layer_hi <= tr->L, "layer range out of bounds");
  if (layer_hi == layer_lo) return ACEZ_OK;
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  ACEZ_HIP_CHECK(hipMemcpyAsync(d_dst, tr->Wb + (size_t)layer_lo * 262144, (size_t)(layer_hi - layer_lo) * 262144 * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return ACEZ_OK;
}
extern "C" int acez_trainer_import_weights16(acez_trainer* tr, int layer_lo, int layer_hi, const void* d_src, void* stream) {
  ACEZ_REQUIRE(tr && d_src, "null pointer");
  ACEZ_REQUIRE(layer_lo >= 0 && layer_lo <= layer_hi && layer_hi <= tr->L, "layer range out of bounds");
  if (layer_hi == layer_lo) return ACEZ_OK;
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  ACEZ_HIP_CHECK(hipMemcpyAsync(tr->Wb + (size_t)layer_lo * 262144, d_src, (size_t)(layer_hi - layer_lo) * 262144 * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  hipLaunchKernelGGL(transpose16_kernel, dim3((layer_hi - layer_lo) * 64), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)tr->Wb, tr->WbT, layer_lo);
  ACEZ_HIP_CHECK(hipGetLastError());
  return ACEZ_OK;
}

extern "C" int acez_trainer_import_weights16_all(acez_trainer* tr, int own_lo, int own_hi, const void* d_src_all, void* stream) {
  ACEZ_REQUIRE(tr && d_src_all, "null pointer");
  ACEZ_REQUIRE(own_lo >= 0 && own_lo <= own_hi && own_hi <= tr->L, "layer range out of bounds");
  if (own_hi - own_lo == tr->L) return ACEZ_OK;
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  hipLaunchKernelGGL(import16_kernel, dim3(tr->L * 64), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)d_src_all, tr->Wb, tr->WbT, own_lo, own_hi);
  ACEZ_HIP_CHECK(hipGetLastError());
  return ACEZ_OK;
}

// Single-GPU step: backward + update with the wide-layer gradients handed from the weight-gradient slabs straight to the
// optimiser (no flat-gradient round trip through HBM). Bitwise the same parameters as acez_train_backward + acez_train_update;
// afterwards d_grad holds the bias / fc3 gradients and the statistics, its wide-layer weight part is NOT written.
extern "C" int acez_train_step(acez_trainer* tr, const int64_t* d_indices, int n, void* stream) {
  int rc = train_backward_impl(tr, d_indices, n, stream, true);
  if (rc != ACEZ_OK) return rc;
  return train_update_impl(tr, stream, true);
}

// acez_train_step with the NEXT step's batch announced: when the following call is acez_train_step / acez_train_step_next with exactly
// these indices (same device pointer, same count), its gather has already happened -- inside this step's optimiser launch, together
// with this step's schedule bookkeeping -- and that call starts with the forward GEMM chain. Any other next call (other indices, a
// split backward, a state read) is still correct: it simply gathers again. d_indices_next may be NULL (= acez_train_step).
extern "C" int acez_train_step_next(acez_trainer* tr, const int64_t* d_indices, int n, const int64_t* d_indices_next, int n_next, void* stream) {
  ACEZ_REQUIRE(n_next >= 0 && n_next <= (tr ? tr->max_batch : 0), "n_next must be in [0, max_batch]");
  int rc = train_backward_impl(tr, d_indices, n, stream, true, d_indices_next, n_next);
  if (rc != ACEZ_OK) return rc;
  return train_update_impl(tr, stream, true, 0, -1, d_indices_next, n_next);
}

extern "C" int acez_trainer_get_state(acez_trainer* tr, acez_train_state* h_out, void* stream) {
  ACEZ_REQUIRE(tr && h_out, "null pointer");
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  flush_post(tr, (hipStream_t)stream);
  // a hand-off poll of the one-launch chains expired since the last read: the steps since then were no-ops on the device; from
  // here on this trainer uses per-layer launches (acez_trainer_seq_status reports it)
  seq_fault_check(tr, (hipStream_t)stream);
  TrainState hs;
  ACEZ_HIP_CHECK(hipMemcpyAsync(&hs, tr->st, offsetof(TrainState, dz_absmax_slots), hipMemcpyDeviceToHost, (hipStream_t)stream));
  ACEZ_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  h_out->iteration = hs.iteration; h_out->max_iterations = hs.max_iterations; h_out->in_cooldown = hs.in_cooldown;
  h_out->nan_flag = hs.nan_flag; h_out->lr = hs.lr; h_out->last_loss = hs.last_loss;
  h_out->last_batch_inliers = hs.last_inliers; h_out->focal_scale = 1.0 + hs.calib_g;
  h_out->grad_scale = hs.grad_scale; h_out->opt_steps = hs.opt_steps;
  return hs.nan_flag ? ACEZ_ERR_NAN : ACEZ_OK;
}

extern "C" int acez_trainer_get_log(acez_trainer* tr, int first, int count, float* h_loss, float* h_inliers, void* stream) {
  ACEZ_REQUIRE(tr, "null trainer");
  ACEZ_REQUIRE(first >= 0 && count >= 0 && first + count <= tr->log_cap, "log range out of bounds");
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  hipStream_t s = (hipStream_t)stream;
  flush_post(tr, s);
  if (h_loss) ACEZ_HIP_CHECK(hipMemcpyAsync(h_loss, tr->log_loss + first, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, s));
  if (h_inliers) ACEZ_HIP_CHECK(hipMemcpyAsync(h_inliers, tr->log_inl + first, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, s));
  ACEZ_HIP_CHECK(hipStreamSynchronize(s));
  return ACEZ_OK;
}

extern "C" int acez_trainer_last_scene_coords(acez_trainer* tr, float* h_xyz, int n, void* stream) {
  ACEZ_REQUIRE(tr && h_xyz, "null pointer");
  ACEZ_REQUIRE(!tr->inference_only, "an inference-only context has no training-step scene coordinates");
  ACEZ_REQUIRE(n > 0 && n <= tr->max_batch, "n out of range");
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  ACEZ_HIP_CHECK(hipMemcpyAsync(h_xyz, tr->xyz, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  ACEZ_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return ACEZ_OK;
}

// Inference on many rows (whole frames): the head's layers are 1x1 convolutions over [rows][512], so the large-tile implicit-GEMM
// kernels of the encoder (256 x 256 tiles) run them; same rounding points as the training forward (round_before_add).
static uint16_t* launch_forward_conv(acez_trainer* tr, const uint16_t* in0, int n, hipStream_t s) {
  const float* P = tr->pb.d_params;
  auto layer = [&](int l, const uint16_t* in, const uint16_t* add, uint16_t* out) {
    ConvGemmArgs g{};
    g.In = in; g.W = tr->Wb + (size_t)l * 262144; g.bias = P + (int64_t)l * 262656 + 262144; g.add = add; g.out = out;
    g.zeros = tr->zeros; g.Hi = 1; g.Wi = 1; g.Ci = 512; g.ci_shift = 9; g.Ho = 1; g.Wo = 1; g.Co = 512; g.ksize = 1; g.stride = 1;
    g.pad = 0; g.K = 512; g.Kp = 512; g.M = n; g.round_before_add = 1; g.dbg = 0; g.f16 = tr->f16 ? 1 : 0;
    if (const char* e = ACEZ_DIAG_ENV("ACEZ_CONV_DBG")) g.dbg = atoi(e);   // (diagnostics build: convgemm512's ablation / stagger bits)
    static const int conv_tile = ACEZ_DIAG_ENV("ACEZ_HEAD_CONV_TILE") ? atoi(ACEZ_DIAG_ENV("ACEZ_HEAD_CONV_TILE")) : 0;   // (diagnostics build: 256 / 512 force a tile; round 5: the 256 x 128 tiles, which keep a third more input bytes in flight, are 8 % SLOWER here: 2.54 against 2.35 ms per 64 frames)
    launch_convgemm(g, true, s, conv_tile);
  };
  const uint16_t* r = in0;
  for (int b = 0; b <= tr->nb; ++b) {
    layer(3 * b, r, nullptr, tr->out[3 * b]);
    layer(3 * b + 1, tr->out[3 * b], nullptr, tr->out[3 * b + 1]);
    layer(3 * b + 2, tr->out[3 * b + 1], r, tr->R[b + 1]);     // R = relu(conv) + R   ace_network.py:126,133
    r = tr->R[b + 1];
  }
  const int f1 = 3 * (tr->nb + 1), f2 = f1 + 1;
  layer(f1, r, nullptr, tr->out[f1]);
  layer(f2, tr->out[f1], nullptr, tr->out[f2]);
  return tr->out[f2];
}

// Whole-frame passes since round 6: the wide layers as ONE launch on LDS-resident 128-row tiles (head_maps.hip). The fragment-ordered weight
// copy is rebuilt from Wb at the start of every pass (4 MiB: whoever changed the weights since -- an optimiser step, a checkpoint load, an
// imported all-gather -- is covered without a dirty flag). Returns null when the context cannot run it (allocation failure, more residual
// blocks than the argument block holds): the caller falls back to the per-layer launches.
// xyz != null: fc3 + de-homogenisation run inside the launch and the scene coordinates go to `xyz` (rows offset by row_offset, planar maps
// with planar_hw > 0); the returned pointer is then only a success token.
static uint16_t* launch_forward_maps(acez_trainer* tr, const uint16_t* in0, int n, hipStream_t s, float* xyz = nullptr, int planar_hw = 0,
                                     int row_offset = 0) {
  static const bool off = ACEZ_DIAG_ENV("ACEZ_HEAD_CHAIN") && atoi(ACEZ_DIAG_ENV("ACEZ_HEAD_CHAIN")) == 0;   // (diagnostics build: the eight launches)
  if (off || tr->nb + 1 > 7) return nullptr;
  if (!tr->Wf && dmalloc(tr, (void**)&tr->Wf, (size_t)tr->L * 262144 * sizeof(uint16_t)) != ACEZ_OK) { tr->Wf = nullptr; return nullptr; }
  const int n_frags = tr->L * 16 * 32;
  hipLaunchKernelGGL(wfrag_pack_kernel, dim3((n_frags + 3) / 4), dim3(256), 0, s, tr->Wb, tr->Wf, n_frags);
  HeadMapsArgs a{};
  a.In = in0; a.Wf = tr->Wf; a.params = tr->pb.d_params; a.n = n; a.nb = tr->nb;
  for (int b = 1; b <= tr->nb; ++b) a.R[b] = tr->R[b];
  const int f2 = 3 * (tr->nb + 1) + 1;
  a.Out = tr->out[f2];
  if (xyz) {
    LossArgs h{};
    fill_loss_head(tr, h);
    a.fc3 = 1; a.W3 = h.W3; a.b3 = h.b3; a.no = h.no; a.use_homogeneous = h.use_homogeneous;
    for (int i = 0; i < 3; ++i) a.mean[i] = h.mean[i];
    a.max_inv_scale = h.max_inv_scale; a.min_inv_scale = h.min_inv_scale; a.h_beta = h.h_beta;
    a.out_xyz = xyz; a.planar_hw = planar_hw; a.row_offset = row_offset;
  }
  const int ntiles = (n + 127) / 128, cus = tr->n_cus > 0 ? tr->n_cus : 256;
  const dim3 grid(ntiles < cus ? ntiles : cus), blk(512);
  if (tr->f16) hipLaunchKernelGGL((head_maps_kernel<EltF16>), grid, blk, 0, s, a);
  else hipLaunchKernelGGL((head_maps_kernel<EltBf16>), grid, blk, 0, s, a);
  return tr->out[f2];
}

static int head_forward_impl(acez_trainer* tr, const void* d_features, int n, float* d_out, int planar_hw, void* stream) {
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  hipStream_t s = (hipStream_t)stream;
  const uint16_t* f = (const uint16_t*)d_features;
  bool used_seq = false;
  auto pass = [&]() {
    for (int done = 0; done < n; done += tr->max_batch) {
      const int cnt = (n - done < tr->max_batch) ? n - done : tr->max_batch;
      const bool seqp = seq_usable(tr, cnt);
      const bool conv = cnt >= 256 * 128;   // large passes: the encoder's large-tile kernels (both operand formats)
      used_seq = used_seq || (seqp && !conv);
      if (conv && launch_forward_maps(tr, f + (size_t)done * 512, cnt, s, planar_hw > 0 ? d_out : d_out + (size_t)done * 3, planar_hw, planar_hw > 0 ? done : 0))
        continue;   // wide layers + fc3 + de-homogenisation in one launch
      uint16_t* act = conv ? launch_forward_conv(tr, f + (size_t)done * 512, cnt, s) : launch_forward(tr, f + (size_t)done * 512, cnt, nullptr, s);
      LossArgs a{};
      fill_loss_head(tr, a);
      a.act = act; a.n = cnt; a.idx = nullptr; a.st = nullptr;
      if (planar_hw > 0) { a.out_xyz = d_out; a.planar_hw = planar_hw; a.row_offset = done; }
      else a.out_xyz = d_out + (size_t)done * 3;
      launch_loss(tr, (cnt + 4 * tr->loss_rows - 1) / (4 * tr->loss_rows), s, a);
    }
  };
  pass();
  // A one-launch chain whose hand-off poll expired has produced garbage (and switched st_infer off): nothing else reads the fault word
  // on the inference path, so it is read here -- a call that used the chains is synchronous -- and the pass is repeated on per-layer
  // launches (seq_fault_check has disabled the chains for good).
  if (used_seq && seq_fault_check(tr, s)) pass();
  ACEZ_HIP_CHECK(hipGetLastError());
  return ACEZ_OK;
}

extern "C" int acez_head_forward(acez_trainer* tr, const void* d_features, int n, float* d_out_xyz, void* stream) {
  ACEZ_REQUIRE(tr && d_features && d_out_xyz, "null pointer");
  ACEZ_REQUIRE(n > 0, "n must be positive");
  return head_forward_impl(tr, d_features, n, d_out_xyz, 0, stream);
}

extern "C" int acez_head_forward_maps(acez_trainer* tr, const void* d_features, int n_frames, int h, int w, float* d_out_maps, void* stream) {
  ACEZ_REQUIRE(tr && d_features && d_out_maps, "null pointer");
  ACEZ_REQUIRE(n_frames > 0 && h > 0 && w > 0 && (int64_t)n_frames * h * w < ((int64_t)1 << 31), "bad frame geometry");
  return head_forward_impl(tr, d_features, n_frames * h * w, d_out_maps, h * w, stream);
}

// Per-kernel-class timing (diagnostics for bench.py's roofline leg): when enabled every launch of the following
// train_backward/update calls is bracketed by HIP events on the launch stream. acez_trainer_get_profile
// synchronises, returns the summed milliseconds and launch counts per class
// {sched, gather, gemm_fwd, loss, gemm_dgrad, wgrad, grad_reduce, adamw} and clears the record.
extern "C" int acez_trainer_set_profiling(acez_trainer* tr, int enable) {
  ACEZ_REQUIRE(tr, "null trainer");
  tr->profiling = enable != 0;
  return ACEZ_OK;
}
extern "C" int acez_trainer_get_profile(acez_trainer* tr, float* h_ms8, int32_t* h_counts8) {
  ACEZ_REQUIRE(tr && h_ms8 && h_counts8, "null pointer");
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  ACEZ_HIP_CHECK(hipDeviceSynchronize());
  for (int i = 0; i < KC_COUNT; ++i) { h_ms8[i] = 0.f; h_counts8[i] = 0; }
  for (auto& u : tr->ev_used) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, tr->ev_pool[u.i0], tr->ev_pool[u.i1]) == hipSuccess) {
      h_ms8[u.cls] += ms;
      h_counts8[u.cls] += u.launches;
    }
  }
  tr->ev_used.clear();
  tr->ev_next = 0;
  return ACEZ_OK;
}

// Diagnostics for the tests: copy one of the trainer's intermediate device buffers to the host. kind 0: post-ReLU output of wide
// layer `index` [n][512] bf16 (training: not kept for the last layer of a block, whose mask bits are); 1: dZ of layer `index` [n][512] bf16;
// 2: residual stream `index` [n][512] bf16 (0 = the gathered batch); 3: weight-gradient slab `index` [n_wide] fp32; 4: bias-gradient partial
// rows of layer `index` [max_blocks][512] fp32; 9: ReLU mask bits of layer `index` (RowGemmArgs::mask_out), 8 KiB per 80-row tile.
extern "C" int acez_trainer_debug_read(acez_trainer* tr, int kind, int index, void* h_out, int64_t bytes, void* stream) {
  ACEZ_REQUIRE(tr && h_out && bytes > 0, "null pointer");
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  const void* src = nullptr;
  int64_t cap = 0;
  const int64_t act_bytes = (int64_t)tr->max_batch * 512 * 2;
  if (kind == 0 && index >= 0 && index < tr->L) { src = tr->out[index]; cap = act_bytes; }
  else if (kind == 1 && index >= 0 && index < tr->L && tr->dZ[index]) { src = tr->dZ[index]; cap = act_bytes; }
  else if (kind == 2 && index >= 0 && index < tr->nb + 2) { src = tr->R[index]; cap = act_bytes; }
  else if (kind == 3 && index >= 0 && index < tr->nslabs && tr->slabs) { src = tr->slabs + (size_t)index * tr->n_wide; cap = tr->n_wide * 4; }
  else if (kind == 4 && index >= 0 && index < tr->L && tr->bias_partials) { src = tr->bias_partials + (size_t)index * tr->bias_layer_stride; cap = tr->bias_layer_stride * 4; }
  else if (kind == 6 && tr->seq_xcc) { src = tr->seq_xcc; cap = (8 + 256) * 4; }
  else if (kind == 9 && index >= 0 && index < tr->L && tr->maskbits) { src = tr->maskbits + (size_t)index * tr->mask_stride; cap = (int64_t)tr->mask_stride * 8; }
  else if (kind == 7 && tr->wgo_trace) { src = tr->wgo_trace; cap = 256 * 12 * 8 * 8; }
  else if (kind == 8 && tr->pose_trace) { src = tr->pose_trace; cap = 3 * 1024 * 16 * 8; }
  ACEZ_REQUIRE(src && bytes <= cap, "unknown buffer or size out of range");
  ACEZ_HIP_CHECK(hipMemcpyAsync(h_out, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  ACEZ_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return ACEZ_OK;
}

extern "C" int acez_trainer_get_poses(acez_trainer* tr, float* h_poses34, void* stream) {
  ACEZ_REQUIRE(tr && h_poses34, "null pointer");
  ACEZ_REQUIRE(tr->have_buf, "acez_trainer_set_buffer has not been called");
  ACEZ_HIP_CHECK(hipSetDevice(tr->device));
  hipStream_t s = (hipStream_t)stream;
  const int I = tr->buf.n_images;
  const float* src = tr->buf.d_image_pose_inv;
  if (tr->cfg.pose_refinement == 2 && tr->pose_fused && tr->pose_tile_fwd != 16) {
    const PoseNetArgs a = pose_net_args(tr, nullptr);
    const int T = tr->pose_tile_fwd, np = (I + T - 1) / T;
    hipLaunchKernelGGL(pose_transpose_kernel, dim3(4, 4, 4), dim3(256), 0, s, a.P, tr->pose_wt, (const int*)nullptr);
    if (T == 4) hipLaunchKernelGGL(pose_fwd_t_kernel<4>, dim3(np), dim3(pose_fwd_threads<4>()), 0, s, a);
    else hipLaunchKernelGGL(pose_fwd_t_kernel<8>, dim3(np), dim3(pose_fwd_threads<8>()), 0, s, a);
    ACEZ_HIP_CHECK(hipGetLastError());
    src = tr->pose_cur;
  } else if (tr->cfg.pose_refinement == 2) {
    pose_forward(tr, nullptr, s);
    ACEZ_HIP_CHECK(hipGetLastError());
    src = tr->pose_cur;
  } else if (tr->cfg.pose_refinement == 1) {
    hipLaunchKernelGGL(pose_compose_kernel, dim3((I + 255) / 256), dim3(256), 0, s, (const float*)tr->pa1, (const float*)tr->pb.d_pose_params, 1.0f,
                       tr->pose_cur, I, (const int*)nullptr, tr->cfg.pose_refinement_ortho);
    ACEZ_HIP_CHECK(hipGetLastError());
    src = tr->pose_cur;
  }
  ACEZ_HIP_CHECK(hipMemcpy2DAsync(h_poses34, 12 * sizeof(float), src, 16 * sizeof(float), 12 * sizeof(float), I, hipMemcpyDeviceToHost, s));
  ACEZ_HIP_CHECK(hipStreamSynchronize(s));
  return ACEZ_OK;
}
