// ablate_rowgemm.hip -- on-GPU ablation of rowgemm_kernel (tools; not part of the library).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I acezero_amd/csrc tools/ablate_rowgemm.hip -o /tmp/ablate && /tmp/ablate
#define ACEZ_DIAG 1   // the ablation bits (RowGemmArgs::dbg, WgradArgs::dbg, LossArgs::dbg) exist in the diagnostics build only
#include "head_kernels.hip"
#include <cstdio>
#include <vector>
using namespace acez;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const int M = 5120;
  uint16_t *In, *W, *out, *aux, *mask, *add; float* bias;
  CK(hipMalloc(&In, (size_t)M * 512 * 2)); CK(hipMalloc(&W, 512 * 512 * 2)); CK(hipMalloc(&out, (size_t)M * 512 * 2));
  CK(hipMalloc(&aux, (size_t)M * 512 * 2)); CK(hipMalloc(&mask, (size_t)M * 512 * 2)); CK(hipMalloc(&add, (size_t)M * 512 * 2));
  CK(hipMalloc(&bias, 512 * 4));
  std::vector<uint16_t> h((size_t)M * 512);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (uint16_t)((i * 2654435761u) >> 22) % 512;  // ~0.01 .. bf16 values
  CK(hipMemcpy(In, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h.data(), 512 * 512 * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(mask, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(add, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, 512 * 4));
  float* bpart; CK(hipMalloc(&bpart, 64 * 2 * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct V { const char* name; int dbg; bool m, a, x; } vars[] = {
      {"full fwd (bias+relu)", 0, false, false, false}, {"full dgrad (mask)", 0, true, false, false},
      {"full dgrad (mask+add+aux)", 0, true, true, true}, {"no epilogue", 1, false, false, false},
      {"no MFMA/ds_read", 2, false, false, false}, {"no loads", 4, false, false, false}, {"no loads, no epilogue", 5, false, false, false},
      {"nothing (launch only)", 7, false, false, false}};
  for (int tile : {80})
  for (int rep = 0; rep < 2; ++rep)
    for (auto& v : vars) {
      RowGemmArgs g{};
      g.In = In; g.W = W; g.bias = v.m ? nullptr : bias; g.add = v.a ? add : nullptr; g.mask_in = v.m ? reinterpret_cast<const uint2*>(mask) : nullptr; g.mask_out = v.m ? nullptr : reinterpret_cast<uint2*>(aux);   // (mask bits: any 512 KiB of device memory) g.res = nullptr;
      g.out_main = out; g.out_aux = v.x ? aux : nullptr; g.M = M; g.N = 512; g.K = 512; g.relu = v.m ? 0 : 1;
      g.aux_mode = v.x ? AUX_UNMASKED : AUX_NONE; g.st = nullptr; g.dbg = v.dbg; g.bias_partials = v.m ? bpart : nullptr;
      for (int i = 0; i < 20; ++i) launch_rowgemm(g, 0);
      CK(hipEventRecord(e0, 0));
      const int n = 200;
      for (int i = 0; i < n; ++i) launch_rowgemm(g, 0);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("tile %3d %-28s %7.2f us/launch  (%.0f TFLOP/s equiv)\n", tile, v.name, ms * 1e3 / n, 2.0 * M * 512 * 512 / (ms * 1e-3 / n) / 1e12);
    }
  // ---- wgrad
  {
    const int L = 8;
    float* slabs; uint16_t* zeros;
    const int64_t n_wide = (int64_t)L * 262656;
    CK(hipMalloc(&slabs, 2 * n_wide * 4)); CK(hipMalloc(&zeros, 1024)); CK(hipMemset(zeros, 0, 1024));
    std::vector<uint16_t*> bufs;
    for (int i = 0; i < 2 * L; ++i) { uint16_t* b; CK(hipMalloc(&b, (size_t)M * 512 * 2)); CK(hipMemcpy(b, h.data(), h.size() * 2, hipMemcpyHostToDevice)); bufs.push_back(b); }
    struct V2 { const char* name; int dbg; } v2[] = {{"wgrad full", 0}, {"wgrad no stores", 1}, {"wgrad no MFMA/tr-read", 2}, {"wgrad no loads", 4},
                                                    {"wgrad no loads no stores", 5}, {"wgrad nothing", 7}};
    for (int rep = 0; rep < 2; ++rep)
      for (auto& v : v2) {
        WgradArgs a{};
        for (int l = 0; l < L; ++l) { a.dZ[l] = bufs[2 * l]; a.In[l] = bufs[2 * l + 1]; a.w_off[l] = (int64_t)l * 262656; a.b_off[l] = a.w_off[l] + 262144; }
        a.slabs = slabs; a.slab_stride = n_wide; a.M = M; a.nslabs = 2; a.n_layers = L; a.st = nullptr; a.zeros = zeros; a.dbg = v.dbg;
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(wgrad_kernel, dim3(256), dim3(WGRAD_THREADS), 0, 0, a);
        CK(hipEventRecord(e0, 0));
        const int n = 50;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(wgrad_kernel, dim3(256), dim3(WGRAD_THREADS), 0, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("%-28s %7.2f us/launch  (%.0f TFLOP/s equiv)\n", v.name, ms * 1e3 / n, 2.0 * L * M * 512 * 512 / (ms * 1e-3 / n) / 1e12);
      }
  }
  // ---- loss kernel phases
  {
    const int n = M, no = 4, nblk = (n + 4 * LOSS_ROWS - 1) / (4 * LOSS_ROWS);
    int64_t* idx; float *tpx, *aug, *Kk, *Ki, *pose, *b3, *fc3p, *statp, *biasp, *xyz; int *vidx, *vimg; uint16_t* W3; TrainState* st;
    CK(hipMalloc(&idx, n * 8)); CK(hipMalloc(&tpx, n * 8)); CK(hipMalloc(&vidx, n * 4)); CK(hipMalloc(&aug, 2000 * 48)); CK(hipMalloc(&Kk, 2000 * 36));
    CK(hipMalloc(&Ki, 2000 * 36)); CK(hipMalloc(&vimg, 2000 * 4)); CK(hipMalloc(&pose, 1000 * 64)); CK(hipMalloc(&b3, 16)); CK(hipMalloc(&W3, 4 * 512 * 2));
    CK(hipMalloc(&fc3p, (size_t)nblk * 2052 * 4)); CK(hipMalloc(&statp, nblk * 16)); CK(hipMalloc(&biasp, (size_t)nblk * 512 * 4)); CK(hipMalloc(&xyz, n * 12));
    CK(hipMalloc(&st, sizeof(TrainState)));
    std::vector<int64_t> hi(n); for (int i = 0; i < n; ++i) hi[i] = i; CK(hipMemcpy(idx, hi.data(), n * 8, hipMemcpyHostToDevice));
    std::vector<int> hv(n); for (int i = 0; i < n; ++i) hv[i] = (i * 7) % 2000; CK(hipMemcpy(vidx, hv.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<int> hm(2000); for (int i = 0; i < 2000; ++i) hm[i] = i / 2; CK(hipMemcpy(vimg, hm.data(), 8000, hipMemcpyHostToDevice));
    std::vector<float> ones(2000 * 16, 0.5f); CK(hipMemcpy(aug, ones.data(), 2000 * 48, hipMemcpyHostToDevice)); CK(hipMemcpy(Kk, ones.data(), 2000 * 36, hipMemcpyHostToDevice));
    CK(hipMemcpy(Ki, ones.data(), 2000 * 36, hipMemcpyHostToDevice)); CK(hipMemcpy(pose, ones.data(), 1000 * 64, hipMemcpyHostToDevice));
    CK(hipMemset(tpx, 0, n * 8)); CK(hipMemset(b3, 0, 16)); CK(hipMemcpy(W3, h.data(), 4 * 512 * 2, hipMemcpyHostToDevice));
    TrainState hs{}; hs.active = 1; hs.loss_weight = 50.f; CK(hipMemcpy(st, &hs, sizeof(hs), hipMemcpyHostToDevice));
    const char* names[] = {"loss full", "loss phase A only", "loss phases A+B"};
    for (int rep = 0; rep < 2; ++rep)
      for (int dbg = 0; dbg < 3; ++dbg) {
        LossArgs a{};
        a.act = In; a.W3 = W3; a.b3 = b3; a.n = n; a.no = no; a.use_homogeneous = 1; a.max_inv_scale = 0.25f; a.min_inv_scale = 100.f; a.h_beta = 0.924f;
        a.idx = idx; a.target_px = tpx; a.target_crds = nullptr; a.view_idx = vidx; a.view_aug_inv = aug; a.view_K = Kk; a.view_Kinv = Ki; a.view_image = vimg;
        a.image_pose_inv = pose; a.row_dT = nullptr; a.row_image = nullptr; a.loss_type = 0; a.refine_calibration = 0; a.hard_clamp = 1000.f; a.depth_min = 0.1f;
        a.depth_max = 1000.f; a.depth_target = 10.f; a.inlier_px = 10.f; a.inv_batch = 1.f / n; a.focal_init = 525.f; a.st = st; a.out_xyz = xyz; a.dZ = out;
        a.fc3_partials = fc3p; a.fc3_stride = 2052; a.stat_partials = statp; a.bias_partials = biasp; a.dbg = dbg == 0 ? 0 : dbg;
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(loss_kernel, dim3(nblk), dim3(256), 0, 0, a);
        CK(hipEventRecord(e0, 0));
        const int nn = 100;
        for (int i = 0; i < nn; ++i) hipLaunchKernelGGL(loss_kernel, dim3(nblk), dim3(256), 0, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("%-28s %7.2f us/launch\n", names[dbg], ms * 1e3 / nn);
      }
  }
  return 0;
}
