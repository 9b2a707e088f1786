// ablate_rowgemm.hip -- on-GPU ablation of rowgemm_kernel (tools; not part of the library).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I acezero_amd/csrc tools/ablate_rowgemm.hip -o /tmp/ablate && /tmp/ablate
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
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct V { const char* name; int dbg; bool m, a, x; } vars[] = {
      {"full fwd (bias+relu)", 0, false, false, false}, {"full dgrad (mask)", 0, true, false, false},
      {"full dgrad (mask+add+aux)", 0, true, true, true}, {"no epilogue", 1, false, false, false},
      {"no MFMA/ds_read", 2, false, false, false}, {"no loads", 4, false, false, false}, {"no loads, no epilogue", 5, false, false, false},
      {"nothing (launch only)", 7, false, false, false}};
  for (int rep = 0; rep < 2; ++rep)
    for (auto& v : vars) {
      RowGemmArgs g{};
      g.In = In; g.W = W; g.bias = v.m ? nullptr : bias; g.add = v.a ? add : nullptr; g.mask = v.m ? mask : nullptr; g.res = nullptr;
      g.out_main = out; g.out_aux = v.x ? aux : nullptr; g.M = M; g.N = 512; g.K = 512; g.relu = v.m ? 0 : 1;
      g.aux_mode = v.x ? AUX_UNMASKED : AUX_NONE; g.st = nullptr; g.dbg = v.dbg; g.bias_partials = nullptr;
      for (int i = 0; i < 20; ++i) launch_rowgemm(g, dim3(160), 0);
      CK(hipEventRecord(e0, 0));
      const int n = 200;
      for (int i = 0; i < n; ++i) launch_rowgemm(g, dim3(160), 0);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-28s %7.2f us/launch  (%.0f TFLOP/s equiv)\n", v.name, ms * 1e3 / n, 2.0 * M * 512 * 512 / (ms * 1e-3 / n) / 1e12);
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
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(wgrad_kernel, dim3(256), dim3(512), 0, 0, a);
        CK(hipEventRecord(e0, 0));
        const int n = 50;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(wgrad_kernel, dim3(256), dim3(512), 0, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("%-28s %7.2f us/launch  (%.0f TFLOP/s equiv)\n", v.name, ms * 1e3 / n, 2.0 * L * M * 512 * 512 / (ms * 1e-3 / n) / 1e12);
      }
  }
  return 0;
}
