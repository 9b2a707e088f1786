// hostmath_probe.hip -- TEST INFRASTRUCTURE: exposes the HOST instantiation of the __host__ __device__ scalar
// geometry in acezero_amd/csrc/ransac_math.h so that CPU-only tests can compare it bit-for-bit with the oracle
// before anything runs on a GPU. Built by tests/test_hostmath.py with `hipcc -ffp-contract=off` (host pass only
// is used). Not part of libacez.so.
#include <hip/hip_runtime.h>
#include "../acezero_amd/csrc/ransac_math.h"

extern "C" {
int probe_p3p(const float* obj12, const float* img8, float focal, float ppx, float ppy, double* pose6) {
  float obj[4][3], img[4][2];
  memcpy(obj, obj12, sizeof(obj));
  memcpy(img, img8, sizeof(img));
  rsm::Cam k{(double)focal, (double)focal, (double)ppx, (double)ppy};
  rsm::Pose p;
  memset(&p, 0, sizeof(p));
  const bool ok = rsm::solve_pnp_p3p(obj, img, k, &p);
  for (int i = 0; i < 3; ++i) { pose6[i] = ok ? p.r[i] : 0; pose6[3 + i] = ok ? p.t[i] : 0; }
  return ok ? 1 : 0;
}
void probe_rodrigues(const double* r3, double* R9, double* J27) { rsm::rodrigues(r3, R9, J27); }
void probe_rodrigues_inv(const double* R9, double* r3) { rsm::rodrigues_inv(R9, r3); }
void probe_project(const double* pose6, float focal, float ppx, float ppy, const float* xyz, int n, double* uv, double* J12n) {
  rsm::Cam k{(double)focal, (double)focal, (double)ppx, (double)ppy};
  double R[9], dRdr[27];
  rsm::rodrigues(pose6, R, dRdr);
  for (int i = 0; i < n; ++i)
    rsm::project(R, pose6 + 3, k, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &uv[2 * i], &uv[2 * i + 1], J12n ? dRdr : nullptr,
                 J12n ? J12n + 12 * i : nullptr, J12n ? J12n + 12 * i + 6 : nullptr);
}
void probe_solve_sym6(const double* A36, const double* b6, double* x6) { rsm::solve_sym6(A36, b6, x6); }
void probe_solve_normal6(const double* A36, const double* b6, double* x6) { rsm::solve_normal6(A36, b6, x6); }
void probe_det_math(const double* x, int n, double* s, double* c, double* ac, double* ex, double* cb) {
  for (int i = 0; i < n; ++i) {
    detm::sincos(x[i], &s[i], &c[i]);
    ac[i] = detm::acos_(x[i] > 1 ? 1 : (x[i] < -1 ? -1 : x[i]));
    ex[i] = detm::exp_(x[i]);
    cb[i] = detm::cbrt_(x[i]);
  }
}
int probe_irand(uint64_t seed, uint64_t frame, uint32_t hyp, uint32_t tr, uint32_t draw, int n) {
  return rsm::irand(rsm::try_key(seed, frame, hyp, tr), draw, n);
}
}
