// dsac_oracle.cpp -- CPU restatement of the reference's DSAC* RGB registration.  TEST INFRASTRUCTURE ONLY:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this file.
// The product (acezero_amd/) never links or imports it.
//
// PARITY UNPINNED.  The reference's dsacstar extension cannot be built here: it needs OpenCV 4.4.0
// (environment.yml:115,154; dsacstar/setup.py:34-35), which is not vendored in /root/reference, is absent from
// this image and cannot be installed (no network).  The reference has no tests or golden vectors for this path
// (SURVEY.md section 4).  What follows therefore restates
//   (a) the reference's own control flow and constants, cited per function as dsacstar/<file>:<lines>, and
//   (b) the PUBLISHED algorithm of the five OpenCV calls it makes, marked [upstream]: cv::solvePnP
//       (SOLVEPNP_P3P: calib3d/src/solvepnp.cpp solveP3P + p3p.cpp + polynom_solver.cpp; SOLVEPNP_ITERATIVE with
//       extrinsic guess: calibration.cpp cvFindExtrinsicCameraParams2 + compat_ptsetreg.cpp CvLevMarq),
//       cv::projectPoints (cvProjectPoints2Internal), cv::Rodrigues (cvRodrigues2), cv::Mat::inv (LU), cv::norm.
// It is anchored by tests/test_dsac_oracle.py on known-answer cases (exact correspondences -> ground-truth pose,
// P3P on hand-built configurations incl. duplicates, Rodrigues round trips, LM convergence).
//
// Deliberate, documented differences from the reference binary (DESIGN.md "R path: deviations"):
//   D1  random numbers: counter-based stream keyed by (seed, frame, hypothesis, try, draw) instead of the
//       per-OpenMP-thread mt19937 whose consumption depends on thread count and call history
//       (thread_rand.cpp:13-42; SURVEY.md section 8c "RNG parity caveat").
//   D2  reductions over pixels (soft inlier score; J^T J, J^T e, |e|^2 of the LM solver) use a fixed
//       interleaved-partials + pairwise-tree order (64-way resp. 256-way) instead of a sequential left fold, so
//       that a wavefront / workgroup can reproduce them bit-for-bit. The LM sums run over the inlier LIST in the
//       reference's scan order (entry j -> partial j % 256), the soft score over all pixels (pixel p -> partial p % 64).
//   D3  transcendental functions are the deterministic kernels of det_math.h instead of libm.
//   D4  the 6x6 damped normal equations are solved by a Cholesky factorisation, falling back to a Jacobi eigen-decomposition
//       with OpenCV's SVBkSb threshold when a pivot is not safely positive (cv::solve(DECOMP_SVD) of a symmetric matrix:
//       the same solution up to rounding for the positive definite systems LM produces, the pseudo-inverse otherwise) and cvRodrigues2's matrix->vector branch
//       skips the SVD re-orthonormalisation of its input (the input is a rotation built from a unit quaternion).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <random>
#include <vector>

#include "det_math.h"

namespace {

// ----------------------------------------------------------------------------------------------------
// Test-time options (oracle_set_options): which 6x6 solver the LM step uses and which random stream sampleHypotheses draws from.
//   solver 0 (default)  D4': Cholesky, eigen pseudo-inverse only when a pivot is not safely positive (what the GPU kernel runs)
//   solver 1            always the eigen pseudo-inverse == cv::solve(DECOMP_SVD) of the symmetric system (what OpenCV runs)
//   rng 0 (default)     D1: counter-based stream keyed by (seed, frame, hypothesis, try, draw)
//   rng 1               reference style (thread_rand.cpp:13-42,68-71): one std::mt19937 per OpenMP thread, seeded seed + tid ONCE per
//                       process (init() is guarded by a static flag, dsacstar.cpp:80 calls init, not forceInit), irand(a, b) =
//                       std::uniform_int_distribution<int>(a, b - 1); hypothesis h is drawn by the thread libgomp's static schedule
//                       gives it (dsacstar_util.h:162 "#pragma omp parallel for", no schedule clause): contiguous blocks of
//                       ceil / floor (nH / T) iterations in thread order. The streams continue across calls until
//                       oracle_rng_reset(). NOTE: uniform_int_distribution is the one of THIS compiler's libstdc++ (GCC >= 11:
//                       Lemire's method); a reference built with an older GCC draws differently.
// ----------------------------------------------------------------------------------------------------
int g_solver_mode = 0, g_rng_mode = 0, g_rng_threads = 1;
std::vector<std::mt19937> g_generators;
bool g_rng_initialised = false;
void ref_rng_init(unsigned seed) {
  if (g_rng_initialised) return;
  g_generators.clear();
  for (int i = 0; i < g_rng_threads; ++i) {
    g_generators.push_back(std::mt19937());
    g_generators[i].seed((unsigned)i + seed);
  }
  g_rng_initialised = true;
}
int ref_irand(int incMin, int excMax, int tid) {
  std::uniform_int_distribution<int> dist(incMin, excMax - 1);
  return dist(g_generators[tid]);
}
// libgomp static schedule without chunk size: thread t of T gets q + (t < r) consecutive iterations, q = n / T, r = n % T
int ref_thread_of(int h, int n) {
  const int T = g_rng_threads, q = n / T, r = n % T;
  const int big = r * (q + 1);
  return (h < big) ? h / (q + 1) : r + (q ? (h - big) / q : 0);
}

// ----------------------------------------------------------------------------------------------------
// D1: counter-based random stream
// ----------------------------------------------------------------------------------------------------
inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
inline uint64_t try_key(uint64_t seed, uint64_t frame, uint32_t hyp, uint32_t tr) {
  return mix64(mix64(mix64(seed) ^ frame) ^ (((uint64_t)hyp << 32) | (uint64_t)tr));
}
// irand(0, n) of thread_rand.cpp:32-36 (uniform integer in [0, n))
inline int irand(uint64_t key, uint32_t draw, int n) {
  const uint64_t r = mix64(key + (uint64_t)draw * 0xD1B54A32D192ED03ULL);
  return (int)(((r >> 32) * (uint64_t)n) >> 32);
}

// ----------------------------------------------------------------------------------------------------
// D2: canonical reductions
// ----------------------------------------------------------------------------------------------------
inline double tree64(double* p) {  // p[0..63], butterfly order 32,16,...,1
  for (int off = 32; off >= 1; off >>= 1)
    for (int l = 0; l < off; ++l) p[l] = p[l] + p[l + off];
  return p[0];
}
inline double tree256(double* p) {  // four 64-lane groups, then ((g0+g1)+g2)+g3
  const double g0 = tree64(p), g1 = tree64(p + 64), g2 = tree64(p + 128), g3 = tree64(p + 192);
  return ((g0 + g1) + g2) + g3;
}

struct Pose {
  double r[3], t[3];
};
struct Cam {
  double fx, fy, cx, cy;  // double copies of the float camMat entries (dsacstar.cpp:91-95)
};

// ----------------------------------------------------------------------------------------------------
// [upstream] cvRodrigues2, vector -> matrix (+ 3x9 Jacobian dR/dr, row i = d R(:) / d r_i)
// ----------------------------------------------------------------------------------------------------
void rodrigues(const double rv[3], double R[9], double* J /* 27 or null */) {
  const double theta = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  if (theta < 2.220446049250313e-16) {
    for (int i = 0; i < 9; ++i) R[i] = 0;
    R[0] = R[4] = R[8] = 1;
    if (J) {
      for (int i = 0; i < 27; ++i) J[i] = 0;
      J[5] = J[15] = J[19] = -1;
      J[7] = J[11] = J[21] = 1;
    }
    return;
  }
  double s, c;
  detm::sincos(theta, &s, &c);
  const double c1 = 1. - c;
  const double itheta = theta ? 1. / theta : 0.;
  const double rx = rv[0] * itheta, ry = rv[1] * itheta, rz = rv[2] * itheta;
  const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
  const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; ++k) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
  if (J) {
    const double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0, 0, rx, 0, rx, ry + ry, rz, 0, rz, 0, 0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
    const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) {
      const double ri = i == 0 ? rx : i == 1 ? ry : rz;
      const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
      const double a3 = (c - s * itheta) * ri, a4 = s * itheta;
      for (int k = 0; k < 9; ++k) J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
    }
  }
}

// [upstream] cvRodrigues2, matrix -> vector (D4: without the SVD re-orthonormalisation)
void rodrigues_inv(const double R[9], double rv[3]) {
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = detm::acos_(c);
  if (s < 1e-5) {
    if (c > 0) {
      rx = ry = rz = 0;
    } else {
      double t;
      t = (R[0] + 1) * 0.5;
      rx = sqrt(t > 0. ? t : 0.);
      t = (R[4] + 1) * 0.5;
      ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
      t = (R[8] + 1) * 0.5;
      rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
      if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
      theta /= sqrt(rx * rx + ry * ry + rz * rz);
      rx *= theta;
      ry *= theta;
      rz *= theta;
    }
  } else {
    double vth = 1 / (2 * s);
    vth *= theta;
    rx *= vth;
    ry *= vth;
    rz *= vth;
  }
  rv[0] = rx;
  rv[1] = ry;
  rv[2] = rz;
}

// ----------------------------------------------------------------------------------------------------
// [upstream] cvProjectPoints2Internal without distortion; optional 2x6 Jacobian (dp/dr | dp/dt)
// ----------------------------------------------------------------------------------------------------
inline void project(const double R[9], const double t[3], const Cam& k, double X, double Y, double Z, double* u, double* v,
                    const double* dRdr /* 27 or null */, double* Ju /*6*/, double* Jv /*6*/) {
  double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
  double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
  double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
  z = z ? 1. / z : 1;
  x *= z;
  y *= z;
  *u = x * k.fx + k.cx;
  *v = y * k.fy + k.cy;
  if (dRdr) {
    const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
    for (int j = 0; j < 3; ++j) {
      Ju[3 + j] = k.fx * dxdt[j];
      Jv[3 + j] = k.fy * dydt[j];
    }
    const double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                             X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
    const double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                             X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
    const double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                             X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
    for (int j = 0; j < 3; ++j) {
      const double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
      const double dydr = z * (dy0dr[j] - y * dz0dr[j]);
      Ju[j] = k.fx * dxdr;
      Jv[j] = k.fy * dydr;
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// [upstream] polynom_solver.cpp
// ----------------------------------------------------------------------------------------------------
int solve_deg2(double a, double b, double c, double& x1, double& x2) {
  const double delta = b * b - 4 * a * c;
  if (delta < 0) return 0;
  const double inv_2a = 0.5 / a;
  if (delta == 0) {
    x1 = -b * inv_2a;
    x2 = x1;
    return 1;
  }
  const double sqrt_delta = sqrt(delta);
  x1 = (-b + sqrt_delta) * inv_2a;
  x2 = (-b - sqrt_delta) * inv_2a;
  return 2;
}

int solve_deg3(double a, double b, double c, double d, double& x0, double& x1, double& x2) {
  if (a == 0) {
    if (b == 0) {
      if (c == 0) return 0;
      x0 = -d / c;
      return 1;
    }
    x2 = 0;
    return solve_deg2(b, c, d, x0, x1);
  }
  const double inv_a = 1. / a;
  const double b_a = inv_a * b, b_a2 = b_a * b_a;
  const double c_a = inv_a * c;
  const double d_a = inv_a * d;
  const double Q = (3 * c_a - b_a2) / 9;
  const double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
  const double Q3 = Q * Q * Q;
  const double D = Q3 + R * R;
  const double b_a_3 = (1. / 3.) * b_a;
  if (Q == 0) {
    if (R == 0) {
      x0 = x1 = x2 = -b_a_3;
      return 3;
    } else {
      x0 = detm::cbrt_(2 * R) - b_a_3;  // pow(2*R, 1/3.0) upstream
      return 1;
    }
  }
  if (D <= 0) {
    const double theta = detm::acos_(R / sqrt(-Q3));
    const double sqrt_Q = sqrt(-Q);
    double s_, c0, c1, c2;
    detm::sincos(theta / 3.0, &s_, &c0);
    detm::sincos((theta + 2 * detm::PI) / 3.0, &s_, &c1);
    detm::sincos((theta + 4 * detm::PI) / 3.0, &s_, &c2);
    x0 = 2 * sqrt_Q * c0 - b_a_3;
    x1 = 2 * sqrt_Q * c1 - b_a_3;
    x2 = 2 * sqrt_Q * c2 - b_a_3;
    return 3;
  }
  const double AD = detm::cbrt_(fabs(R) + sqrt(D)) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
  const double BD = (AD == 0) ? 0 : -Q / AD;
  x0 = AD + BD - b_a_3;
  return 1;
}

int solve_deg4(double a, double b, double c, double d, double e, double& x0, double& x1, double& x2, double& x3) {
  if (a == 0) {
    x3 = 0;
    return solve_deg3(b, c, d, e, x0, x1, x2);
  }
  const double inv_a = 1. / a;
  b *= inv_a;
  c *= inv_a;
  d *= inv_a;
  e *= inv_a;
  const double b2 = b * b, bc = b * c, b3 = b2 * b;
  double r0 = 0, r1 = 0, r2 = 0;
  const int n = solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2);
  if (n == 0) return 0;
  const double R2 = 0.25 * b2 - c + r0;
  if (R2 < 0) return 0;
  const double R = sqrt(R2);
  const double inv_R = 1. / R;
  int nb_real_roots = 0;
  double D2, E2;
  if (R < 10E-12) {
    const double temp = r0 * r0 - 4 * e;
    if (temp < 0)
      D2 = E2 = -1;
    else {
      const double sqrt_temp = sqrt(temp);
      D2 = 0.75 * b2 - 2 * c + 2 * sqrt_temp;
      E2 = D2 - 4 * sqrt_temp;
    }
  } else {
    const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
    D2 = u + v;
    E2 = u - v;
  }
  const double b_4 = 0.25 * b, R_2 = 0.5 * R;
  if (D2 >= 0) {
    const double D = sqrt(D2);
    nb_real_roots = 2;
    const double D_2 = 0.5 * D;
    x0 = R_2 + D_2 - b_4;
    x1 = x0 - D;
  }
  if (E2 >= 0) {
    const double E = sqrt(E2);
    const double E_2 = 0.5 * E;
    if (nb_real_roots == 0) {
      x0 = -R_2 + E_2 - b_4;
      x1 = x0 - E;
      nb_real_roots = 2;
    } else {
      x2 = -R_2 + E_2 - b_4;
      x3 = x2 - E;
      nb_real_roots = 4;
    }
  }
  return nb_real_roots;
}

// ----------------------------------------------------------------------------------------------------
// cyclic Jacobi eigen-solver for a symmetric NxN matrix (Numerical Recipes form; p3p.cpp jacobi_4x4 for N = 4)
// A is destroyed; D receives the eigenvalues, U the eigenvectors (columns).
// ----------------------------------------------------------------------------------------------------
template <int N>
bool jacobi_sym(double* A, double* D, double* U) {
  double B[N], Z[N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) U[i * N + j] = (i == j) ? 1. : 0.;
  for (int i = 0; i < N; ++i) {
    B[i] = A[i * N + i];
    D[i] = B[i];
    Z[i] = 0;
  }
  for (int iter = 0; iter < 50; iter++) {
    double sum = 0;
    for (int i = 0; i < N - 1; ++i)
      for (int j = i + 1; j < N; ++j) sum += fabs(A[i * N + j]);
    if (sum == 0.0) return true;
    const double tresh = (iter < 3) ? 0.2 * sum / (double)(N * N) : 0.0;
    for (int i = 0; i < N - 1; i++) {
      for (int j = i + 1; j < N; j++) {
        const double Aij = A[i * N + j];
        const double eps_machine = 100.0 * fabs(Aij);
        if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j]))
          A[i * N + j] = 0.0;
        else if (fabs(Aij) > tresh) {
          double hh = D[j] - D[i], t;
          if (fabs(hh) + eps_machine == fabs(hh))
            t = Aij / hh;
          else {
            const double theta = 0.5 * hh / Aij;
            t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
            if (theta < 0.0) t = -t;
          }
          hh = t * Aij;
          Z[i] -= hh;
          Z[j] += hh;
          D[i] -= hh;
          D[j] += hh;
          A[i * N + j] = 0.0;
          const double c = 1.0 / sqrt(1 + t * t);
          const double s = t * c;
          const double tau = s / (1.0 + c);
          for (int k = 0; k <= i - 1; k++) {
            const double g = A[k * N + i], h = A[k * N + j];
            A[k * N + i] = g - s * (h + g * tau);
            A[k * N + j] = h + s * (g - h * tau);
          }
          for (int k = i + 1; k <= j - 1; k++) {
            const double g = A[i * N + k], h = A[k * N + j];
            A[i * N + k] = g - s * (h + g * tau);
            A[k * N + j] = h + s * (g - h * tau);
          }
          for (int k = j + 1; k < N; k++) {
            const double g = A[i * N + k], h = A[j * N + k];
            A[i * N + k] = g - s * (h + g * tau);
            A[j * N + k] = h + s * (g - h * tau);
          }
          for (int k = 0; k < N; k++) {
            const double g = U[k * N + i], h = U[k * N + j];
            U[k * N + i] = g - s * (h + g * tau);
            U[k * N + j] = h + s * (g - h * tau);
          }
        }
      }
    }
    for (int i = 0; i < N; i++) {
      B[i] += Z[i];
      D[i] = B[i];
      Z[i] = 0;
    }
  }
  return false;
}

// ----------------------------------------------------------------------------------------------------
// [upstream] p3p.cpp (Gao et al. 2003)
// ----------------------------------------------------------------------------------------------------
struct P3P {
  double fx, fy, cx, cy, inv_fx, inv_fy, cx_fx, cy_fy;
  explicit P3P(const Cam& k) {
    fx = k.fx; fy = k.fy; cx = k.cx; cy = k.cy;
    inv_fx = 1. / fx; inv_fy = 1. / fy; cx_fx = cx / fx; cy_fy = cy / fy;
  }

  int solve_for_lengths(double lengths[4][3], double distances[3], double cosines[3]) {
    const double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    const double inv_d22 = 1. / (distances[2] * distances[2]);
    const double a = inv_d22 * (distances[0] * distances[0]);
    const double b = inv_d22 * (distances[1] * distances[1]);
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    const double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;
    const double ab = a * b, a_2 = 2 * a;
    const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    const double a_4 = 4 * a;
    const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    const double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    const double b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double real_roots[4] = {0, 0, 0, 0};
    const int n = solve_deg4(A, B, C, D, E, real_roots[0], real_roots[1], real_roots[2], real_roots[3]);
    if (n == 0) return 0;
    int nb_solutions = 0;
    const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    const double inv_b0 = 1. / b0;
    for (int i = 0; i < n; i++) {
      const double x = real_roots[i];
      if (x <= 0) continue;
      const double x2 = x * x;
      const double b1 =
          ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
          (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
            (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
           (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
            pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
           2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
           p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
      if (b1 <= 0) continue;
      const double y = inv_b0 * b1;
      const double v = x2 + y * y - x * y * r;
      if (v <= 0) continue;
      const double Z = distances[2] / sqrt(v);
      const double X = x * Z;
      const double Y = y * Z;
      lengths[nb_solutions][0] = X;
      lengths[nb_solutions][1] = Y;
      lengths[nb_solutions][2] = Z;
      nb_solutions++;
    }
    return nb_solutions;
  }

  bool align(double M_end[3][3], double X0, double Y0, double Z0, double X1, double Y1, double Z1, double X2, double Y2, double Z2,
             double R[3][3], double T[3]) {
    double C_start[3], C_end[3];
    for (int i = 0; i < 3; i++) C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
    C_start[0] = (X0 + X1 + X2) / 3;
    C_start[1] = (Y0 + Y1 + Y2) / 3;
    C_start[2] = (Z0 + Z1 + Z2) / 3;
    double s[9];
    for (int j = 0; j < 3; j++) {
      s[0 * 3 + j] = (X0 * M_end[0][j] + X1 * M_end[1][j] + X2 * M_end[2][j]) / 3 - C_end[j] * C_start[0];
      s[1 * 3 + j] = (Y0 * M_end[0][j] + Y1 * M_end[1][j] + Y2 * M_end[2][j]) / 3 - C_end[j] * C_start[1];
      s[2 * 3 + j] = (Z0 * M_end[0][j] + Z1 * M_end[1][j] + Z2 * M_end[2][j]) / 3 - C_end[j] * C_start[2];
    }
    double Qs[16], evs[4], U[16];
    Qs[0 * 4 + 0] = s[0 * 3 + 0] + s[1 * 3 + 1] + s[2 * 3 + 2];
    Qs[1 * 4 + 1] = s[0 * 3 + 0] - s[1 * 3 + 1] - s[2 * 3 + 2];
    Qs[2 * 4 + 2] = s[1 * 3 + 1] - s[2 * 3 + 2] - s[0 * 3 + 0];
    Qs[3 * 4 + 3] = s[2 * 3 + 2] - s[0 * 3 + 0] - s[1 * 3 + 1];
    Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
    Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
    Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
    Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
    Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
    Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
    jacobi_sym<4>(Qs, evs, U);
    int i_ev = 0;
    double ev_max = evs[i_ev];
    for (int i = 1; i < 4; i++)
      if (evs[i] > ev_max) ev_max = evs[i_ev = i];
    double q[4];
    for (int i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
    const double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    const double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
    const double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3];
    const double q2_3 = q[2] * q[3];
    R[0][0] = q02 + q12 - q22 - q32;
    R[0][1] = 2. * (q1_2 - q0_3);
    R[0][2] = 2. * (q1_3 + q0_2);
    R[1][0] = 2. * (q1_2 + q0_3);
    R[1][1] = q02 + q22 - q12 - q32;
    R[1][2] = 2. * (q2_3 - q0_1);
    R[2][0] = 2. * (q1_3 - q0_2);
    R[2][1] = 2. * (q2_3 + q0_1);
    R[2][2] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++) T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
    return true;
  }

  // mu/mv are pixel coordinates (extract_points re-applies fx, cx to the normalised input)
  int solve(double R[4][3][3], double t[4][3], double mu0, double mv0, double X0, double Y0, double Z0, double mu1, double mv1,
            double X1, double Y1, double Z1, double mu2, double mv2, double X2, double Y2, double Z2, double mu3, double mv3,
            double X3, double Y3, double Z3) {
    double mk0, mk1, mk2, norm;
    mu0 = inv_fx * mu0 - cx_fx;
    mv0 = inv_fy * mv0 - cy_fy;
    norm = sqrt(mu0 * mu0 + mv0 * mv0 + 1);
    mk0 = 1. / norm; mu0 *= mk0; mv0 *= mk0;
    mu1 = inv_fx * mu1 - cx_fx;
    mv1 = inv_fy * mv1 - cy_fy;
    norm = sqrt(mu1 * mu1 + mv1 * mv1 + 1);
    mk1 = 1. / norm; mu1 *= mk1; mv1 *= mk1;
    mu2 = inv_fx * mu2 - cx_fx;
    mv2 = inv_fy * mv2 - cy_fy;
    norm = sqrt(mu2 * mu2 + mv2 * mv2 + 1);
    mk2 = 1. / norm; mu2 *= mk2; mv2 *= mk2;
    mu3 = inv_fx * mu3 - cx_fx;
    mv3 = inv_fy * mv3 - cy_fy;
    double distances[3];
    distances[0] = sqrt((X1 - X2) * (X1 - X2) + (Y1 - Y2) * (Y1 - Y2) + (Z1 - Z2) * (Z1 - Z2));
    distances[1] = sqrt((X0 - X2) * (X0 - X2) + (Y0 - Y2) * (Y0 - Y2) + (Z0 - Z2) * (Z0 - Z2));
    distances[2] = sqrt((X0 - X1) * (X0 - X1) + (Y0 - Y1) * (Y0 - Y1) + (Z0 - Z1) * (Z0 - Z1));
    double cosines[3];
    cosines[0] = mu1 * mu2 + mv1 * mv2 + mk1 * mk2;
    cosines[1] = mu0 * mu2 + mv0 * mv2 + mk0 * mk2;
    cosines[2] = mu0 * mu1 + mv0 * mv1 + mk0 * mk1;
    double lengths[4][3] = {};
    const int n = solve_for_lengths(lengths, distances, cosines);
    int nb_solutions = 0;
    double reproj_errors[4];
    for (int i = 0; i < n; i++) {
      double M_orig[3][3];
      M_orig[0][0] = lengths[i][0] * mu0; M_orig[0][1] = lengths[i][0] * mv0; M_orig[0][2] = lengths[i][0] * mk0;
      M_orig[1][0] = lengths[i][1] * mu1; M_orig[1][1] = lengths[i][1] * mv1; M_orig[1][2] = lengths[i][1] * mk1;
      M_orig[2][0] = lengths[i][2] * mu2; M_orig[2][1] = lengths[i][2] * mv2; M_orig[2][2] = lengths[i][2] * mk2;
      if (!align(M_orig, X0, Y0, Z0, X1, Y1, Z1, X2, Y2, Z2, R[nb_solutions], t[nb_solutions])) continue;
      const double X3p = R[nb_solutions][0][0] * X3 + R[nb_solutions][0][1] * Y3 + R[nb_solutions][0][2] * Z3 + t[nb_solutions][0];
      const double Y3p = R[nb_solutions][1][0] * X3 + R[nb_solutions][1][1] * Y3 + R[nb_solutions][1][2] * Z3 + t[nb_solutions][1];
      const double Z3p = R[nb_solutions][2][0] * X3 + R[nb_solutions][2][1] * Y3 + R[nb_solutions][2][2] * Z3 + t[nb_solutions][2];
      const double mu3p = X3p / Z3p;
      const double mv3p = Y3p / Z3p;
      reproj_errors[nb_solutions] = (mu3p - mu3) * (mu3p - mu3) + (mv3p - mv3) * (mv3p - mv3);
      nb_solutions++;
    }
    for (int i = 1; i < nb_solutions; i++) {  // insertion sort by the 4th point's error
      for (int j = i; j > 0 && reproj_errors[j - 1] > reproj_errors[j]; j--) {
        double tmp = reproj_errors[j]; reproj_errors[j] = reproj_errors[j - 1]; reproj_errors[j - 1] = tmp;
        for (int a = 0; a < 3; ++a) {
          for (int b = 0; b < 3; ++b) { tmp = R[j][a][b]; R[j][a][b] = R[j - 1][a][b]; R[j - 1][a][b] = tmp; }
          tmp = t[j][a]; t[j][a] = t[j - 1][a]; t[j - 1][a] = tmp;
        }
      }
    }
    return nb_solutions;
  }
};

// ----------------------------------------------------------------------------------------------------
// [upstream] solvePnP(SOLVEPNP_P3P) = solveP3P + take the first solution (solvepnp.cpp)
//   obj: 4 float 3-D points, img: 4 float pixel positions. Returns false when there is no solution.
// ----------------------------------------------------------------------------------------------------
bool solve_pnp_p3p(const float obj[4][3], const float img[4][2], const Cam& k, Pose* out) {
  // undistortPoints without distortion: output keeps the input depth (float)
  const double ifx = 1. / k.fx, ify = 1. / k.fy;
  float un[4][2];
  for (int i = 0; i < 4; ++i) {
    un[i][0] = (float)(((double)img[i][0] - k.cx) * ifx);
    un[i][1] = (float)(((double)img[i][1] - k.cy) * ify);
  }
  // p3p::extract_points: back to pixels in double
  double pts[4][5];
  for (int i = 0; i < 4; ++i) {
    pts[i][0] = (double)un[i][0] * k.fx + k.cx;
    pts[i][1] = (double)un[i][1] * k.fy + k.cy;
    pts[i][2] = obj[i][0];
    pts[i][3] = obj[i][1];
    pts[i][4] = obj[i][2];
  }
  P3P solver(k);
  double Rs[4][3][3] = {}, ts[4][3] = {};
  const int solutions =
      solver.solve(Rs, ts, pts[0][0], pts[0][1], pts[0][2], pts[0][3], pts[0][4], pts[1][0], pts[1][1], pts[1][2], pts[1][3], pts[1][4],
                   pts[2][0], pts[2][1], pts[2][2], pts[2][3], pts[2][4], pts[3][0], pts[3][1], pts[3][2], pts[3][3], pts[3][4]);
  if (solutions == 0) return false;
  double rvecs[4][3], errs[4];
  for (int i = 0; i < solutions; ++i) {
    double Rm[9];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) Rm[a * 3 + b] = Rs[i][a][b];
    rodrigues_inv(Rm, rvecs[i]);
    double R2[9];
    rodrigues(rvecs[i], R2, nullptr);
    double e = 0;
    for (int j = 0; j < 4; ++j) {
      double u, v;
      project(R2, ts[i], k, obj[j][0], obj[j][1], obj[j][2], &u, &v, nullptr, nullptr, nullptr);
      const double ex = (double)img[j][0] - u, ey = (double)img[j][1] - v;
      e += ex * ex;
      e += ey * ey;
    }
    errs[i] = e;
  }
  for (int i = 1; i < solutions; i++) {  // stable insertion sort by total reprojection error
    for (int j = i; j > 0 && errs[j - 1] > errs[j]; j--) {
      double tmp = errs[j]; errs[j] = errs[j - 1]; errs[j - 1] = tmp;
      for (int a = 0; a < 3; ++a) {
        tmp = rvecs[j][a]; rvecs[j][a] = rvecs[j - 1][a]; rvecs[j - 1][a] = tmp;
        tmp = ts[j][a]; ts[j][a] = ts[j - 1][a]; ts[j - 1][a] = tmp;
      }
    }
  }
  for (int a = 0; a < 3; ++a) {
    out->r[a] = rvecs[0][a];
    out->t[a] = ts[0][a];
  }
  return true;
}

// ----------------------------------------------------------------------------------------------------
// D4: x = pinv(A) b for symmetric A (6x6) via Jacobi eigen-decomposition, SVBkSb threshold
// ----------------------------------------------------------------------------------------------------
void solve_sym6(const double A_in[36], const double b[6], double x[6]) {
  double A[36], w[6], V[36];
  for (int i = 0; i < 36; ++i) A[i] = A_in[i];
  jacobi_sym<6>(A, w, V);
  double threshold = 0;
  for (int i = 0; i < 6; ++i) threshold += fabs(w[i]);
  threshold *= 2.220446049250313e-16 * 2;
  for (int i = 0; i < 6; ++i) x[i] = 0;
  for (int i = 0; i < 6; ++i) {
    if (fabs(w[i]) <= threshold) continue;
    double s = 0;
    for (int k = 0; k < 6; ++k) s += V[k * 6 + i] * b[k];
    s = s / w[i];
    for (int k = 0; k < 6; ++k) x[k] += s * V[k * 6 + i];
  }
}

// ----------------------------------------------------------------------------------------------------
// D4': the damped normal equations (J^T J with its diagonal scaled by 1 + lambda) are symmetric positive definite unless the
// inlier set is degenerate, so they are solved by a Cholesky factorisation (6 square roots + 6 reciprocals on the critical
// path: ~2 us on a GPU lane, where the 50-sweep-capable Jacobi eigen-solve above is ~55 us of serial fp64 divisions and square
// roots); when a pivot is not safely positive (rank-deficient system) the eigen pseudo-inverse of D4 is used, as
// cv::solve(DECOMP_SVD) would behave. Fixed operation order: the oracle and the kernel run the same sequence.
// ----------------------------------------------------------------------------------------------------
 void solve_normal6(const double A[36], const double b[6], double x[6]) {
  if (g_solver_mode == 1) { solve_sym6(A, b, x); return; }
  double L[36], Linv[6], y[6];
  bool ok = true;
  for (int j = 0; j < 6; ++j) {
    double s = A[j * 6 + j];
    for (int k = 0; k < j; ++k) s = s - L[j * 6 + k] * L[j * 6 + k];
    if (!(s > A[j * 6 + j] * 1e-12)) ok = false;
    const double d = sqrt(ok ? s : 1.0);
    const double inv = 1.0 / d;
    L[j * 6 + j] = d;
    Linv[j] = inv;
    for (int i = j + 1; i < 6; ++i) {
      double t = A[i * 6 + j];
      for (int k = 0; k < j; ++k) t = t - L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = t * inv;
    }
  }
  for (int i = 0; i < 6; ++i) {
    double t = b[i];
    for (int k = 0; k < i; ++k) t = t - L[i * 6 + k] * y[k];
    y[i] = t * Linv[i];
  }
  for (int i = 5; i >= 0; --i) {
    double t = y[i];
    for (int k = i + 1; k < 6; ++k) t = t - L[k * 6 + i] * x[k];
    x[i] = t * Linv[i];
  }
  if (!ok) solve_sym6(A, b, x);
}

struct Frame {
  const float* sc;
  int64_t sC, sH, sW;
  int H, W;
  int sub;
  inline void coord(int x, int y, float o[3]) const {
    o[0] = sc[0 * sC + y * sH + x * sW];
    o[1] = sc[1 * sC + y * sH + x * sW];
    o[2] = sc[2 * sC + y * sH + x * sW];
  }
  // createSampling (dsacstar_util.h:59-76), shift 0
  inline int px(int x) const { return x * sub + sub / 2; }
  inline int py(int y) const { return y * sub + sub / 2; }
};

// getReproErrs (dsacstar_util.h:356-446, calcJ = false): error image in pixel order p = x * H + y
void repro_errs(const Frame& f, const Pose& hyp, const Cam& k, float maxReproj, std::vector<float>& errs) {
  double R[9];
  rodrigues(hyp.r, R, nullptr);
  errs.resize((size_t)f.W * f.H);
  for (int x = 0; x < f.W; x++)
    for (int y = 0; y < f.H; y++) {
      float c[3];
      f.coord(x, y, c);
      double u, v;
      project(R, hyp.t, k, c[0], c[1], c[2], &u, &v, nullptr, nullptr, nullptr);
      const float pu = (float)u, pv = (float)v;                    // projections are Point2f
      const float dx = (float)f.px(x) - pu, dy = (float)f.py(y) - pv;  // Point2f - Point2f
      const double nrm = sqrt((double)dx * dx + (double)dy * dy);  // cv::norm(Point2f)
      const float l = (float)nrm;
      errs[(size_t)x * f.H + y] = l < maxReproj ? l : maxReproj;    // std::min((float)norm, maxReproj)
    }
}

// [upstream] cvFindExtrinsicCameraParams2 with useExtrinsicGuess + CvLevMarq(6, 2n, {20 iter, FLT_EPSILON}),
// on the pixels whose flag is set. D2: accumulations in the 256-way canonical order.
struct LMAccum {
  double JtJ[36], JtErr[6], errsq;
};

void lm_accumulate(const Frame& f, const std::vector<uint8_t>& flags, const double param[6], const Cam& k, bool withJ, LMAccum* out) {
  double R[9], dRdr[27];
  rodrigues(param, R, withJ ? dRdr : nullptr);
  const double* t = param + 3;
  const int N = f.W * f.H;
  // the inlier list in scan order: refineHyp's localImgPts / localObjPts (dsacstar_util.h:545-560), the order in which
  // cvFindExtrinsicCameraParams2 walks its correspondences
  static thread_local std::vector<int> list;
  list.clear();
  for (int p = 0; p < N; ++p)
    if (flags[p]) list.push_back(p);
  const int cnt = (int)list.size();
  // 28 quantities x 256 partials: list entry j adds to partial j % 256
  static thread_local std::vector<double> part;
  part.assign(28 * 256, 0.0);
  for (int lane = 0; lane < 256; ++lane) {
    double acc[28];
    for (int i = 0; i < 28; ++i) acc[i] = 0;
    for (int j = lane; j < cnt; j += 256) {
      const int p = list[j];
      const int x = p / f.H, y = p % f.H;
      float c[3];
      f.coord(x, y, c);
      double u, v, Ju[6], Jv[6];
      project(R, t, k, c[0], c[1], c[2], &u, &v, withJ ? dRdr : nullptr, Ju, Jv);
      const double eu = u - (double)(float)f.px(x), ev = v - (double)(float)f.py(y);
      if (withJ) {
        int q = 0;
        for (int a = 0; a < 6; ++a)
          for (int b = a; b < 6; ++b) {
            acc[q] = acc[q] + Ju[a] * Ju[b];
            acc[q] = acc[q] + Jv[a] * Jv[b];
            ++q;
          }
        for (int a = 0; a < 6; ++a) {
          acc[21 + a] = acc[21 + a] + Ju[a] * eu;
          acc[21 + a] = acc[21 + a] + Jv[a] * ev;
        }
      }
      acc[27] = acc[27] + eu * eu;
      acc[27] = acc[27] + ev * ev;
    }
    for (int i = 0; i < 28; ++i) part[(size_t)i * 256 + lane] = acc[i];
  }
  double red[28];
  for (int i = 0; i < 28; ++i) red[i] = tree256(&part[(size_t)i * 256]);
  int q = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) {
      out->JtJ[a * 6 + b] = red[q];
      out->JtJ[b * 6 + a] = red[q];
      ++q;
    }
  for (int a = 0; a < 6; ++a) out->JtErr[a] = red[21 + a];
  out->errsq = red[27];
}

void lm_step(const LMAccum& acc, const double prevParam[6], int lambdaLg10, double param[6]) {
  const double lambda = detm::pow10i(lambdaLg10);
  double A[36], x[6];
  for (int i = 0; i < 36; ++i) A[i] = acc.JtJ[i];
  for (int i = 0; i < 6; ++i) A[i * 6 + i] *= 1. + lambda;
  solve_normal6(A, acc.JtErr, x);
  for (int i = 0; i < 6; ++i) param[i] = prevParam[i] - x[i];
}

typedef std::function<void(const double*, bool, LMAccum*)> AccumFn;
void lm_solve(const AccumFn& accumulate, Pose* pose) {
  const int max_iter = 20;
  const double epsilon = 1.1920928955078125e-07;  // FLT_EPSILON
  double param[6] = {pose->r[0], pose->r[1], pose->r[2], pose->t[0], pose->t[1], pose->t[2]};
  double prevParam[6];
  int lambdaLg10 = -3, iters = 0;
  double prevErrNorm = 1.7976931348623157e308, errNorm;
  LMAccum acc, tmp;
  accumulate(param, true, &acc);  // state STARTED -> CALC_J
  for (;;) {
    // CALC_J
    for (int i = 0; i < 6; ++i) prevParam[i] = param[i];
    lm_step(acc, prevParam, lambdaLg10, param);
    if (iters == 0) prevErrNorm = sqrt(acc.errsq);
    // CHECK_ERR
    for (;;) {
      accumulate(param, false, &tmp);
      errNorm = sqrt(tmp.errsq);
      if (errNorm > prevErrNorm) {
        if (++lambdaLg10 <= 16) {
          lm_step(acc, prevParam, lambdaLg10, param);
          continue;
        }
      }
      break;
    }
    lambdaLg10 = (lambdaLg10 - 1 > -16) ? lambdaLg10 - 1 : -16;
    double dn = 0, pn = 0;
    for (int i = 0; i < 6; ++i) {
      const double d = param[i] - prevParam[i];
      dn += d * d;
      pn += prevParam[i] * prevParam[i];
    }
    const double rel = sqrt(dn) / (sqrt(pn) + 2.220446049250313e-16);  // cvNorm(param, prevParam, CV_RELATIVE_L2)
    if (++iters >= max_iter || rel < epsilon) break;
    prevErrNorm = errNorm;
    accumulate(param, true, &acc);
  }
  for (int i = 0; i < 3; ++i) {
    pose->r[i] = param[i];
    pose->t[i] = param[3 + i];
  }
}

void solve_pnp_iterative(const Frame& f, const std::vector<uint8_t>& flags, const Cam& k, Pose* pose) {
  lm_solve([&](const double* param, bool withJ, LMAccum* out) { lm_accumulate(f, flags, param, k, withJ, out); }, pose);
}

// the same solver on an arbitrary list of correspondences (float object / image points), accumulated in list order: used to
// consume golden vectors of cv::solvePnP(ITERATIVE, useExtrinsicGuess) produced in the reference's environment
void solve_pnp_iterative_pts(const float* obj, const float* img, int n, const Cam& k, Pose* pose) {
  lm_solve(
      [&](const double* param, bool withJ, LMAccum* out) {
        double R[9], dRdr[27];
        rodrigues(param, R, withJ ? dRdr : nullptr);
        for (int i = 0; i < 36; ++i) out->JtJ[i] = 0;
        for (int i = 0; i < 6; ++i) out->JtErr[i] = 0;
        out->errsq = 0;
        for (int p = 0; p < n; ++p) {
          double u, v, Ju[6], Jv[6];
          project(R, param + 3, k, obj[3 * p], obj[3 * p + 1], obj[3 * p + 2], &u, &v, withJ ? dRdr : nullptr, Ju, Jv);
          const double eu = u - (double)img[2 * p], ev = v - (double)img[2 * p + 1];
          if (withJ)
            for (int a = 0; a < 6; ++a) {
              for (int b = 0; b < 6; ++b) out->JtJ[a * 6 + b] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
              out->JtErr[a] += Ju[a] * eu + Jv[a] * ev;
            }
          out->errsq += eu * eu + ev * ev;
        }
      },
      pose);
}

// [upstream] cv::Mat::inv() of a 4x4 double matrix (DECOMP_LU, hal::LU64f): returns false if singular
bool inv4x4(const double Ain[16], double out[16]) {
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

}  // namespace

// ====================================================================================================
// C interface for the tests (ctypes)
// ====================================================================================================
extern "C" {

// dsacstar_rgb_forward (dsacstar.cpp:66-186). Returns 0. Optional debug outputs may be null.
int oracle_forward_rgb(const float* sc, int64_t strideC, int64_t strideH, int64_t strideW, int H, int W, int ransacHypotheses,
                       float inlierThreshold, float focalLength, float ppointX, float ppointY, float inlierAlpha, float maxReproj,
                       int subSampling, uint64_t randomSeed, uint64_t frameId, int maxTries, int maxRefSteps, float* outPose16,
                       int* outInliers, uint8_t* outMask, double* dbgHypPoses, double* dbgScores, int* dbgBest, double* dbgRefined) {
  Frame f{sc, strideC, strideH, strideW, H, W, subSampling};
  Cam k{(double)focalLength, (double)focalLength, (double)ppointX, (double)ppointY};
  const int N = W * H;
  if (maxRefSteps <= 0) maxRefSteps = 100;  // MAX_REF_STEPS, dsacstar.cpp:47

  // ---- sampleHypotheses (dsacstar_util.h:135-221)
  if (g_rng_mode == 1) ref_rng_init((unsigned)randomSeed);   // ThreadRand::init(randomSeed), dsacstar.cpp:80
  std::vector<Pose> hyps(ransacHypotheses);
  for (int h = 0; h < ransacHypotheses; h++) {
    Pose cur;
    memset(&cur, 0, sizeof(cur));
    for (int t = 0; t < maxTries; t++) {
      const uint64_t key = try_key(randomSeed, frameId, (uint32_t)h, (uint32_t)t);
      float obj[4][3], img[4][2];
      for (int j = 0; j < 4; j++) {
        const int x = g_rng_mode == 1 ? ref_irand(0, W, ref_thread_of(h, ransacHypotheses)) : irand(key, 2 * j, W);   // x first, then y
        const int y = g_rng_mode == 1 ? ref_irand(0, H, ref_thread_of(h, ransacHypotheses)) : irand(key, 2 * j + 1, H);
        img[j][0] = (float)f.px(x);
        img[j][1] = (float)f.py(y);
        f.coord(x, y, obj[j]);
      }
      if (!solve_pnp_p3p(obj, img, k, &cur)) {
        memset(&cur, 0, sizeof(cur));  // safeSolvePnP zeroes the pose (dsacstar_util.h:114-116)
        continue;
      }
      double R[9];
      rodrigues(cur.r, R, nullptr);
      bool foundOutlier = false;
      for (int j = 0; j < 4; j++) {
        double u, v;
        project(R, cur.t, k, obj[j][0], obj[j][1], obj[j][2], &u, &v, nullptr, nullptr, nullptr);
        const float dx = img[j][0] - (float)u, dy = img[j][1] - (float)v;
        if (sqrt((double)dx * dx + (double)dy * dy) < (double)inlierThreshold) continue;
        foundOutlier = true;
        break;
      }
      if (foundOutlier) continue;
      break;
    }
    hyps[h] = cur;
    if (dbgHypPoses)
      for (int i = 0; i < 3; ++i) {
        dbgHypPoses[h * 6 + i] = cur.r[i];
        dbgHypPoses[h * 6 + 3 + i] = cur.t[i];
      }
  }

  // ---- getReproErrs + getHypScores (dsacstar.cpp:125-143, dsacstar_util.h:316-343)
  std::vector<double> scores(ransacHypotheses, 0.0);
  const float inlierBeta = 5 / inlierThreshold;
  std::vector<float> errs;
  for (int h = 0; h < ransacHypotheses; h++) {
    repro_errs(f, hyps[h], k, maxReproj, errs);
    double part[64];
    for (int l = 0; l < 64; ++l) {
      double acc = 0;
      for (int p = l; p < N; p += 64) {
        double softThreshold = inlierBeta * (errs[p] - inlierThreshold);  // float product, widened
        softThreshold = 1 / (1 + detm::exp_(-softThreshold));
        acc += 1 - softThreshold;
      }
      part[l] = acc;
    }
    scores[h] = tree64(part);
    scores[h] *= inlierAlpha / W / H;  // float / int / int
    if (dbgScores) dbgScores[h] = scores[h];
  }

  // ---- softMax + draw(argmax) (dsacstar_util.h:684-752)
  int hypIdx = 0;
  {
    double maxScore = 0;
    for (int i = 0; i < ransacHypotheses; i++)
      if (i == 0 || scores[i] > maxScore) maxScore = scores[i];
    std::vector<double> sf(ransacHypotheses);
    double sum = 0.0;
    for (int i = 0; i < ransacHypotheses; i++) {
      sf[i] = detm::exp_(scores[i] - maxScore);
      sum += sf[i];
    }
    for (int i = 0; i < ransacHypotheses; i++) sf[i] /= sum;
    double maxProb = -1;
    int maxIdx = 0;
    for (int idx = 0; idx < ransacHypotheses; idx++) {
      if (sf[idx] < 0.00000001) continue;
      if (maxProb < 0 || sf[idx] > maxProb) {
        maxProb = sf[idx];
        maxIdx = idx;
      }
    }
    hypIdx = maxIdx;
  }
  if (dbgBest) *dbgBest = hypIdx;

  // ---- refineHyp (dsacstar_util.h:522-597)
  Pose hyp = hyps[hypIdx];
  std::vector<uint8_t> inlierMap;  // empty == cv::Mat() (dsacstar.cpp:161)
  {
    std::vector<float> localErrs;
    repro_errs(f, hyp, k, maxReproj, localErrs);
    unsigned bestInliers = 4;
    for (int rStep = 0; rStep < maxRefSteps; rStep++) {
      std::vector<uint8_t> localMap((size_t)N, 0);
      unsigned cnt = 0;
      for (int p = 0; p < N; ++p)
        if (localErrs[p] < inlierThreshold) {
          localMap[p] = 1;
          cnt++;
        }
      if (cnt <= bestInliers) break;
      bestInliers = cnt;
      Pose upd = hyp;
      solve_pnp_iterative(f, localMap, k, &upd);  // cnt > 4 always here -> SOLVEPNP_ITERATIVE (dsacstar_util.h:577-580)
      hyp = upd;
      inlierMap = localMap;
      repro_errs(f, hyp, k, maxReproj, localErrs);
    }
  }
  if (dbgRefined)
    for (int i = 0; i < 3; ++i) {
      dbgRefined[i] = hyp.r[i];
      dbgRefined[3 + i] = hyp.t[i];
    }

  // ---- pose2trans (dsacstar_util.h:759-770) + write-out (dsacstar.cpp:177-185)
  double R[9], T[16], Ti[16];
  rodrigues(hyp.r, R, nullptr);
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1. : 0.;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) T[a * 4 + b] = R[a * 3 + b];
    T[a * 4 + 3] = hyp.t[a];
  }
  if (!inv4x4(T, Ti))
    for (int i = 0; i < 16; ++i) Ti[i] = 0;  // cv::Mat::inv of a singular matrix yields zeros
  for (int i = 0; i < 16; ++i) outPose16[i] = (float)Ti[i];
  int count = 0;
  for (size_t p = 0; p < inlierMap.size(); ++p) count += inlierMap[p];
  *outInliers = count;
  if (outMask) {
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) outMask[y * W + x] = inlierMap.empty() ? 0 : inlierMap[(size_t)x * H + y];
  }
  return 0;
}

// ---- options ------------------------------------------------------------------------------------------
void oracle_set_options(int solver_mode, int rng_mode, int rng_threads) {
  g_solver_mode = solver_mode;
  g_rng_mode = rng_mode;
  g_rng_threads = rng_threads > 0 ? rng_threads : 1;
  g_rng_initialised = false;
}
void oracle_rng_reset() { g_rng_initialised = false; }
// cv::solvePnP(obj, img, K, noArray, rvec, tvec, useExtrinsicGuess = true, SOLVEPNP_ITERATIVE) on a point list
void oracle_pnp_iterative_pts(const float* obj3n, const float* img2n, int n, float focal, float ppx, float ppy, double* pose6) {
  Cam k{(double)focal, (double)focal, (double)ppx, (double)ppy};
  Pose p;
  for (int i = 0; i < 3; ++i) { p.r[i] = pose6[i]; p.t[i] = pose6[3 + i]; }
  solve_pnp_iterative_pts(obj3n, img2n, n, k, &p);
  for (int i = 0; i < 3; ++i) { pose6[i] = p.r[i]; pose6[3 + i] = p.t[i]; }
}

// ---- unit hooks --------------------------------------------------------------------------------------
int oracle_p3p(const float* obj12, const float* img8, float focal, float ppx, float ppy, double* pose6) {
  float obj[4][3], img[4][2];
  memcpy(obj, obj12, sizeof(obj));
  memcpy(img, img8, sizeof(img));
  Cam k{(double)focal, (double)focal, (double)ppx, (double)ppy};
  Pose p;
  memset(&p, 0, sizeof(p));
  const bool ok = solve_pnp_p3p(obj, img, k, &p);
  for (int i = 0; i < 3; ++i) {
    pose6[i] = ok ? p.r[i] : 0;
    pose6[3 + i] = ok ? p.t[i] : 0;
  }
  return ok ? 1 : 0;
}
void oracle_rodrigues(const double* r3, double* R9, double* J27) { rodrigues(r3, R9, J27); }
void oracle_rodrigues_inv(const double* R9, double* r3) { rodrigues_inv(R9, r3); }
void oracle_project(const double* pose6, float focal, float ppx, float ppy, const float* xyz, int n, double* uv, double* J12n) {
  Cam k{(double)focal, (double)focal, (double)ppx, (double)ppy};
  double R[9], dRdr[27];
  rodrigues(pose6, R, dRdr);
  for (int i = 0; i < n; ++i)
    project(R, pose6 + 3, k, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &uv[2 * i], &uv[2 * i + 1], J12n ? dRdr : nullptr,
            J12n ? J12n + 12 * i : nullptr, J12n ? J12n + 12 * i + 6 : nullptr);
}
int oracle_solve_deg4(const double* c5, double* roots4) {
  return solve_deg4(c5[0], c5[1], c5[2], c5[3], c5[4], roots4[0], roots4[1], roots4[2], roots4[3]);
}
void oracle_solve_sym6(const double* A36, const double* b6, double* x6) { solve_sym6(A36, b6, x6); }
void oracle_solve_normal6(const double* A36, const double* b6, double* x6) { solve_normal6(A36, b6, x6); }
int oracle_inv4x4(const double* A16, double* out16) { return inv4x4(A16, out16) ? 1 : 0; }
void oracle_det_math(const double* x, int n, double* s, double* c, double* ac, double* ex, double* cb) {
  for (int i = 0; i < n; ++i) {
    detm::sincos(x[i], &s[i], &c[i]);
    ac[i] = detm::acos_(x[i] > 1 ? 1 : (x[i] < -1 ? -1 : x[i]));
    ex[i] = detm::exp_(x[i]);
    cb[i] = detm::cbrt_(x[i]);
  }
}
// refinement alone: LM-PnP on flagged pixels from an initial pose
void oracle_pnp_iterative(const float* sc, int64_t strideC, int64_t strideH, int64_t strideW, int H, int W, int sub, float focal,
                          float ppx, float ppy, const uint8_t* flags_xmajor, double* pose6) {
  Frame f{sc, strideC, strideH, strideW, H, W, sub};
  Cam k{(double)focal, (double)focal, (double)ppx, (double)ppy};
  std::vector<uint8_t> fl(flags_xmajor, flags_xmajor + (size_t)H * W);
  Pose p;
  for (int i = 0; i < 3; ++i) { p.r[i] = pose6[i]; p.t[i] = pose6[3 + i]; }
  solve_pnp_iterative(f, fl, k, &p);
  for (int i = 0; i < 3; ++i) { pose6[i] = p.r[i]; pose6[3 + i] = p.t[i]; }
}
}
