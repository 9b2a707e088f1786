// ransac_math.h -- scalar fp64 geometry of the DSAC* path, usable from host and device code.
//
// These are the per-hypothesis / per-pixel formulas of OpenCV 4.4.0's solvePnP(P3P), solvePnP(ITERATIVE),
// projectPoints and Rodrigues as the reference calls them (dsacstar_util.h:104-112,199-205,395-401,570-580,762)
// [upstream semantics; see DESIGN.md "R path"].  The RANSAC parity gate is BIT-exact against the CPU oracle, so the
// fp64 expressions here are written operation-for-operation in the oracle's order (compile with
// -ffp-contract=off); what differs from the oracle is everything around them: the wave/workgroup decomposition,
// LDS staging and the shuffle reductions in ransac_api.hip.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifndef ACEZ_HD
#define ACEZ_HD __host__ __device__
#endif
#include "det_math.h"

namespace rsm {

// ----------------------------------------------------------------------------------------------------
// D1: counter-based random stream
// ----------------------------------------------------------------------------------------------------
ACEZ_HD inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
ACEZ_HD inline uint64_t try_key(uint64_t seed, uint64_t frame, uint32_t hyp, uint32_t tr) {
  return mix64(mix64(mix64(seed) ^ frame) ^ (((uint64_t)hyp << 32) | (uint64_t)tr));
}
// irand(0, n) of thread_rand.cpp:32-36 (uniform integer in [0, n))
ACEZ_HD inline int irand(uint64_t key, uint32_t draw, int n) {
  const uint64_t r = mix64(key + (uint64_t)draw * 0xD1B54A32D192ED03ULL);
  return (int)(((r >> 32) * (uint64_t)n) >> 32);
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
ACEZ_HD inline void rodrigues(const double rv[3], double R[9], double* J /* 27 or null */) {
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
ACEZ_HD inline void rodrigues_inv(const double R[9], double rv[3]) {
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
ACEZ_HD inline void project(const double R[9], const double t[3], const Cam& k, double X, double Y, double Z, double* u, double* v,
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
ACEZ_HD inline int solve_deg2(double a, double b, double c, double& x1, double& x2) {
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

ACEZ_HD inline int solve_deg3(double a, double b, double c, double d, double& x0, double& x1, double& x2) {
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

ACEZ_HD inline int solve_deg4(double a, double b, double c, double d, double e, double& x0, double& x1, double& x2, double& x3) {
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
// Small arrays that must live in registers on the GPU: a dynamically indexed local array is placed in scratch memory there (every
// access a round trip to the vector cache), so dynamic positions are read and written through compare-and-select chains and every
// loop over such an array is fully unrolled. On the host the same code is ordinary scalar code.
// ----------------------------------------------------------------------------------------------------
ACEZ_HD inline double pick4(const double v[4], int i) {
  double r = v[0];
  r = (i == 1) ? v[1] : r;
  r = (i == 2) ? v[2] : r;
  r = (i == 3) ? v[3] : r;
  return r;
}
ACEZ_HD inline void put4(double v[4], int i, double x) {
  v[0] = (i == 0) ? x : v[0];
  v[1] = (i == 1) ? x : v[1];
  v[2] = (i == 2) ? x : v[2];
  v[3] = (i == 3) ? x : v[3];
}
// the stable insertion sort p3p.cpp / solvepnp.cpp run on up to four solutions (strict >, inner loop stops at the first pair in
// order), on the keys and a permutation instead of on the poses themselves
ACEZ_HD inline void insertion_sort4(double key[4], int ord[4], int m) {
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    bool moving = i < m;
#pragma unroll
    for (int j = i; j > 0; --j) {
      const bool sw = moving && key[j - 1] > key[j];
      const double ka = key[j - 1], kb = key[j];
      const int oa = ord[j - 1], ob = ord[j];
      key[j - 1] = sw ? kb : ka;
      key[j] = sw ? ka : kb;
      ord[j - 1] = sw ? ob : oa;
      ord[j] = sw ? oa : ob;
      moving = sw;
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// cyclic Jacobi eigen-solver for a symmetric NxN matrix (Numerical Recipes form; p3p.cpp jacobi_4x4 for N = 4)
// A is destroyed; D receives the eigenvalues, U the eigenvectors (columns). For N = 4 (once per P3P solution, on every lane of the
// sampling wavefronts) the pair loops are unrolled so that A, U, D stay in registers; N = 6 (the rare fallback of the LM solve) is
// left rolled.
// ----------------------------------------------------------------------------------------------------
template <int N>
ACEZ_HD inline bool jacobi_sym(double* A, double* D, double* U) {
  constexpr int UNR = (N <= 4) ? N : 1;
  double B[N], Z[N];
#pragma unroll UNR
  for (int i = 0; i < N; ++i)
#pragma unroll UNR
    for (int j = 0; j < N; ++j) U[i * N + j] = (i == j) ? 1. : 0.;
#pragma unroll UNR
  for (int i = 0; i < N; ++i) {
    B[i] = A[i * N + i];
    D[i] = B[i];
    Z[i] = 0;
  }
#pragma unroll 1
  for (int iter = 0; iter < 50; iter++) {
    double sum = 0;
#pragma unroll UNR
    for (int i = 0; i < N - 1; ++i)
#pragma unroll UNR
      for (int j = i + 1; j < N; ++j) sum += fabs(A[i * N + j]);
    if (sum == 0.0) return true;
    const double tresh = (iter < 3) ? 0.2 * sum / (double)(N * N) : 0.0;
#pragma unroll UNR
    for (int i = 0; i < N - 1; i++) {
#pragma unroll UNR
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
#pragma unroll UNR
          for (int k = 0; k <= i - 1; k++) {
            const double g = A[k * N + i], h = A[k * N + j];
            A[k * N + i] = g - s * (h + g * tau);
            A[k * N + j] = h + s * (g - h * tau);
          }
#pragma unroll UNR
          for (int k = i + 1; k <= j - 1; k++) {
            const double g = A[i * N + k], h = A[k * N + j];
            A[i * N + k] = g - s * (h + g * tau);
            A[k * N + j] = h + s * (g - h * tau);
          }
#pragma unroll UNR
          for (int k = j + 1; k < N; k++) {
            const double g = A[i * N + k], h = A[j * N + k];
            A[i * N + k] = g - s * (h + g * tau);
            A[j * N + k] = h + s * (g - h * tau);
          }
#pragma unroll UNR
          for (int k = 0; k < N; k++) {
            const double g = U[k * N + i], h = U[k * N + j];
            U[k * N + i] = g - s * (h + g * tau);
            U[k * N + j] = h + s * (g - h * tau);
          }
        }
      }
    }
#pragma unroll UNR
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
  ACEZ_HD explicit P3P(const Cam& k) {
    fx = k.fx; fy = k.fy; cx = k.cx; cy = k.cy;
    inv_fx = 1. / fx; inv_fy = 1. / fy; cx_fx = cx / fx; cy_fy = cy / fy;
  }

  // p3p::solve_for_lengths: the depths (X, Y, Z) of the three points along their viewing rays, one triple per admissible root
  ACEZ_HD int solve_for_lengths(double lenX[4], double lenY[4], double lenZ[4], const double distances[3], const double cosines[3]) {
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
#pragma unroll 1
    for (int i = 0; i < n; i++) {
      const double x = pick4(real_roots, i);
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
      put4(lenX, nb_solutions, X);
      put4(lenY, nb_solutions, Y);
      put4(lenZ, nb_solutions, Z);
      nb_solutions++;
    }
    return nb_solutions;
  }

  // p3p::align: rotation + translation that carry the three world points onto the three camera-frame points (Horn's quaternion
  // method: the eigenvector of the largest eigenvalue of a symmetric 4x4)
  ACEZ_HD void align(const double M_end[3][3], double X0, double Y0, double Z0, double X1, double Y1, double Z1, double X2, double Y2, double Z2,
                     double R[3][3], double T[3]) {
    double C_start[3], C_end[3];
#pragma unroll
    for (int i = 0; i < 3; i++) C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
    C_start[0] = (X0 + X1 + X2) / 3;
    C_start[1] = (Y0 + Y1 + Y2) / 3;
    C_start[2] = (Z0 + Z1 + Z2) / 3;
    double s[9];
#pragma unroll
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
    double ev_max = evs[0];
#pragma unroll
    for (int i = 1; i < 4; i++)
      if (evs[i] > ev_max) {
        ev_max = evs[i];
        i_ev = i;
      }
    double q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = pick4(&U[i * 4], i_ev);
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
#pragma unroll
    for (int i = 0; i < 3; i++) T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
  }
};

// ----------------------------------------------------------------------------------------------------
// [upstream] solvePnP(SOLVEPNP_P3P) = solveP3P + take the first solution (solvepnp.cpp)
//   obj: 4 float 3-D points, img: 4 float pixel positions. Returns false when there is no solution.
// p3p::solve sorts its up-to-four solutions by the reprojection error of the 4th point, solveP3P converts each to (rvec, tvec) and
// sorts again by the total reprojection error of the four points, solvePnP keeps the first. Here every solution is finished as
// soon as its depths are known -- pose, 4th-point error, Rodrigues vector, total error -- and the two stable sorts run on the keys
// and a permutation; the arithmetic per solution and the winner are the same.
// ----------------------------------------------------------------------------------------------------
ACEZ_HD inline bool solve_pnp_p3p(const float obj[4][3], const float img[4][2], const Cam& k, Pose* out) {
  // undistortPoints without distortion: output keeps the input depth (float); p3p::extract_points: back to pixels in double
  const double ifx = 1. / k.fx, ify = 1. / k.fy;
  double mu[4], mv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float unx = (float)(((double)img[i][0] - k.cx) * ifx);
    const float uny = (float)(((double)img[i][1] - k.cy) * ify);
    mu[i] = (double)unx * k.fx + k.cx;
    mv[i] = (double)uny * k.fy + k.cy;
  }
  const double X0 = obj[0][0], Y0 = obj[0][1], Z0 = obj[0][2], X1 = obj[1][0], Y1 = obj[1][1], Z1 = obj[1][2];
  const double X2 = obj[2][0], Y2 = obj[2][1], Z2 = obj[2][2], X3 = obj[3][0], Y3 = obj[3][1], Z3 = obj[3][2];
  P3P solver(k);
  // p3p::solve: unit viewing rays of the first three points, the 4th stays in normalised image coordinates
  double mk[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    mu[i] = solver.inv_fx * mu[i] - solver.cx_fx;
    mv[i] = solver.inv_fy * mv[i] - solver.cy_fy;
    const double norm = sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1);
    mk[i] = 1. / norm;
    mu[i] *= mk[i];
    mv[i] *= mk[i];
  }
  mu[3] = solver.inv_fx * mu[3] - solver.cx_fx;
  mv[3] = solver.inv_fy * mv[3] - solver.cy_fy;
  double distances[3];
  distances[0] = sqrt((X1 - X2) * (X1 - X2) + (Y1 - Y2) * (Y1 - Y2) + (Z1 - Z2) * (Z1 - Z2));
  distances[1] = sqrt((X0 - X2) * (X0 - X2) + (Y0 - Y2) * (Y0 - Y2) + (Z0 - Z2) * (Z0 - Z2));
  distances[2] = sqrt((X0 - X1) * (X0 - X1) + (Y0 - Y1) * (Y0 - Y1) + (Z0 - Z1) * (Z0 - Z1));
  double cosines[3];
  cosines[0] = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
  cosines[1] = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
  cosines[2] = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];
  double lenX[4] = {0, 0, 0, 0}, lenY[4] = {0, 0, 0, 0}, lenZ[4] = {0, 0, 0, 0};
  const int n = solver.solve_for_lengths(lenX, lenY, lenZ, distances, cosines);
  if (n == 0) return false;

  double rv[3][4], tv[3][4], err4[4] = {0, 0, 0, 0}, errAll[4] = {0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) rv[a][q] = tv[a][q] = 0;
#pragma unroll 1
  for (int i = 0; i < n; i++) {
    const double l0 = pick4(lenX, i), l1 = pick4(lenY, i), l2 = pick4(lenZ, i);
    double M_orig[3][3];
    M_orig[0][0] = l0 * mu[0]; M_orig[0][1] = l0 * mv[0]; M_orig[0][2] = l0 * mk[0];
    M_orig[1][0] = l1 * mu[1]; M_orig[1][1] = l1 * mv[1]; M_orig[1][2] = l1 * mk[1];
    M_orig[2][0] = l2 * mu[2]; M_orig[2][1] = l2 * mv[2]; M_orig[2][2] = l2 * mk[2];
    double R[3][3], t[3];
    solver.align(M_orig, X0, Y0, Z0, X1, Y1, Z1, X2, Y2, Z2, R, t);
    const double X3p = R[0][0] * X3 + R[0][1] * Y3 + R[0][2] * Z3 + t[0];
    const double Y3p = R[1][0] * X3 + R[1][1] * Y3 + R[1][2] * Z3 + t[1];
    const double Z3p = R[2][0] * X3 + R[2][1] * Y3 + R[2][2] * Z3 + t[2];
    const double mu3p = X3p / Z3p;
    const double mv3p = Y3p / Z3p;
    const double e4 = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
    // solveP3P: rotation matrix -> Rodrigues vector, then the total reprojection error of (rvec, tvec) through projectPoints
    double Rm[9], rvec[3], R2[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) Rm[a * 3 + b] = R[a][b];
    rodrigues_inv(Rm, rvec);
    rodrigues(rvec, R2, nullptr);
    double e = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double u, v;
      project(R2, t, k, obj[j][0], obj[j][1], obj[j][2], &u, &v, nullptr, nullptr, nullptr);
      const double ex = (double)img[j][0] - u, ey = (double)img[j][1] - v;
      e += ex * ex;
      e += ey * ey;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      put4(rv[a], i, rvec[a]);
      put4(tv[a], i, t[a]);
    }
    put4(err4, i, e4);
    put4(errAll, i, e);
  }
  int ord[4] = {0, 1, 2, 3};
  insertion_sort4(err4, ord, n);        // p3p::solve: by the 4th point's error
  double key[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) key[j] = pick4(errAll, ord[j]);
  insertion_sort4(key, ord, n);         // solveP3P: by the total error (stable: ties keep the order above)
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    out->r[a] = pick4(rv[a], ord[0]);
    out->t[a] = pick4(tv[a], ord[0]);
  }
  return true;
}

// ----------------------------------------------------------------------------------------------------
// D4: x = pinv(A) b for symmetric A (6x6) via Jacobi eigen-decomposition, SVBkSb threshold
// ----------------------------------------------------------------------------------------------------
ACEZ_HD inline void solve_sym6(const double A_in[36], const double b[6], double x[6]) {
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
ACEZ_HD inline void solve_normal6(const double A[36], const double b[6], double x[6]) {
  double L[36], Linv[6], y[6];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s = s - L[j * 6 + k] * L[j * 6 + k];
    if (!(s > A[j * 6 + j] * 1e-12)) ok = false;
    const double d = sqrt(ok ? s : 1.0);
    const double inv = 1.0 / d;
    L[j * 6 + j] = d;
    Linv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t = t - L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = t * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double t = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) t = t - L[i * 6 + k] * y[k];
    y[i] = t * Linv[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double t = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) t = t - L[k * 6 + i] * x[k];
    x[i] = t * Linv[i];
  }
  if (!ok) solve_sym6(A, b, x);
}

}  // namespace rsm
