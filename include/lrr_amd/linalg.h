// linalg.h — the two small dense factorizations the ICP host code needs, standing in for
// vnl_svd<double> (absent): inverse of the symmetric dof x dof normal matrix
// (cdcl/cdcl_estimation_ICP_matching_all.txx:1073-1074) and the 3x3 SVD behind
// cdcl_trans::inverse() / recompose_increment() (cdcl/cdcl_trans_rigid3d.cxx:115, 318).
#ifndef LRR_AMD_LINALG_H_
#define LRR_AMD_LINALG_H_

#include <cmath>
#include <cstring>

// the same source also serves the device side of the batched location loop (lrr_icp_batch.inc: the solve of every
// location without a host round trip): under hipcc the routines compile for both sides, and the variants ending in _ws
// take their work arrays from the caller (LDS on the device; the plain names keep them on the stack)
#if defined(__HIPCC__)
#define LRR_LINALG_HD __host__ __device__
#else
#define LRR_LINALG_HD
#endif

namespace lrr_amd {
namespace linalg {

// inverse of a symmetric positive definite n x n matrix (n <= 12) by Cholesky factorisation:
// A = L L^T, A^-1 = L^-T L^-1.  Returns false — and leaves `out` untouched — when a pivot shows the
// matrix to be indefinite, singular or close to it (pivot below 1e-10 of its diagonal entry).
LRR_LINALG_HD inline bool chol_inverse_ws(int n, const double *M, double *out, double *L, double *Li) {
  for (int j = 0; j < n; ++j) {
    const double ajj = 0.5 * (M[j * n + j] + M[j * n + j]);
    double d = ajj;
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (!(ajj > 0.0) || !(d > 1e-10 * ajj)) return false;
    const double ljj = std::sqrt(d);
    L[j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double v = 0.5 * (M[i * n + j] + M[j * n + i]);
      for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = v / ljj;
    }
  }
  for (int j = 0; j < n; ++j) {  // Li = L^-1, column by column (lower triangular)
    Li[j * n + j] = 1.0 / L[j * n + j];
    for (int i = j + 1; i < n; ++i) {
      double v = 0.0;
      for (int k = j; k < i; ++k) v -= L[i * n + k] * Li[k * n + j];
      Li[i * n + j] = v / L[i * n + i];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) {
      double v = 0.0;
      for (int k = j; k < n; ++k) v += Li[k * n + i] * Li[k * n + j];
      if (!std::isfinite(v)) return false;
      out[i * n + j] = v;
      out[j * n + i] = v;
    }
  return true;
}
inline bool chol_inverse(int n, const double *M, double *out) {
  double L[144], Li[144];
  return chol_inverse_ws(n, M, out, L, Li);
}

// inverse of the symmetric n x n normal matrix (n <= 12), standing in for vnl_svd::inverse().
// The well-conditioned positive definite case — every ICP iteration with enough matches — takes the
// Cholesky route (2 us for n = 12); anything else goes through a cyclic-Jacobi eigen-decomposition
// (27 us) where zero eigenvalues are left out, which is what vnl_svd::inverse() does with zero
// singular values.
LRR_LINALG_HD inline void sym_inverse_ws(int n, const double *M, double *out, double *A, double *V) {
  if (chol_inverse_ws(n, M, out, A, V)) return;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      A[i * n + j] = 0.5 * (M[i * n + j] + M[j * n + i]);
      V[i * n + j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, dg = 0.0;
    for (int i = 0; i < n; ++i) {
      dg += A[i * n + i] * A[i * n + i];
      for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    }
    if (off == 0.0 || off <= 1e-60 * dg) break;
    for (int p = 0; p + 1 < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = A[p * n + q];
        if (apq == 0.0) continue;
        double th = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        double t = (th >= 0.0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          double x = A[k * n + p], y = A[k * n + q];
          A[k * n + p] = c * x - s * y;
          A[k * n + q] = s * x + c * y;
        }
        for (int k = 0; k < n; ++k) {
          double x = A[p * n + k], y = A[q * n + k];
          A[p * n + k] = c * x - s * y;
          A[q * n + k] = s * x + c * y;
        }
        for (int k = 0; k < n; ++k) {
          double x = V[k * n + p], y = V[k * n + q];
          V[k * n + p] = c * x - s * y;
          V[k * n + q] = s * x + c * y;
        }
      }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int k = 0; k < n; ++k) {
        double lam = A[k * n + k];
        if (lam != 0.0) s += V[i * n + k] * V[j * n + k] / lam;
      }
      out[i * n + j] = s;
    }
}
inline void sym_inverse(int n, const double *M, double *out) {
  double A[144], V[144];
  sym_inverse_ws(n, M, out, A, V);
}

// one-sided Jacobi SVD of a 3x3 (row-major): M = U diag(s) V^T
LRR_LINALG_HD inline void svd3(const double *M, double *U, double *s, double *V) {
  double W[9];
  for (int i = 0; i < 9; ++i) W[i] = M[i];
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double a = 0, b = 0, g = 0;
        for (int k = 0; k < 3; ++k) {
          a += W[k * 3 + p] * W[k * 3 + p];
          b += W[k * 3 + q] * W[k * 3 + q];
          g += W[k * 3 + p] * W[k * 3 + q];
        }
        if (g == 0.0 || std::fabs(g) <= 1e-17 * std::sqrt(a * b)) continue;
        rotated = true;
        double z = (b - a) / (2.0 * g);
        double t = (z >= 0.0 ? 1.0 : -1.0) / (std::fabs(z) + std::sqrt(1.0 + z * z));
        double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int k = 0; k < 3; ++k) {
          double x = W[k * 3 + p], y = W[k * 3 + q];
          W[k * 3 + p] = c * x - sn * y;
          W[k * 3 + q] = sn * x + c * y;
          x = V[k * 3 + p]; y = V[k * 3 + q];
          V[k * 3 + p] = c * x - sn * y;
          V[k * 3 + q] = sn * x + c * y;
        }
      }
    if (!rotated) break;
  }
  for (int j = 0; j < 3; ++j) {
    double nrm = std::sqrt(W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j]);
    s[j] = nrm;
    for (int k = 0; k < 3; ++k) U[k * 3 + j] = nrm > 0.0 ? W[k * 3 + j] / nrm : 0.0;
  }
}

LRR_LINALG_HD inline void inverse3(const double *M, double *out) {
  double U[9], s[3], V[9];
  svd3(M, U, s, V);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
      for (int k = 0; k < 3; ++k)
        if (s[k] != 0.0) a += V[i * 3 + k] * U[j * 3 + k] / s[k];
      out[i * 3 + j] = a;
    }
}

}  // namespace linalg
}  // namespace lrr_amd
#endif
