/*
 * lrr_oracle.c — CPU restatement of LRR's robust-ICP estimation inner loop.
 *
 * TEST INFRASTRUCTURE ONLY (see lrr_oracle.h).  Plain C11, double precision,
 * no dependencies.  Every function cites the reference file:line it follows
 * (paths relative to the reference repo root, msofka/LRR).
 *
 * PARITY UNPINNED.  The reference has no tests/golden vectors and cannot be
 * built in this image (VXL, ITK absent).  Three pieces of arithmetic live in
 * un-vendored, un-versioned VXL (circa 1.x, 2008) and are restated from their
 * published algorithms:
 *   - rsdl_kd_tree::n_nearest(k=1)  -> exact nearest neighbour, ties -> lowest index
 *   - rrel_muset_obj::scale         -> MUSE (Miller & Stewart, CVPR'96), trimmed form
 *   - vnl_svd<double>::inverse/U/V  -> Jacobi eigen / one-sided Jacobi SVD
 *
 * Documented deviations from the reference (all invisible at 1e-10):
 *   - std::random_shuffle(rand()) -> a fresh seeded keyed permutation per call (lrr_oracle_shuffle);
 *     the reference's shuffles are cumulative, which no caller can observe
 *   - std::set<smart_ptr> iteration (heap-address order) -> ascending feature index
 *   - the unused indices_vector shuffles in the symmetric estimate_LS are skipped
 *     (they only burn rand() state in the reference)
 *   - rrl Voronoi-volume lookup (1 mm quantised) -> exact NN
 *
 * Floating-point contract shared with the HIP path (so per-pair quantities are
 * bit-identical and only reduction ORDER differs): compiled with
 * -ffp-contract=off; the only fused operations are the explicit fma() calls in
 * dist2().
 */
#define _DEFAULT_SOURCE
#include "lrr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXDOF 12

/* ------------------------------------------------------------------------- */
/* small linear algebra                                                      */
/* ------------------------------------------------------------------------- */

/* Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 12).
 * A is destroyed; V columns are eigenvectors. */
static void jacobi_eig_sym(int n, double *A, double *V, double *lam) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += A[i * n + i] * A[i * n + i];
      for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    }
    if (off == 0.0 || off <= 1e-60 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = A[p * n + q];
        if (apq == 0.0) continue;
        double app = A[p * n + p], aqq = A[q * n + q];
        double theta = (aqq - app) / (2.0 * apq);
        double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) { /* columns p,q */
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) { /* rows p,q */
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; ++i) lam[i] = A[i * n + i];
}

/* vnl_svd<double>(M).inverse() for the (symmetric) normal matrix
 * (cdcl/cdcl_estimation_ICP_matching_all.txx:1073-1074).  vnl_svd with the
 * default zero tolerance inverts every non-zero singular value; for a symmetric
 * matrix U S V^T = V |L| sign(L) V^T, so the inverse is V diag(1/lambda) V^T. */
void lrr_oracle_sym_pinv(int n, const double *M, double *out) {
  double A[MAXDOF * MAXDOF], V[MAXDOF * MAXDOF], lam[MAXDOF];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) A[i * n + j] = 0.5 * (M[i * n + j] + M[j * n + i]);
  jacobi_eig_sym(n, A, V, lam);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int k = 0; k < n; ++k)
        if (lam[k] != 0.0) s += V[i * n + k] * V[j * n + k] / lam[k];
      out[i * n + j] = s;
    }
}

/* One-sided (Hestenes) Jacobi SVD of a 3x3 matrix: M = U diag(s) V^T.
 * Stands in for vnl_svd<double> on 3x3 inputs (cdcl/cdcl_trans_rigid3d.cxx:115,318;
 * cdcl/cdcl_trans_affine.txx:56). */
void lrr_oracle_svd3(const double *M, double *U, double *s, double *V) {
  double W[9];
  memcpy(W, M, sizeof W);
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double a = 0, b = 0, g = 0;
        for (int k = 0; k < 3; ++k) {
          a += W[k * 3 + p] * W[k * 3 + p];
          b += W[k * 3 + q] * W[k * 3 + q];
          g += W[k * 3 + p] * W[k * 3 + q];
        }
        if (g == 0.0 || fabs(g) <= 1e-17 * sqrt(a * b)) continue;
        rotated = 1;
        double zeta = (b - a) / (2.0 * g);
        double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int k = 0; k < 3; ++k) {
          double wp = W[k * 3 + p], wq = W[k * 3 + q];
          W[k * 3 + p] = c * wp - sn * wq;
          W[k * 3 + q] = sn * wp + c * wq;
          double vp = V[k * 3 + p], vq = V[k * 3 + q];
          V[k * 3 + p] = c * vp - sn * vq;
          V[k * 3 + q] = sn * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  for (int j = 0; j < 3; ++j) {
    double nrm = sqrt(W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j]);
    s[j] = nrm;
    for (int k = 0; k < 3; ++k) U[k * 3 + j] = (nrm > 0.0) ? W[k * 3 + j] / nrm : 0.0;
  }
}

/* vnl_svd<double>(M).inverse() for 3x3 M. */
static void inv3_svd(const double *M, double *out) {
  double U[9], s[3], V[9];
  lrr_oracle_svd3(M, U, s, V);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
      for (int k = 0; k < 3; ++k)
        if (s[k] != 0.0) a += V[i * 3 + k] * U[j * 3 + k] / s[k];
      out[i * 3 + j] = a;
    }
}

static void mat3_mul(const double *X, const double *Y, double *Z) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      T[i * 3 + j] = (X[i * 3] * Y[j] + X[i * 3 + 1] * Y[3 + j]) + X[i * 3 + 2] * Y[6 + j];
  memcpy(Z, T, sizeof T);
}

static void mat3_vec(const double *X, const double *v, double *o) {
  double r[3];
  for (int i = 0; i < 3; ++i) r[i] = (X[i * 3] * v[0] + X[i * 3 + 1] * v[1]) + X[i * 3 + 2] * v[2];
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}

/* ------------------------------------------------------------------------- */
/* transforms: cdcl_trans_rigid3d (dof 6) and cdcl_trans_affine<3,12>        */
/* ------------------------------------------------------------------------- */

typedef struct {
  int dof;           /* 6 rigid3d, 12 affine */
  double A[9];       /* R_ / A_ */
  double t[3];       /* t_ */
  double c[3];       /* center_moving_ (cdcl/cdcl_trans.h:147) */
  double covar[MAXDOF * MAXDOF];
  double covarJ[9];
  double abg[3];     /* alpha_, beta_, gamma_ (rigid only) */
} trans;

/* default constructors: identity (cdcl_trans_rigid3d.cxx:15-27, cdcl_trans_affine.txx:18-25, cdcl_trans.txx:14-24) */
static void trans_identity(trans *T, int dof) {
  memset(T, 0, sizeof *T);
  T->dof = dof;
  T->A[0] = T->A[4] = T->A[8] = 1.0;
}

/* map_loc: R (x - c) + t  (cdcl_trans_rigid3d.cxx:171-174, cdcl_trans_affine.txx:112-115).
 * Canonical operation order shared with the HIP kernels. */
static inline void trans_map(const trans *T, const double *x, double *y) {
  double d0 = x[0] - T->c[0], d1 = x[1] - T->c[1], d2 = x[2] - T->c[2];
  y[0] = ((T->A[0] * d0 + T->A[1] * d1) + T->A[2] * d2) + T->t[0];
  y[1] = ((T->A[3] * d0 + T->A[4] * d1) + T->A[5] * d2) + T->t[1];
  y[2] = ((T->A[6] * d0 + T->A[7] * d1) + T->A[8] * d2) + T->t[2];
}

/* jacobian_wrt_par (rigid: cdcl_trans_rigid3d.cxx:197-225; affine: cdcl_trans_affine.txx:141-161).
 * J is 3 x dof row-major. */
static void trans_jac(const trans *T, const double *x, double *J) {
  double l0 = x[0] - T->c[0], l1 = x[1] - T->c[1], l2 = x[2] - T->c[2];
  int dof = T->dof;
  memset(J, 0, sizeof(double) * 3 * dof);
  if (dof == 6) {
    J[1 * 6 + 0] = -l2; J[2 * 6 + 0] = l1;
    J[0 * 6 + 1] = l2;  J[2 * 6 + 1] = -l0;
    J[0 * 6 + 2] = -l1; J[1 * 6 + 2] = l0;
    J[0 * 6 + 3] = 1.0; J[1 * 6 + 4] = 1.0; J[2 * 6 + 5] = 1.0;
  } else {
    for (int d = 0; d < 3; ++d) {
      J[d * 12 + 3 * d + 0] = l0;
      J[d * 12 + 3 * d + 1] = l1;
      J[d * 12 + 3 * d + 2] = l2;
      J[d * 12 + 9 + d] = 1.0;
    }
  }
}

/* set_parameterization (rigid: cdcl_trans_rigid3d.cxx:140-166, small-angle R = I + [w]x;
 * affine: cdcl_trans_affine.txx:89-106, row-major A then t). */
static void trans_set_param(trans *T, const double *p) {
  if (T->dof == 6) {
    T->abg[0] = p[0]; T->abg[1] = p[1]; T->abg[2] = p[2];
    double da = p[0], db = p[1], dg = p[2];
    T->A[0] = 1.0; T->A[1] = -dg; T->A[2] = db;
    T->A[3] = dg;  T->A[4] = 1.0; T->A[5] = -da;
    T->A[6] = -db; T->A[7] = da;  T->A[8] = 1.0;
    T->t[0] = p[3]; T->t[1] = p[4]; T->t[2] = p[5];
  } else {
    for (int i = 0; i < 9; ++i) T->A[i] = p[i];
    for (int i = 0; i < 3; ++i) T->t[i] = p[9 + i];
  }
}

/* normalize (rigid: cdcl_trans_rigid3d.cxx:260-280; affine: cdcl_trans_affine.txx:220-253) */
static void trans_normalize(trans *T, double rm, double rf, const double *cm, const double *cf) {
  double f = rm / rf;
  for (int i = 0; i < 9; ++i) T->A[i] = f * T->A[i];
  double d[3] = { cm[0] - T->c[0], cm[1] - T->c[1], cm[2] - T->c[2] };
  double Ad[3];
  mat3_vec(T->A, d, Ad); /* uses the already scaled A, as the reference does */
  for (int i = 0; i < 3; ++i) T->t[i] = (T->t[i] / rf + Ad[i] / rm) - cf[i] / rf;
  if (T->dof == 12) {
    double B[144], tmp[144], out[144];
    memset(B, 0, sizeof B);
    for (int i = 0; i < 9; ++i) B[i * 12 + i] = f;
    for (int d1 = 0; d1 < 3; ++d1)
      for (int d2 = 0; d2 < 3; ++d2) B[(9 + d1) * 12 + d1 * 3 + d2] = d[d2] / rf;
    for (int i = 9; i < 12; ++i) B[i * 12 + i] = 1.0 / rf;
    for (int i = 0; i < 12; ++i)
      for (int j = 0; j < 12; ++j) {
        double s = 0.0;
        for (int k = 0; k < 12; ++k) s += B[i * 12 + k] * T->covar[k * 12 + j];
        tmp[i * 12 + j] = s;
      }
    for (int i = 0; i < 12; ++i)
      for (int j = 0; j < 12; ++j) {
        double s = 0.0;
        for (int k = 0; k < 12; ++k) s += tmp[i * 12 + k] * B[j * 12 + k];
        out[i * 12 + j] = s;
      }
    memcpy(T->covar, out, sizeof out);
    T->c[0] = T->c[1] = T->c[2] = 0.0;
  }
  /* rigid: covariance and centre untouched (cdcl_trans_rigid3d.cxx:271-279) */
}

/* unnormalize (rigid: cdcl_trans_rigid3d.cxx:284-305; affine: cdcl_trans_affine.txx:259-286) */
static void trans_unnormalize(trans *T, double rm, double rf, const double *cm, const double *cf) {
  double f = rf / rm;
  for (int i = 0; i < 9; ++i) T->A[i] = f * T->A[i];
  if (T->dof == 6) {
    double Ac[3];
    mat3_vec(T->A, cm, Ac);
    for (int i = 0; i < 3; ++i) T->t[i] = (-Ac[i] + rf * T->t[i]) + cf[i];
  } else {
    for (int i = 0; i < 3; ++i) T->t[i] = rf * T->t[i] + cf[i];
    double Bd[12];
    for (int i = 0; i < 9; ++i) Bd[i] = f;
    for (int i = 9; i < 12; ++i) Bd[i] = rf;
    for (int i = 0; i < 12; ++i)
      for (int j = 0; j < 12; ++j) T->covar[i * 12 + j] = Bd[i] * T->covar[i * 12 + j] * Bd[j];
  }
  T->c[0] = cm[0]; T->c[1] = cm[1]; T->c[2] = cm[2];
}

/* ctor from (R, t, centre): recovers Euler angles for the rigid type
 * (cdcl_trans_rigid3d.cxx:83-108); covariance starts at zero. */
static void trans_from_Atc(trans *T, int dof, const double *A, const double *t, const double *c) {
  trans_identity(T, dof);
  memcpy(T->A, A, sizeof T->A);
  memcpy(T->t, t, sizeof T->t);
  if (c) memcpy(T->c, c, sizeof T->c);
  if (dof == 6) {
    double beta = asin(A[2]);
    double C = cos(beta);
    T->abg[1] = beta;
    if (fabs(C) > 1e-6) {
      T->abg[0] = atan2(-A[5] / C, A[8] / C);
      T->abg[2] = atan2(-A[1] / C, A[0] / C);
    } else {
      T->abg[0] = 0.0;
      T->abg[2] = atan2(A[3], A[4]);
    }
  }
}

/* inverse(): (A^-1, -A^-1 t + c, centre 0)  (cdcl_trans_rigid3d.cxx:113-118, cdcl_trans_affine.txx:54-59).
 * The new object's covariance is zero (fresh ctor). */
static void trans_inverse(const trans *T, trans *out) {
  double Ai[9], v[3], t2[3];
  inv3_svd(T->A, Ai);
  mat3_vec(Ai, T->t, v);
  for (int i = 0; i < 3; ++i) t2[i] = -v[i] + T->c[i];
  double zero[3] = { 0, 0, 0 };
  trans_from_Atc(out, T->dof, Ai, t2, zero);
}

/* recompose_increment (cdcl_trans_rigid3d.cxx:309-325): dR <- U V^T; R <- R dR; t <- dR t + dt */
static void trans_recompose_increment(trans *G, const trans *inc) {
  double U[9], s[3], V[9], dR[9], Vt[9];
  lrr_oracle_svd3(inc->A, U, s, V);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Vt[i * 3 + j] = V[j * 3 + i];
  mat3_mul(U, Vt, dR);
  double R2[9], t2[3];
  mat3_mul(G->A, dR, R2);
  mat3_vec(dR, G->t, t2);
  memcpy(G->A, R2, sizeof R2);
  for (int i = 0; i < 3; ++i) G->t[i] = t2[i] + inc->t[i];
}

/* ------------------------------------------------------------------------- */
/* robust statistics                                                         */
/* ------------------------------------------------------------------------- */

/* weight_BT (cdcl/cdcl_utils.h:142-152): residual already squared and scaled */
double lrr_oracle_weight_BT(double u) {
  if (u <= 16.0) {
    double temp = 1 - u * 0.0625;
    return temp * temp;
  }
  return 0;
}

static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* k-th smallest (0-based), quickselect; permutes v (std::nth_element stand-in) */
static double kth_smallest(double *v, int64_t n, int64_t k) {
  int64_t lo = 0, hi = n - 1;
  while (lo < hi) {
    double pivot = v[lo + (hi - lo) / 2];
    int64_t i = lo, j = hi;
    while (i <= j) {
      while (v[i] < pivot) ++i;
      while (v[j] > pivot) --j;
      if (i <= j) { double tmp = v[i]; v[i] = v[j]; v[j] = tmp; ++i; --j; }
    }
    if (k <= j) hi = j;
    else if (k >= i) lo = i;
    else break;
  }
  return v[k];
}

/* rrel_util_median_abs_dev_scale with dof = 1
 * (cdcl/cdcl_estimation_ICP_matching_all.txx:520-536, rrl/rrl_estimation_ICP_matching_all.txx:473-489) */
double lrr_oracle_mad_scale(double *v, int64_t n) {
  const int dof = 1;
  if (n <= dof) return 0;
  for (int64_t i = 0; i < n; ++i) v[i] = fabs(v[i]);
  int64_t loc = (n - dof) / 2 + dof;
  double kth = kth_smallest(v, n, loc);
  return 1.4826 * (1 + 5.0 / (double)(n - dof)) * kth;
}

/* inverse standard normal CDF (Acklam's rational approximation + one Halley step) */
static double norm_inv_cdf(double p) {
  static const double a[] = { -3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                              1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00 };
  static const double b[] = { -5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                              6.680131188771972e+01, -1.328068155288572e+01 };
  static const double c[] = { -7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                              -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00 };
  static const double d[] = { 7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                              3.754408661907416e+00 };
  double x;
  if (p < 0.02425) {
    double q = sqrt(-2 * log(p));
    x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
        ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
  } else if (p <= 1 - 0.02425) {
    double q = p - 0.5, r = q * q;
    x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
        (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
  } else {
    double q = sqrt(-2 * log(1 - p));
    x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
        ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
  }
  double e = 0.5 * erfc(-x / sqrt(2.0)) - p;
  double u = e * sqrt(2.0 * M_PI) * exp(x * x / 2.0);
  x = x - u / (1.0 + x * u / 2.0);
  return x;
}

/* MUSE order-statistic tables, computed on the fly (table size 0 in the reference call,
 * cdcl/cdcl_estimation_ICP_matching_all.txx:721-725).  Restated from the published VXL
 * rrel_muse_table: first-order expected k-th order statistic of |N(0,1)| and the
 * David-Johnson variance expansion.  NOT reference-verified (VXL absent). */
static double muse_expected_kth(int64_t k, int64_t n) {
  double pk = (double)k / (double)(n + 1);
  return norm_inv_cdf((1.0 + pk) / 2.0);
}
static double muse_stddev_kth(int64_t k, int64_t n) {
  double pk = (double)k / (double)(n + 1), qk = 1.0 - pk;
  double Qk = norm_inv_cdf((1.0 + pk) / 2.0);
  double pQk = sqrt(2.0 / M_PI) * exp(-Qk * Qk / 2.0);
  double Qp = 1.0 / pQk;
  double Qpp = Qk / (pQk * pQk);
  double Qppp = (1.0 + 2.0 * Qk * Qk) / (pQk * pQk * pQk);
  double np2 = (double)n + 2.0;
  double vrk = (pk * qk / np2) * Qp * Qp +
               (pk * qk / (np2 * np2)) *
                   (2.0 * (qk - pk) * Qp * Qpp + pk * qk * (Qp * Qppp + Qpp * Qpp / 2.0));
  return sqrt(vrk);
}
static double muse_divisor(int64_t k, int64_t n) {
  double ek = muse_expected_kth(k, n);
  return (double)(n + 1) * sqrt(2.0 / M_PI) * (1.0 - exp(-ek * ek / 2.0));
}

/* rrel_muset_obj(0, false).scale(begin, end): trimmed MUSE, inlier fractions
 * 0.25..0.95 step 0.05, no s_k refinement
 * (call sites cdcl/cdcl_estimation_ICP_matching_all.txx:721-725, rrl/...:630-634). */
double lrr_oracle_muse_scale(const double *res, int64_t n) {
  if (n <= 0) return 0.0;
  double *a = (double *)malloc(sizeof(double) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) a[i] = fabs(res[i]);
  qsort(a, (size_t)n, sizeof(double), cmp_double);
  const double min_frac = 0.25, max_frac = 0.95, frac_inc = 0.05;
  const double min_exp_kth_to_stddev_ratio = 3.0;
  int at_start = 1;
  double best_sk = 0.0, best_obj = 0.0, sum = 0.0;
  int64_t prev_k = 0;
  for (double frac = min_frac; frac <= max_frac + 0.00001; frac += frac_inc) {
    int64_t k = (int64_t)floor(frac * (double)n + 0.5); /* vnl_math_rnd */
    if (k > n) k = n;
    if (k <= 0) k = 1;
    double ek = muse_expected_kth(k, n), sd = muse_stddev_kth(k, n);
    if (ek / sd < min_exp_kth_to_stddev_ratio) continue;
    for (int64_t i = prev_k; i < k; ++i) sum += a[i];
    double sk = sum / muse_divisor(k, n);
    double obj = sk * sd / ek;
    if (at_start || obj < best_obj) {
      at_start = 0;
      best_sk = sk;
      best_obj = obj;
    }
    prev_k = k;
  }
  free(a);
  return best_sk;
}

/* Deterministic stand-in for std::random_shuffle (unseeded rand() in the reference,
 * cdcl/cdcl_estimation_ICP_matching_all.txx:557).  Each call draws a fresh keyed pseudo-random
 * PERMUTATION of [0, n) that is computable per element (so the HIP match kernel evaluates it
 * in-register): a 4-round Feistel network over the next even power of two, cycle-walked into
 * [0, n).  Keys come from a splitmix64 stream.  Shared, bit for bit, with the product
 * (include/lrr_icp.h: lrr_icp_perm_keys / the device function subset_perm). */
static uint64_t splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
void lrr_oracle_perm_keys(uint64_t *state, uint32_t keys[4]) {
  for (int r = 0; r < 4; ++r) keys[r] = (uint32_t)(splitmix64(state) >> 32);
}
int64_t lrr_oracle_subset_perm(int64_t i, int64_t n, const uint32_t keys[4]) {
  int half = 1;
  while ((1LL << (2 * half)) < n) ++half;
  uint32_t mask = (uint32_t)((1u << half) - 1u);
  uint64_t x = (uint64_t)i;
  do {
    uint32_t L = (uint32_t)(x >> half), R = (uint32_t)(x & mask);
    for (int r = 0; r < 4; ++r) {
      uint32_t F = fmix32(R ^ keys[r]) & mask;
      uint32_t nL = R;
      R = L ^ F;
      L = nL;
    }
    x = ((uint64_t)L << half) | R;
  } while (x >= (uint64_t)n);
  return (int64_t)x;
}
void lrr_oracle_shuffle(int64_t *order, int64_t n, uint64_t *state) {
  uint32_t keys[4];
  lrr_oracle_perm_keys(state, keys);
  for (int64_t i = 0; i < n; ++i) order[i] = lrr_oracle_subset_perm(i, n, keys);
}

/* ------------------------------------------------------------------------- */
/* exact nearest neighbour: brute force and kd-tree                          */
/* (rsdl_kd_tree::n_nearest(query, 1, ...) stand-in, call site               */
/*  cdcl/cdcl_estimation_ICP_matching_all.txx:572)                           */
/* ------------------------------------------------------------------------- */

/* canonical squared distance: two explicit fused multiply-adds */
static inline double dist2(const double *a, const double *b) {
  double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return fma(dz, dz, fma(dy, dy, dx * dx));
}

static int64_t nn_brute(const double *loc, int64_t n, const double *q) {
  int64_t best = -1;
  double bd = INFINITY;
  for (int64_t j = 0; j < n; ++j) {
    double d = dist2(q, loc + 3 * j);
    if (d < bd) { bd = d; best = j; }
  }
  return best;
}

typedef struct {
  int64_t n;
  const double *loc;   /* borrowed, n x 3 (never the in-place-normalised state: built once) */
  double *pts;         /* private copy of locations, kd order */
  int64_t *idx;        /* original index, kd order */
  int32_t *split_dim;  /* per node */
  double *split_val;
  int64_t *lo, *hi;    /* per node point range */
  int64_t *left, *right;
  int64_t nnodes, cap;
} kdtree;

#define KD_LEAF 8

static int64_t kd_build_rec(kdtree *t, int64_t lo, int64_t hi) {
  int64_t node = t->nnodes++;
  t->lo[node] = lo; t->hi[node] = hi; t->left[node] = t->right[node] = -1;
  t->split_dim[node] = -1;
  if (hi - lo <= KD_LEAF) return node;
  double mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (int64_t i = lo; i < hi; ++i)
    for (int d = 0; d < 3; ++d) {
      double v = t->pts[3 * i + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  int dim = 0;
  if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
  if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
  if (mx[dim] == mn[dim]) return node; /* all identical: keep as leaf */
  /* median split by quickselect on dim */
  int64_t mid = lo + (hi - lo) / 2, l = lo, h = hi - 1;
  while (l < h) {
    double pivot = t->pts[3 * (l + (h - l) / 2) + dim];
    int64_t i = l, j = h;
    while (i <= j) {
      while (t->pts[3 * i + dim] < pivot) ++i;
      while (t->pts[3 * j + dim] > pivot) --j;
      if (i <= j) {
        for (int d = 0; d < 3; ++d) { double tmp = t->pts[3 * i + d]; t->pts[3 * i + d] = t->pts[3 * j + d]; t->pts[3 * j + d] = tmp; }
        int64_t ti = t->idx[i]; t->idx[i] = t->idx[j]; t->idx[j] = ti;
        ++i; --j;
      }
    }
    if (mid <= j) h = j;
    else if (mid >= i) l = i;
    else break;
  }
  t->split_dim[node] = dim;
  t->split_val[node] = t->pts[3 * mid + dim];
  int64_t L = kd_build_rec(t, lo, mid);
  int64_t R = kd_build_rec(t, mid, hi);
  t->left[node] = L; t->right[node] = R;
  return node;
}

static kdtree *kd_build(const double *loc, int64_t n) {
  kdtree *t = (kdtree *)calloc(1, sizeof *t);
  t->n = n; t->loc = loc;
  t->pts = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
  t->idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  memcpy(t->pts, loc, sizeof(double) * 3 * (size_t)n);
  for (int64_t i = 0; i < n; ++i) t->idx[i] = i;
  t->cap = 2 * (n / (KD_LEAF / 2) + 2) + 8;
  t->split_dim = (int32_t *)malloc(sizeof(int32_t) * (size_t)t->cap);
  t->split_val = (double *)malloc(sizeof(double) * (size_t)t->cap);
  t->lo = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->cap);
  t->hi = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->cap);
  t->left = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->cap);
  t->right = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->cap);
  t->nnodes = 0;
  if (n > 0) kd_build_rec(t, 0, n);
  return t;
}

static void kd_free(kdtree *t) {
  if (!t) return;
  free(t->pts); free(t->idx); free(t->split_dim); free(t->split_val);
  free(t->lo); free(t->hi); free(t->left); free(t->right); free(t);
}

/* exact search; a far subtree is pruned only when it cannot hold a strictly closer
 * point nor an equally close point with a lower index (plane distance > best). */
static void kd_search(const kdtree *t, int64_t node, const double *q, double *bd, int64_t *bi) {
  if (t->split_dim[node] < 0) {
    for (int64_t i = t->lo[node]; i < t->hi[node]; ++i) {
      double d = dist2(q, t->pts + 3 * i);
      if (d < *bd || (d == *bd && t->idx[i] < *bi)) { *bd = d; *bi = t->idx[i]; }
    }
    return;
  }
  int dim = t->split_dim[node];
  double diff = q[dim] - t->split_val[node];
  int64_t near = diff < 0 ? t->left[node] : t->right[node];
  int64_t far = diff < 0 ? t->right[node] : t->left[node];
  kd_search(t, near, q, bd, bi);
  if (diff * diff <= *bd) kd_search(t, far, q, bd, bi);
}

static int64_t nn_kd(const kdtree *t, const double *q) {
  double bd = INFINITY;
  int64_t bi = -1;
  if (t->n > 0) kd_search(t, 0, q, &bd, &bi);
  return bi;
}

/* ------------------------------------------------------------------------- */
/* Voronoi index volume of the production rrl matcher                         */
/* ------------------------------------------------------------------------- */
/* rrl/examples/compute_distance_map.cxx:55-151 builds, per feature set, a 1 mm volume whose voxels hold
 * the index of the nearest feature: origin = minimum corner of the features' bounding box, size[d] =
 * ceil(extent[d]) (:104), every feature i writes its index into the voxel its location rounds to (:122-131:
 * a later feature overwrites an earlier one of the same voxel, and index 0 is the background value of the
 * distance filter, so feature 0 is never a site), and itk::DanielssonDistanceMapImageFilter's Voronoi map
 * gives every voxel the value of its nearest occupied voxel (:136-150).  The matcher rounds the mapped point
 * to a voxel, clamps it to the volume and reads the index (rrl/rrl_estimation_ICP_matching_all.txx:530-543).
 *
 * Restated with two choices the reference leaves to ITK (not in this image, version unpinned): physical point
 * -> index rounds half up (floor(x + 0.5), ITK >= 3.16 with centred pixel coordinates), and the nearest
 * occupied voxel is EXACT with ties to the lowest feature index, where Danielsson's vector propagation is an
 * approximation with an order-dependent tie rule.  A feature whose voxel falls outside the volume (the
 * reference writes out of bounds there) is clamped into it.  PARITY UNPINNED on both counts.
 * The volume is filled lazily: a voxel is resolved (exact nearest site through a kd-tree over the occupied
 * voxel centres) the first time it is looked up; afterwards a lookup is O(1), as in the reference. */
typedef struct {
  double origin[3];
  int64_t size[3];
  int64_t nsites;
  double *site_loc;    /* voxel centres of the occupied voxels, ascending label order */
  int64_t *site_label; /* the feature index each carries */
  kdtree *kd;          /* over site_loc: ties -> lowest site number = lowest label */
  int32_t *vol;        /* -1 = not resolved yet */
} voronoi_map;

static inline int64_t vor_index(const voronoi_map *m, int d, double x) {
  double v = floor((x - m->origin[d]) + 0.5);
  if (v < 0.0) v = 0.0;
  if (v > (double)(m->size[d] - 1)) v = (double)(m->size[d] - 1);
  return (int64_t)v;
}

static int cmp_i64_pair(const void *a, const void *b) {
  const int64_t *x = (const int64_t *)a, *y = (const int64_t *)b;
  if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
  return x[1] < y[1] ? -1 : (x[1] > y[1] ? 1 : 0);
}

static voronoi_map *vor_build(const double *loc, int64_t n) {
  voronoi_map *m = (voronoi_map *)calloc(1, sizeof *m);
  double mx[3] = { 0, 0, 0 };
  for (int d = 0; d < 3; ++d) { m->origin[d] = n > 0 ? loc[d] : 0.0; mx[d] = m->origin[d]; }
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      if (loc[3 * i + d] < m->origin[d]) m->origin[d] = loc[3 * i + d];
      if (loc[3 * i + d] > mx[d]) mx[d] = loc[3 * i + d];
    }
  for (int d = 0; d < 3; ++d) {
    m->size[d] = (int64_t)ceil(mx[d] - m->origin[d]);
    if (m->size[d] < 1) m->size[d] = 1;
  }
  /* (voxel, feature) pairs sorted: the last of a voxel's run is the feature that wrote it last */
  int64_t *pairs = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)(n > 0 ? n : 1));
  int64_t np = 0;
  for (int64_t i = 1; i < n; ++i) { /* feature 0 = background value */
    int64_t ix = vor_index(m, 0, loc[3 * i]), iy = vor_index(m, 1, loc[3 * i + 1]), iz = vor_index(m, 2, loc[3 * i + 2]);
    pairs[2 * np] = (iz * m->size[1] + iy) * m->size[0] + ix;
    pairs[2 * np + 1] = i;
    ++np;
  }
  qsort(pairs, (size_t)np, 2 * sizeof(int64_t), cmp_i64_pair);
  int64_t *lab = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)(np > 0 ? np : 1));
  int64_t ns = 0;
  for (int64_t k = 0; k < np; ++k)
    if (k + 1 == np || pairs[2 * (k + 1)] != pairs[2 * k]) { lab[2 * ns] = pairs[2 * k + 1]; lab[2 * ns + 1] = pairs[2 * k]; ++ns; }
  qsort(lab, (size_t)ns, 2 * sizeof(int64_t), cmp_i64_pair); /* ascending label */
  m->nsites = ns;
  m->site_loc = (double *)malloc(sizeof(double) * 3 * (size_t)(ns > 0 ? ns : 1));
  m->site_label = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ns > 0 ? ns : 1));
  for (int64_t k = 0; k < ns; ++k) {
    int64_t v = lab[2 * k + 1];
    int64_t ix = v % m->size[0], iy = (v / m->size[0]) % m->size[1], iz = v / (m->size[0] * m->size[1]);
    m->site_loc[3 * k] = m->origin[0] + (double)ix;
    m->site_loc[3 * k + 1] = m->origin[1] + (double)iy;
    m->site_loc[3 * k + 2] = m->origin[2] + (double)iz;
    m->site_label[k] = lab[2 * k];
  }
  free(pairs); free(lab);
  m->kd = kd_build(m->site_loc, ns);
  size_t nv = (size_t)(m->size[0] * m->size[1] * m->size[2]);
  m->vol = (int32_t *)malloc(sizeof(int32_t) * nv);
  memset(m->vol, 0xff, sizeof(int32_t) * nv);
  return m;
}

static void vor_free(voronoi_map *m) {
  if (!m) return;
  free(m->site_loc); free(m->site_label); kd_free(m->kd); free(m->vol); free(m);
}

/* rrl .txx:530-543 */
static int64_t vor_lookup(voronoi_map *m, const double *y) {
  if (m->nsites == 0) return 0; /* an empty map reads the background value */
  int64_t ix = vor_index(m, 0, y[0]), iy = vor_index(m, 1, y[1]), iz = vor_index(m, 2, y[2]);
  size_t v = (size_t)((iz * m->size[1] + iy) * m->size[0] + ix);
  if (m->vol[v] < 0) {
    double c[3] = { m->origin[0] + (double)ix, m->origin[1] + (double)iy, m->origin[2] + (double)iz };
    m->vol[v] = (int32_t)m->site_label[nn_kd(m->kd, c)];
  }
  return m->vol[v];
}

void lrr_oracle_nn(const double *fixed_loc, int64_t nf, const double *query, int64_t nq, int backend,
                   int64_t *idx_out) {
  if (backend == LRR_ORACLE_NN_KDTREE) {
    kdtree *t = kd_build(fixed_loc, nf);
    for (int64_t i = 0; i < nq; ++i) idx_out[i] = nn_kd(t, query + 3 * i);
    kd_free(t);
  } else if (backend == LRR_ORACLE_NN_VORONOI) {
    voronoi_map *m = vor_build(fixed_loc, nf);
    for (int64_t i = 0; i < nq; ++i) idx_out[i] = vor_lookup(m, query + 3 * i);
    vor_free(m);
  } else {
    for (int64_t i = 0; i < nq; ++i) idx_out[i] = nn_brute(fixed_loc, nf, query + 3 * i);
  }
}

/* ------------------------------------------------------------------------- */
/* feature clouds                                                            */
/* ------------------------------------------------------------------------- */

typedef struct {
  int64_t n;
  double *loc;      /* n x 3; normalised IN PLACE during estimation and restored, as the reference does */
  double *proj;     /* n x 9 */
  double *strength; /* n; float-rounded for the rrl feature type (cdcl_feature_ICP.h:83) */
  uint8_t *shape;
  double *dirs;     /* n x 6 or NULL */
  double *evec;     /* n x 6 or NULL: eigenvectors of covariance_ for the smallest and the largest eigenvalue */
} cloud;

/* columns 0 and 2 of vnl_symmetric_eigensystem_compute(C, V, D) (eigenvalues ascending) for a symmetric positive
 * semi-definite 3x3: its singular vectors, picked by singular value (first index wins among equal values) */
static void sym_eig_ends(const double *Cm, double *v_small, double *v_large) {
  double U[9], sv[3], V[9];
  lrr_oracle_svd3(Cm, U, sv, V);
  int lo = 0, hi = 0;
  for (int j = 1; j < 3; ++j) { if (sv[j] < sv[lo]) lo = j; if (sv[j] > sv[hi]) hi = j; }
  for (int k = 0; k < 3; ++k) { v_small[k] = V[3 * k + lo]; v_large[k] = V[3 * k + hi]; }
}

static void cloud_free(cloud *c) {
  free(c->loc); free(c->proj); free(c->strength); free(c->shape); free(c->dirs); free(c->evec);
  memset(c, 0, sizeof *c);
}

static void cloud_copy_in(cloud *c, const lrr_oracle_features *f, int float_strength) {
  memset(c, 0, sizeof *c);
  int64_t n = f->n;
  size_t m = (size_t)(n > 0 ? n : 1);
  c->n = n;
  c->loc = (double *)malloc(sizeof(double) * 3 * m);
  c->proj = (double *)malloc(sizeof(double) * 9 * m);
  c->strength = (double *)malloc(sizeof(double) * m);
  c->shape = (uint8_t *)calloc(m, 1);
  memcpy(c->loc, f->loc, sizeof(double) * 3 * (size_t)n);
  memcpy(c->proj, f->proj, sizeof(double) * 9 * (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    double s = f->strength ? f->strength[i] : 0.0;
    c->strength[i] = float_strength ? (double)(float)s : s;
  }
  if (f->shape) memcpy(c->shape, f->shape, (size_t)n);
  if (f->dirs) {
    c->dirs = (double *)malloc(sizeof(double) * 6 * m);
    memcpy(c->dirs, f->dirs, sizeof(double) * 6 * (size_t)n);
  }
  if (f->cov) {
    c->evec = (double *)malloc(sizeof(double) * 6 * m);
    for (int64_t i = 0; i < n; ++i) sym_eig_ends(f->cov + 9 * i, c->evec + 6 * i, c->evec + 6 * i + 3);
  }
}

/* ------------------------------------------------------------------------- */
/* SURVEY section 8(f)4, first step: cdcl_estimation::compute_weights          */
/* ------------------------------------------------------------------------- */
/* vnl_symmetric_eigensystem_compute_eigenvals (VXL core/vnl/algo, not in this image; restated from the published
 * closed form — the trigonometric solution of the characteristic cubic — as VXL writes it: PARITY UNPINNED):
 * eigenvalues of a symmetric 3x3 in ascending order. */
static void sym3_eigenvalues(double M11, double M12, double M13, double M22, double M23, double M33,
                             double *l1, double *l2, double *l3) {
  const double b = -M11 - M22 - M33;
  const double c = M11 * M22 + M11 * M33 + M22 * M33 - M12 * M12 - M13 * M13 - M23 * M23;
  const double d = M11 * M23 * M23 + M12 * M12 * M33 + M13 * M13 * M22 - 2.0 * M12 * M13 * M23 - M11 * M22 * M33;
  const double b_3 = b / 3.0;
  const double f = b_3 * b_3 - c / 3.0;
  const double g = b * c / 6.0 - b_3 * b_3 * b_3 - 0.5 * d;
  if (f == 0.0 && g == 0.0) { *l1 = *l2 = *l3 = -b_3; return; }
  const double f3 = f * f * f, g2 = g * g;
  const double sqrt_f = -sqrt(f);
  if (g2 >= f3) { /* repeated root (complex pairs are treated as inaccurate repeated roots) */
    if (g < 0.0) { *l1 = 2.0 * sqrt_f - b_3; *l2 = *l3 = -sqrt_f - b_3; }
    else { *l1 = *l2 = sqrt_f - b_3; *l3 = -2.0 * sqrt_f - b_3; }
    return;
  }
  const double sqrt_f3 = sqrt_f * sqrt_f * sqrt_f;
  const double k = acos(g / sqrt_f3) / 3.0;
  const double j = 2.0 * sqrt_f;
  double a1 = j * cos(k) - b_3, a2 = j * cos(k + M_PI * 2.0 / 3.0) - b_3, a3 = j * cos(k - M_PI * 2.0 / 3.0) - b_3, t;
  if (a2 < a1) { t = a2; a2 = a1; a1 = t; }
  if (a3 < a2) { t = a2; a2 = a3; a3 = t; if (a2 < a1) { t = a2; a2 = a1; a1 = t; } }
  *l1 = a1; *l2 = a2; *l3 = a3;
}

/* vnl_inverse of a 3x3 (adjugate over determinant) */
static void inverse3_cofactor(const double *m, double *o) {
  double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  double id = 1.0 / det;
  o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* weight() of cdcl/cdcl_utils.h:212-232: f (1 - r / b^2)^2 up to r = b^2 = 16, f = 2.9872 */
static double cdc_weight(double residual) {
  const double b_2 = 16.0, f = 2.9872;
  if (residual <= b_2) return f * (1 - residual / b_2) * (1 - residual / b_2);
  return 0;
}

/* cdcl_estimation::compute_weights at one resolution (cdcl/cdcl_estimation.txx:342-545): for every moving feature p,
 * y = T(p); C_p' = J_th C_theta J_th^T + J_p C_p J_p^T (:375-383); all fixed features within 5 sqrt(lambda_max(C_p'))
 * of y (:385-393); at most `max_matches` of them (:400-404: the reference shuffles the list with rand() and keeps the
 * first 100 — here the candidates with the smallest rank under a keyed permutation of the fixed set, the stand-in used
 * for every shuffle of this project); Mahalanobis residual e^T (C_p' + C_q)^-1 e and weight() (:411-423); matches with a
 * positive weight are kept; finally w_ij <- w_ij^2 / (sum_j w_ij  sum_i w_ij) (:524-537).
 * Output per moving feature: row_count[i] matches, to_idx / w at [i * max_matches ...], ascending fixed index.
 * Brute-force radius search: checker code for small cases.  Returns the total number of matches. */
int64_t lrr_oracle_cdc_compute_weights(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, int dof,
                                       const double *A, const double *t, const double *c, const double *param_covar,
                                       const uint32_t keys[4], int max_matches, int64_t *row_count, int64_t *to_idx, double *w) {
  trans T;
  trans_identity(&T, dof);
  memcpy(T.A, A, sizeof T.A); memcpy(T.t, t, sizeof T.t); memcpy(T.c, c, sizeof T.c);
  memcpy(T.covar, param_covar, sizeof(double) * (size_t)dof * (size_t)dof);
  const double eval_multiplier = 5.0;
  int64_t nm = moving->n, nf = fixed->n, total = 0;
  double *to_weights = (double *)calloc((size_t)(nf > 0 ? nf : 1), sizeof(double));
  double *from_weights = (double *)calloc((size_t)(nm > 0 ? nm : 1), sizeof(double));
  int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)(nf > 0 ? nf : 1));
  for (int64_t i = 0; i < nm; ++i) {
    const double *p = moving->loc + 3 * i;
    double y[3], J[3 * MAXDOF], JC[3 * MAXDOF], Cp[9];
    trans_map(&T, p, y);
    trans_jac(&T, p, J);
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < dof; ++cc) {
        double s2 = 0.0;
        for (int k = 0; k < dof; ++k) s2 += J[r * dof + k] * T.covar[k * dof + cc];
        JC[r * dof + cc] = s2;
      }
    for (int r = 0; r < 3; ++r)
      for (int e = 0; e < 3; ++e) {
        double s2 = 0.0;
        for (int k = 0; k < dof; ++k) s2 += JC[r * dof + k] * J[e * dof + k];
        Cp[3 * r + e] = s2;
      }
    { /* + J_p C_p J_p^T, J_p = A (jacobian_wrt_loc) */
      const double *Cm = moving->cov + 9 * i;
      double AC[9];
      for (int r = 0; r < 3; ++r)
        for (int e = 0; e < 3; ++e) AC[3 * r + e] = (T.A[3 * r] * Cm[e] + T.A[3 * r + 1] * Cm[3 + e]) + T.A[3 * r + 2] * Cm[6 + e];
      for (int r = 0; r < 3; ++r)
        for (int e = 0; e < 3; ++e) Cp[3 * r + e] += (AC[3 * r] * T.A[3 * e] + AC[3 * r + 1] * T.A[3 * e + 1]) + AC[3 * r + 2] * T.A[3 * e + 2];
    }
    double l1, l2, l3;
    sym3_eigenvalues(Cp[0], Cp[1], Cp[2], Cp[4], Cp[5], Cp[8], &l1, &l2, &l3);
    const double radius = eval_multiplier * sqrt(l3), r2 = radius * radius;
    int64_t nc = 0;
    for (int64_t j = 0; j < nf; ++j)
      if (dist2(y, fixed->loc + 3 * j) <= r2) { cand[2 * nc] = lrr_oracle_subset_perm(j, nf, keys); cand[2 * nc + 1] = j; ++nc; }
    if (nc > max_matches) { qsort(cand, (size_t)nc, 2 * sizeof(int64_t), cmp_i64_pair); nc = max_matches; }
    for (int64_t k = 0; k < nc; ++k) { cand[2 * k] = cand[2 * k + 1]; }  /* order the kept ones by fixed index */
    qsort(cand, (size_t)nc, 2 * sizeof(int64_t), cmp_i64_pair);
    int64_t kept = 0;
    double sum_w = 0.0;
    for (int64_t k = 0; k < nc; ++k) {
      int64_t j = cand[2 * k];
      const double *q = fixed->loc + 3 * j, *Cq = fixed->cov + 9 * j;
      double e[3] = { y[0] - q[0], y[1] - q[1], y[2] - q[2] }, Cij[9], Ci[9];
      for (int a = 0; a < 9; ++a) Cij[a] = Cp[a] + Cq[a];
      inverse3_cofactor(Cij, Ci);
      double v0 = (e[0] * Ci[0] + e[1] * Ci[3]) + e[2] * Ci[6];
      double v1 = (e[0] * Ci[1] + e[1] * Ci[4]) + e[2] * Ci[7];
      double v2 = (e[0] * Ci[2] + e[1] * Ci[5]) + e[2] * Ci[8];
      double residual = (v0 * e[0] + v1 * e[1]) + v2 * e[2];
      double w0 = cdc_weight(residual);
      if (w0 > 0.0) {
        to_idx[i * max_matches + kept] = j;
        w[i * max_matches + kept] = w0;
        to_weights[j] += w0;
        sum_w += w0;
        ++kept;
      }
    }
    row_count[i] = kept;
    from_weights[i] = sum_w;
    total += kept;
  }
  for (int64_t i = 0; i < nm; ++i)
    for (int64_t k = 0; k < row_count[i]; ++k) {
      double w0 = w[i * max_matches + k];
      w[i * max_matches + k] = w0 * w0 / (from_weights[i] * to_weights[to_idx[i * max_matches + k]]);
    }
  free(to_weights); free(from_weights); free(cand);
  return total;
}

/* rho() and rho_p() of cdcl/cdcl_utils.h:158-201 (modified Beaton-Tukey, b = 4, f = 2.9872; the residual is squared already) */
static double cdc_rho(double residual) {
  const double b_2 = 16.0, f = 2.9872, c = b_2 / 6.0;
  if (residual > b_2) return f * c;
  return f * c * (1 - pow(1 - residual / b_2, 3.0));
}
static double cdc_rho_p(double residual) {
  const double b_2 = 16.0, f = 2.9872;
  if (residual > b_2) return 0.0;
  double temp = 1 - residual / b_2;
  return f * 0.5 * temp * temp;
}

/* cdcl_obj_fun_par: the objective of the CDC-EM M-step's parameter estimate and its gradient
 * (cdcl/cdcl_obj_fun_par.txx:20-41 constructor, :44-142 compute()).  Matches: rows of max_matches slots per moving
 * feature (row_count, to_idx, w: what compute_weights left); a moving feature takes part when its row is not empty.
 *   constructor: JCthJ_i = J_th(p_i) C_theta J_th(p_i)^T (:30-38)
 *   compute():   J_p = jacobian_wrt_loc() = A (:54);  Mkl = jacobian_of_Jp() (:60: affine: the nine unit matrices in
 *                parameter order, cdcl_trans_affine.txx:199-214; rigid: the three skew generators, cdcl_trans_rigid3d.cxx:230-255)
 *                C_ij = JCthJ_i + J_p C_p J_p^T + C_q (:74-84), e = T(p) - q (:88), r = e^T C_ij^-1 e (:92), inlier = rho_p(r) > 0 (:101)
 *                f += w (rho(r) + ln det C_ij) (:105);
 *                g += w (rho_p(r) (2 J_th^T C_ij^-1 e + correct) + correctS) (:109-128), correct_f = -eC^T dC eC,
 *                correctS_f = trace(C_ij^-1 dC), dC = J_p C_p Mkl^T + Mkl C_p J_p^T, eC = C_ij^-1 e
 *                f /= #rows, g /= #rows (:133-141)
 * norm (may be NULL): {cm[3], rm, cf[3], rf} — the features as cdcl_normalize_matches leaves them
 * (cdcl/cdcl_estimation_ICP.txx:751-833): locations (x - c) / r, covariances C / (r r).  The transform is taken as given
 * (the caller normalises it).  g may be NULL.  Returns the number of rows that took part. */
int64_t lrr_oracle_cdc_obj_par(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, int dof, const double *A,
                               const double *t, const double *c, const double *param_covar, int max_matches,
                               const int64_t *row_count, const int64_t *to_idx, const double *w, const double *norm,
                               double *f_out, double *g_out) {
  trans T;
  trans_identity(&T, dof);
  memcpy(T.A, A, sizeof T.A); memcpy(T.t, t, sizeof T.t); memcpy(T.c, c, sizeof T.c);
  memcpy(T.covar, param_covar, sizeof(double) * (size_t)dof * (size_t)dof);
  const int nrot = dof - 3;
  double Mkl[9][9];
  memset(Mkl, 0, sizeof Mkl);
  if (dof == 12) {
    for (int d1 = 0; d1 < 3; ++d1) for (int d2 = 0; d2 < 3; ++d2) Mkl[3 * d1 + d2][3 * d1 + d2] = 1.0;
  } else {
    Mkl[0][3 * 1 + 2] = -1.0; Mkl[0][3 * 2 + 1] = 1.0;
    Mkl[1][3 * 0 + 2] = 1.0;  Mkl[1][3 * 2 + 0] = -1.0;
    Mkl[2][3 * 0 + 1] = -1.0; Mkl[2][3 * 1 + 0] = 1.0;
  }
  double f = 0.0, g[MAXDOF];
  for (int k = 0; k < dof; ++k) g[k] = 0.0;
  int64_t rows = 0;
  for (int64_t i = 0; i < moving->n; ++i) {
    if (row_count[i] <= 0) continue;
    ++rows;
    double p[3], Cp[9];
    for (int d = 0; d < 3; ++d) p[d] = moving->loc[3 * i + d];
    for (int a = 0; a < 9; ++a) Cp[a] = moving->cov[9 * i + a];
    if (norm) {
      for (int d = 0; d < 3; ++d) { p[d] -= norm[d]; p[d] /= norm[3]; }
      for (int a = 0; a < 9; ++a) Cp[a] /= (norm[3] * norm[3]);
    }
    double J[3 * MAXDOF], JC[3 * MAXDOF], Cij_p[9];
    trans_jac(&T, p, J);
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < dof; ++cc) {
        double s2 = 0.0;
        for (int k = 0; k < dof; ++k) s2 += J[r * dof + k] * T.covar[k * dof + cc];
        JC[r * dof + cc] = s2;
      }
    for (int r = 0; r < 3; ++r)
      for (int e = 0; e < 3; ++e) {
        double s2 = 0.0;
        for (int k = 0; k < dof; ++k) s2 += JC[r * dof + k] * J[e * dof + k];
        Cij_p[3 * r + e] = s2;
      }
    double JpC[9], At[9];  /* J_p C_p and J_p^T */
    mat3_mul(T.A, Cp, JpC);
    for (int r = 0; r < 3; ++r) for (int e = 0; e < 3; ++e) At[3 * r + e] = T.A[3 * e + r];
    { double X[9]; mat3_mul(JpC, At, X); for (int a = 0; a < 9; ++a) Cij_p[a] += X[a]; }
    double y[3];
    trans_map(&T, p, y);
    for (int64_t k = 0; k < row_count[i]; ++k) {
      const int64_t j = to_idx[i * max_matches + k];
      const double wj = w[i * max_matches + k];
      double q[3], Cq[9];
      for (int d = 0; d < 3; ++d) q[d] = fixed->loc[3 * j + d];
      for (int a = 0; a < 9; ++a) Cq[a] = fixed->cov[9 * j + a];
      if (norm) {
        for (int d = 0; d < 3; ++d) { q[d] -= norm[4 + d]; q[d] /= norm[7]; }
        for (int a = 0; a < 9; ++a) Cq[a] /= (norm[7] * norm[7]);
      }
      double Cij[9], Ci[9], e[3], eC[3];
      for (int a = 0; a < 9; ++a) Cij[a] = Cij_p[a] + Cq[a];
      inverse3_cofactor(Cij, Ci);
      for (int d = 0; d < 3; ++d) e[d] = y[d] - q[d];
      mat3_vec(Ci, e, eC);
      const double residual = (e[0] * eC[0] + e[1] * eC[1]) + e[2] * eC[2];
      const double rp = cdc_rho_p(residual), rh = cdc_rho(residual);
      if (!(rp > 0)) continue;
      const double det = Cij[0] * (Cij[4] * Cij[8] - Cij[5] * Cij[7]) - Cij[1] * (Cij[3] * Cij[8] - Cij[5] * Cij[6]) + Cij[2] * (Cij[3] * Cij[7] - Cij[4] * Cij[6]);
      f += wj * (rh + log(det));
      if (!g_out) continue;
      double correct[MAXDOF], correctS[MAXDOF];
      for (int k2 = 0; k2 < dof; ++k2) correct[k2] = correctS[k2] = 0.0;
      for (int ff = 0; ff < nrot; ++ff) {
        double Mt[9], X[9], Y[9], dC[9], CpAt[9], Z[9];
        for (int r = 0; r < 3; ++r) for (int e2 = 0; e2 < 3; ++e2) Mt[3 * r + e2] = Mkl[ff][3 * e2 + r];
        mat3_mul(JpC, Mt, X);        /* J_p C_p Mkl^T */
        mat3_mul(Cp, At, CpAt);      /* C_p J_p^T */
        mat3_mul(Mkl[ff], CpAt, Y);  /* Mkl C_p J_p^T */
        for (int a = 0; a < 9; ++a) dC[a] = X[a] + Y[a];
        double dv[3];
        mat3_vec(dC, eC, dv);
        correct[ff] = -((eC[0] * dv[0] + eC[1] * dv[1]) + eC[2] * dv[2]);
        mat3_mul(Ci, dC, Z);
        correctS[ff] = (Z[0] + Z[4]) + Z[8];
      }
      for (int k2 = 0; k2 < dof; ++k2) {
        /* (J_th^T C_ij^-1 e)_k */
        const double jt = (J[0 * dof + k2] * eC[0] + J[1 * dof + k2] * eC[1]) + J[2 * dof + k2] * eC[2];
        g[k2] += wj * (rp * (jt * 2.0 + correct[k2]) + correctS[k2]);
      }
    }
  }
  if (rows > 0) {
    f /= (double)rows;
    for (int k = 0; k < dof; ++k) g[k] *= (1.0 / (double)rows);
  }
  *f_out = f;
  if (g_out) for (int k = 0; k < dof; ++k) g_out[k] = g[k];
  return rows;
}

/* cdcl_obj_fun_cov: the objective of the M-step's covariance estimate and its gradient (cdcl/cdcl_obj_fun_cov.txx:17-38
 * constructor, :42-134 compute()).  x: the upper triangle of the Cholesky factor of the parameter covariance, row by row
 * (:48-54); C_theta = U^T U (:57).
 *   constructor: JCpJ_i = J_p C_p J_p^T (:24-37)
 *   compute():   C_ij = JCpJ_i + J_th C_theta J_th^T + C_q (:71-80), e = q - T(p) (:84), r = e^T C_ij^-1 e;
 *                f += w (rho(r) + ln det C_ij) for EVERY pair (:94-96: no inlier test here);
 *                kl += w (-rho_p(r) temp temp^T + J_th^T C_ij^-1 J_th), temp = J_th^T C_ij^-1 e (:99-102);
 *                g_ind = <kl, U^T se + se^T U> for the single-entry matrices se(d1, d2), d1 <= d2 (:106-118);
 *                f /= #pairs, g /= #pairs (:122-131)
 * norm as in lrr_oracle_cdc_obj_par.  g may be NULL.  Returns the number of pairs. */
int64_t lrr_oracle_cdc_obj_cov(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, int dof, const double *A,
                               const double *t, const double *c, const double *x_upper, int max_matches,
                               const int64_t *row_count, const int64_t *to_idx, const double *w, const double *norm,
                               double *f_out, double *g_out) {
  trans T;
  trans_identity(&T, dof);
  memcpy(T.A, A, sizeof T.A); memcpy(T.t, t, sizeof T.t); memcpy(T.c, c, sizeof T.c);
  double U[MAXDOF * MAXDOF];
  memset(U, 0, sizeof U);
  { int ind = 0; for (int d1 = 0; d1 < dof; ++d1) for (int d2 = d1; d2 < dof; ++d2) U[d1 * dof + d2] = x_upper[ind++]; }
  for (int a = 0; a < dof; ++a)
    for (int b = 0; b < dof; ++b) {
      double s2 = 0.0;
      for (int k = 0; k < dof; ++k) s2 += U[k * dof + a] * U[k * dof + b];
      T.covar[a * dof + b] = s2;
    }
  double kl[MAXDOF * MAXDOF];
  memset(kl, 0, sizeof kl);
  double f = 0.0, pairs = 0.0;
  for (int64_t i = 0; i < moving->n; ++i) {
    if (row_count[i] <= 0) continue;
    double p[3], Cp[9];
    for (int d = 0; d < 3; ++d) p[d] = moving->loc[3 * i + d];
    for (int a = 0; a < 9; ++a) Cp[a] = moving->cov[9 * i + a];
    if (norm) {
      for (int d = 0; d < 3; ++d) { p[d] -= norm[d]; p[d] /= norm[3]; }
      for (int a = 0; a < 9; ++a) Cp[a] /= (norm[3] * norm[3]);
    }
    double J[3 * MAXDOF], JC[3 * MAXDOF], Cij_p[9], At[9], X[9];
    mat3_mul(T.A, Cp, X);
    for (int r = 0; r < 3; ++r) for (int e = 0; e < 3; ++e) At[3 * r + e] = T.A[3 * e + r];
    mat3_mul(X, At, Cij_p);  /* JCpJ */
    trans_jac(&T, p, J);
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < dof; ++cc) {
        double s2 = 0.0;
        for (int k = 0; k < dof; ++k) s2 += J[r * dof + k] * T.covar[k * dof + cc];
        JC[r * dof + cc] = s2;
      }
    for (int r = 0; r < 3; ++r)
      for (int e = 0; e < 3; ++e) {
        double s2 = 0.0;
        for (int k = 0; k < dof; ++k) s2 += JC[r * dof + k] * J[e * dof + k];
        Cij_p[3 * r + e] += s2;
      }
    double y[3];
    trans_map(&T, p, y);
    pairs += (double)row_count[i];
    for (int64_t k = 0; k < row_count[i]; ++k) {
      const int64_t j = to_idx[i * max_matches + k];
      const double wj = w[i * max_matches + k];
      double q[3], Cq[9];
      for (int d = 0; d < 3; ++d) q[d] = fixed->loc[3 * j + d];
      for (int a = 0; a < 9; ++a) Cq[a] = fixed->cov[9 * j + a];
      if (norm) {
        for (int d = 0; d < 3; ++d) { q[d] -= norm[4 + d]; q[d] /= norm[7]; }
        for (int a = 0; a < 9; ++a) Cq[a] /= (norm[7] * norm[7]);
      }
      double Cij[9], Ci[9], e[3], eC[3];
      for (int a = 0; a < 9; ++a) Cij[a] = Cij_p[a] + Cq[a];
      inverse3_cofactor(Cij, Ci);
      for (int d = 0; d < 3; ++d) e[d] = q[d] - y[d];
      mat3_vec(Ci, e, eC);
      const double residual = (e[0] * eC[0] + e[1] * eC[1]) + e[2] * eC[2];
      const double det = Cij[0] * (Cij[4] * Cij[8] - Cij[5] * Cij[7]) - Cij[1] * (Cij[3] * Cij[8] - Cij[5] * Cij[6]) + Cij[2] * (Cij[3] * Cij[7] - Cij[4] * Cij[6]);
      f += wj * (cdc_rho(residual) + log(det));
      if (!g_out) continue;
      const double rp = cdc_rho_p(residual);
      double temp[MAXDOF], CiJ[3 * MAXDOF];
      for (int a = 0; a < dof; ++a) temp[a] = (J[0 * dof + a] * eC[0] + J[1 * dof + a] * eC[1]) + J[2 * dof + a] * eC[2];
      for (int r = 0; r < 3; ++r)
        for (int a = 0; a < dof; ++a) CiJ[r * dof + a] = (Ci[3 * r] * J[a] + Ci[3 * r + 1] * J[dof + a]) + Ci[3 * r + 2] * J[2 * dof + a];
      for (int a = 0; a < dof; ++a)
        for (int b = 0; b < dof; ++b) {
          const double jcj = (J[a] * CiJ[b] + J[dof + a] * CiJ[dof + b]) + J[2 * dof + a] * CiJ[2 * dof + b];
          kl[a * dof + b] += wj * (-rp * temp[a] * temp[b] + jcj);
        }
    }
  }
  if (pairs > 0) f /= pairs;
  *f_out = f;
  if (g_out) {
    int ind = 0;
    for (int d1 = 0; d1 < dof; ++d1)
      for (int d2 = d1; d2 < dof; ++d2) {
        /* dC = U^T se + se^T U: row d2 of (U^T se) is column... (U^T se)[a][b] = U[d1][a] (b == d2); (se^T U)[a][b] = U[d1][b] (a == d2) */
        double dot = 0.0;
        for (int a = 0; a < dof; ++a) dot += kl[a * dof + d2] * U[d1 * dof + a];
        for (int b = 0; b < dof; ++b) dot += kl[d2 * dof + b] * U[d1 * dof + b];
        g_out[ind++] = pairs > 0 ? dot * (1.0 / pairs) : 0.0;
      }
  }
  return (int64_t)pairs;
}

/* cdcl_normalize_matches over one-to-many rows (cdcl/cdcl_estimation_ICP.txx:751-795): centroid of the moving features
 * whose row is not empty, centroid of the DISTINCT matched fixed features, mean radii about them -> {cm[3], rm, cf[3], rf} */
void lrr_oracle_cdc_normalize_stats(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, int max_matches,
                                    const int64_t *row_count, const int64_t *to_idx, double *norm) {
  unsigned char *seen = (unsigned char *)calloc((size_t)(fixed->n > 0 ? fixed->n : 1), 1);
  double cm[3] = {0, 0, 0}, cf[3] = {0, 0, 0}, rows = 0.0, nd = 0.0, rm = 0.0, rf = 0.0;
  for (int64_t i = 0; i < moving->n; ++i) {
    if (row_count[i] <= 0) continue;
    rows += 1.0;
    for (int d = 0; d < 3; ++d) cm[d] += moving->loc[3 * i + d];
    for (int64_t k = 0; k < row_count[i]; ++k) seen[to_idx[i * max_matches + k]] = 1;
  }
  for (int64_t j = 0; j < fixed->n; ++j)
    if (seen[j]) { nd += 1.0; for (int d = 0; d < 3; ++d) cf[d] += fixed->loc[3 * j + d]; }
  for (int d = 0; d < 3; ++d) { cm[d] /= rows; cf[d] /= nd; }
  for (int64_t i = 0; i < moving->n; ++i) {
    if (row_count[i] <= 0) continue;
    const double a = moving->loc[3 * i] - cm[0], b = moving->loc[3 * i + 1] - cm[1], c = moving->loc[3 * i + 2] - cm[2];
    rm += sqrt((a * a + b * b) + c * c);
  }
  for (int64_t j = 0; j < fixed->n; ++j)
    if (seen[j]) {
      const double a = fixed->loc[3 * j] - cf[0], b = fixed->loc[3 * j + 1] - cf[1], c = fixed->loc[3 * j + 2] - cf[2];
      rf += sqrt((a * a + b * b) + c * c);
    }
  free(seen);
  for (int d = 0; d < 3; ++d) { norm[d] = cm[d]; norm[4 + d] = cf[d]; }
  norm[3] = rm / rows;
  norm[7] = rf / nd;
}

/* One EM iteration of cdcl_estimation (cdcl/cdcl_estimation.txx:95-187) for the affine transform (dof 12) and for the rigid
 * one with its small-angle increment (dof 6), over a pyramid of `levels` resolutions ([0] the finest; the estimator starts at
 * the coarsest, cdcl/cdcl_estimation_ICP.txx:118):
 *   compute_weights(resolution_switch_allowed = iteration > 5) (:342-545) with its resolution switch (:491-517: the statistic
 *   avg_largest_eval of :384 against twice the fixed spacing, one restart at most) — the restart APPENDS to from_weights (:356-358
 *   are outside the loop), so match i of the second pass is normalised with entry i of the first pass (:524-537) — and the
 *   closest-point fallback (:543-544);
 *   rigid: the transform becomes an increment (identity, the global covariance), every from_ a copy at G(p) with Jp C Jp^T (:111-141);
 *   from iteration 6 on estimate_parameters() (:210-260), rigid: recompose_increment + compute_weights on the global transform + the
 *   copies again (:150-171), else compute_weights; estimate_covariance() (:264-338); rigid: the global transform takes the covariance (:178-182).
 * Both estimates: cdcl_normalize_matches + trans->normalize, the stand-in optimiser of include/lrr_amd/lbfgs_standin.h (vnl_lbfgs is
 * third party and absent) capped at five evaluations with cdcl_lbfgs' 5 % stop, un-normalise.  T (A, t, c, covar) is updated in place;
 * returns parameters_converged && covariance_converged && |covar|_F < 1e-3 (:186), -1 when the covariance lost positive definiteness. */
#include "lrr_amd/lbfgs_standin.h"
typedef struct {
  const lrr_oracle_features *mv, *fx;   /* the features the objective sees (rigid: the pre-mapped copies) */
  int dof, max_matches;
  const int64_t *rc, *ti;
  const double *w;
  const double *norm;
  trans *T;      /* normalised transform: centre and (par objective) covariance */
} em_ctx;
static void em_fg_par(void *u, const double *x, double *f, double *g) {
  em_ctx *E = (em_ctx *)u;
  trans Tx = *E->T;
  trans_set_param(&Tx, x);  /* (cdcl_obj_fun_par.txx:50) */
  lrr_oracle_cdc_obj_par(E->mv, E->fx, E->dof, Tx.A, Tx.t, Tx.c, Tx.covar, E->max_matches, E->rc, E->ti, E->w, E->norm, f, g);
}
static void em_fg_cov(void *u, const double *x, double *f, double *g) {
  em_ctx *E = (em_ctx *)u;
  lrr_oracle_cdc_obj_cov(E->mv, E->fx, E->dof, E->T->A, E->T->t, E->T->c, x, E->max_matches, E->rc, E->ti, E->w, E->norm, f, g);
}
/* upper-triangular Cholesky factor U (U^T U = C), row-major n x n; returns 0 when C is not positive definite */
static int chol_upper(int n, const double *C, double *U) {
  memset(U, 0, sizeof(double) * (size_t)n * n);
  for (int i = 0; i < n; ++i) {
    for (int j = i; j < n; ++j) {
      double s2 = C[i * n + j];
      for (int k = 0; k < i; ++k) s2 -= U[k * n + i] * U[k * n + j];
      if (i == j) { if (!(s2 > 0.0)) return 0; U[i * n + i] = sqrt(s2); }
      else U[i * n + j] = s2 / U[i * n + i];
    }
  }
  return 1;
}
/* one pass of compute_weights at one resolution, BEFORE the normalisation (:369-488): raw weights in the rows, sum_j w_ij per
 * moving feature, sum_i w_ij per fixed feature; *eval_sum = sum over ALL moving features of sqrt(lambda_max(J C_theta J^T)) (:384).
 * Tm: the transform the features are mapped and the Jacobians taken with; covar: the parameter covariance (transfer_covar uses
 * trans_'s, cdcl_estimation.h:96-98).  Returns matches_.size() (rows that are not empty). */
static int64_t cdc_weights_pass(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, int dof, const trans *Tm, const double *covar,
                                const uint32_t keys[4], int max_matches, int64_t *row_count, int64_t *to_idx, double *w, double *from_sum,
                                double *to_sum, double *eval_sum) {
  const int64_t nm = moving->n, nf = fixed->n;
  int64_t rows = 0;
  int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)(nf > 0 ? nf : 1));
  for (int64_t j = 0; j < nf; ++j) to_sum[j] = 0.0;
  *eval_sum = 0.0;
  for (int64_t i = 0; i < nm; ++i) {
    const double *p = moving->loc + 3 * i;
    double y[3], J[3 * MAXDOF], JC[3 * MAXDOF], Cp[9];
    trans_map(Tm, p, y);
    trans_jac(Tm, p, J);
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < dof; ++cc) {
        double s2 = 0.0;
        for (int k = 0; k < dof; ++k) s2 += J[r * dof + k] * covar[k * dof + cc];
        JC[r * dof + cc] = s2;
      }
    for (int r = 0; r < 3; ++r)
      for (int e = 0; e < 3; ++e) {
        double s2 = 0.0;
        for (int k = 0; k < dof; ++k) s2 += JC[r * dof + k] * J[e * dof + k];
        Cp[3 * r + e] = s2;
      }
    double l1, l2, l3;
    sym3_eigenvalues(Cp[0], Cp[1], Cp[2], Cp[4], Cp[5], Cp[8], &l1, &l2, &l3);
    *eval_sum += sqrt(l3);
    {
      const double *Cm = moving->cov + 9 * i;
      double AC[9];
      for (int r = 0; r < 3; ++r)
        for (int e = 0; e < 3; ++e) AC[3 * r + e] = (Tm->A[3 * r] * Cm[e] + Tm->A[3 * r + 1] * Cm[3 + e]) + Tm->A[3 * r + 2] * Cm[6 + e];
      for (int r = 0; r < 3; ++r)
        for (int e = 0; e < 3; ++e) Cp[3 * r + e] += (AC[3 * r] * Tm->A[3 * e] + AC[3 * r + 1] * Tm->A[3 * e + 1]) + AC[3 * r + 2] * Tm->A[3 * e + 2];
    }
    sym3_eigenvalues(Cp[0], Cp[1], Cp[2], Cp[4], Cp[5], Cp[8], &l1, &l2, &l3);
    const double radius = 5.0 * sqrt(l3), r2 = radius * radius;
    int64_t nc = 0;
    for (int64_t j = 0; j < nf; ++j)
      if (dist2(y, fixed->loc + 3 * j) <= r2) { cand[2 * nc] = lrr_oracle_subset_perm(j, nf, keys); cand[2 * nc + 1] = j; ++nc; }
    if (nc > max_matches) { qsort(cand, (size_t)nc, 2 * sizeof(int64_t), cmp_i64_pair); nc = max_matches; }
    for (int64_t k = 0; k < nc; ++k) { cand[2 * k] = cand[2 * k + 1]; }
    qsort(cand, (size_t)nc, 2 * sizeof(int64_t), cmp_i64_pair);
    int64_t kept = 0;
    double sum_w = 0.0;
    for (int64_t k = 0; k < nc; ++k) {
      int64_t j = cand[2 * k];
      const double *q = fixed->loc + 3 * j, *Cq = fixed->cov + 9 * j;
      double e[3] = { y[0] - q[0], y[1] - q[1], y[2] - q[2] }, Cij[9], Ci[9];
      for (int a2 = 0; a2 < 9; ++a2) Cij[a2] = Cp[a2] + Cq[a2];
      inverse3_cofactor(Cij, Ci);
      double v0 = (e[0] * Ci[0] + e[1] * Ci[3]) + e[2] * Ci[6];
      double v1 = (e[0] * Ci[1] + e[1] * Ci[4]) + e[2] * Ci[7];
      double v2 = (e[0] * Ci[2] + e[1] * Ci[5]) + e[2] * Ci[8];
      double residual = (v0 * e[0] + v1 * e[1]) + v2 * e[2];
      double w0 = cdc_weight(residual);
      if (w0 > 0.0) {
        to_idx[i * max_matches + kept] = j;
        w[i * max_matches + kept] = w0;
        to_sum[j] += w0;
        sum_w += w0;
        ++kept;
      }
    }
    row_count[i] = kept;
    from_sum[i] = sum_w;
    if (kept > 0) ++rows;
  }
  free(cand);
  return rows;
}
typedef struct {
  int dof, levels, max_matches, res, max_res;
  const lrr_oracle_features *const *moving, *const *fixed;
  const double *spacing;
  uint64_t *seed;
  int64_t *rc, *ti;
  double *w;
  int64_t matches;   /* matches_.size() after compute_weights */
  int fell_back;
  double avg;        /* avg_largest_eval of the last pass */
  int downs, ups;    /* resolution switches of this iteration */
} em_state;
/* compute_weights (:342-545) */
static void em_compute_weights(em_state *S, int allowed, const trans *Tm, const trans *Tcur) {
  double *all_from = NULL;   /* from_weights: appended across the restart */
  int64_t n_all = 0, cap_all = 0;
  int stop = 0, not_restarted = 1;
  double *from_sum = NULL, *to_sum = NULL;
  S->fell_back = 0;
  while (!stop) {
    const lrr_oracle_features *mv = S->moving[S->res], *fx = S->fixed[S->res];
    uint32_t keys[4];
    lrr_oracle_perm_keys(S->seed, keys);
    free(from_sum); free(to_sum);
    from_sum = (double *)calloc((size_t)(mv->n > 0 ? mv->n : 1), sizeof(double));
    to_sum = (double *)calloc((size_t)(fx->n > 0 ? fx->n : 1), sizeof(double));
    double eval_sum = 0.0;
    S->matches = cdc_weights_pass(mv, fx, S->dof, Tm, Tcur->covar, keys, S->max_matches, S->rc, S->ti, S->w, from_sum, to_sum, &eval_sum);
    for (int64_t i = 0; i < mv->n; ++i)
      if (S->rc[i] > 0) {
        if (n_all == cap_all) { cap_all = cap_all ? 2 * cap_all : 1024; all_from = (double *)realloc(all_from, sizeof(double) * (size_t)cap_all); }
        all_from[n_all++] = from_sum[i];
      }
    const double avg_largest_eval = eval_sum / (double)S->matches;
    S->avg = avg_largest_eval;
    stop = 1;
    if (avg_largest_eval < 2.0 * S->spacing[S->res] && allowed && not_restarted) {
      if (S->res > 0) { --S->res; stop = 0; not_restarted = 0; ++S->downs; }
    }
    if (S->res + 1 <= S->max_res && avg_largest_eval > 2.0 * S->spacing[S->res + 1] && allowed && not_restarted) {
      ++S->res; stop = 0; not_restarted = 0; ++S->ups;
    }
  }
  /* normalise (:524-537): wi = from_weights[i], i the match's position in matches_ */
  {
    const lrr_oracle_features *mv = S->moving[S->res];
    int64_t k = 0;
    for (int64_t i = 0; i < mv->n; ++i) {
      if (S->rc[i] <= 0) continue;
      const double wi = all_from[k++];
      for (int64_t j = 0; j < S->rc[i]; ++j) {
        const double w0 = S->w[i * S->max_matches + j];
        S->w[i * S->max_matches + j] = w0 * w0 / (wi * to_sum[S->ti[i * S->max_matches + j]]);
      }
    }
  }
  free(all_from); free(from_sum); free(to_sum);
  if (S->matches == 0) {  /* find_closest_euclidean (cdcl_estimation_ICP.txx:451-489): nearest fixed feature under trans_, weight 1 */
    const lrr_oracle_features *mv = S->moving[S->res], *fx = S->fixed[S->res];
    for (int64_t i = 0; i < mv->n; ++i) {
      double y[3], best = INFINITY;
      int64_t bj = 0;
      trans_map(Tcur, mv->loc + 3 * i, y);
      for (int64_t j = 0; j < fx->n; ++j) { const double d = dist2(y, fx->loc + 3 * j); if (d < best) { best = d; bj = j; } }
      S->rc[i] = fx->n > 0 ? 1 : 0;
      S->ti[i * S->max_matches] = bj;
      S->w[i * S->max_matches] = 1.0;
    }
    S->matches = fx->n > 0 ? mv->n : 0;
    S->fell_back = 1;
  }
}
/* the copies the rigid increment works on (:118-141): location G(p), covariance Jp C Jp^T, Jp = R */
static void em_premap(const lrr_oracle_features *mv, const trans *G, lrr_oracle_features *out, double *loc, double *cov) {
  for (int64_t i = 0; i < mv->n; ++i) {
    trans_map(G, mv->loc + 3 * i, loc + 3 * i);
    const double *Cm = mv->cov + 9 * i;
    double RC[9];
    for (int r = 0; r < 3; ++r)
      for (int e = 0; e < 3; ++e) RC[3 * r + e] = (G->A[3 * r] * Cm[e] + G->A[3 * r + 1] * Cm[3 + e]) + G->A[3 * r + 2] * Cm[6 + e];
    for (int r = 0; r < 3; ++r)
      for (int e = 0; e < 3; ++e) cov[9 * i + 3 * r + e] = (RC[3 * r] * G->A[3 * e] + RC[3 * r + 1] * G->A[3 * e + 1]) + RC[3 * r + 2] * G->A[3 * e + 2];
  }
  *out = *mv;
  out->loc = loc;
  out->cov = cov;
}
static void em_get_param(const trans *T, double *x) {
  if (T->dof == 6) { for (int i = 0; i < 3; ++i) { x[i] = T->abg[i]; x[3 + i] = T->t[i]; } }
  else { for (int i = 0; i < 9; ++i) x[i] = T->A[i]; for (int i = 0; i < 3; ++i) x[9 + i] = T->t[i]; }
}
int lrr_oracle_cdc_em_iteration_pyramid(int dof, int levels, const lrr_oracle_features *const *moving, const lrr_oracle_features *const *fixed,
                                        const double *spacing, double *A, double *t, double *c, double *covar, unsigned iteration,
                                        uint64_t *seed_state, int max_matches, int *res_inout, int64_t *row_count, int64_t *to_idx, double *w,
                                        double *info /* 8 */, double *state /* 6: resolution, matches_.size(), fell back, avg_largest_eval, switches to a finer / a coarser resolution; may be NULL */) {
  const int ncov = dof * (dof + 1) / 2;
  const unsigned startEM = 5;
  const int allowed = iteration > startEM;
  const double one = 1.0;
  trans T;
  trans_from_Atc(&T, dof, A, t, c);
  memcpy(T.covar, covar, sizeof(double) * (size_t)dof * dof);
  for (int k = 0; k < 8; ++k) info[k] = 0.0;
  em_state S;
  S.dof = dof; S.levels = levels; S.max_matches = max_matches; S.res = *res_inout; S.max_res = levels - 1;
  S.moving = moving; S.fixed = fixed; S.spacing = spacing ? spacing : &one; S.seed = seed_state;
  S.rc = row_count; S.ti = to_idx; S.w = w; S.matches = 0; S.fell_back = 0; S.avg = 0.0; S.downs = S.ups = 0;
  int64_t nm_max = 1;
  for (int r = 0; r < levels; ++r) if (moving[r]->n > nm_max) nm_max = moving[r]->n;
  double *pm_loc = (double *)malloc(sizeof(double) * 3 * (size_t)nm_max), *pm_cov = (double *)malloc(sizeof(double) * 9 * (size_t)nm_max);
  lrr_oracle_features pm;
  const int rigid = dof == 6;
  int rcode = 0;

  em_compute_weights(&S, allowed, &T, &T);
  trans G = T, *cur = &T;  /* cur: trans_ */
  trans inc;
  if (rigid) {  /* from now on, transform is incremental (:111-141) */
    trans_identity(&inc, dof);
    memcpy(inc.covar, G.covar, sizeof(double) * (size_t)dof * dof);
    em_premap(moving[S.res], &G, &pm, pm_loc, pm_cov);
    cur = &inc;
  }
  int parameters_converged = 0, covariance_converged = 0;
  double norm[8];
  em_ctx E;
  E.dof = dof; E.max_matches = max_matches; E.rc = row_count; E.ti = to_idx; E.w = w; E.norm = norm; E.T = cur;
  if (iteration > startEM) {
    E.mv = rigid ? &pm : moving[S.res]; E.fx = fixed[S.res];
    lrr_oracle_cdc_normalize_stats(E.mv, E.fx, max_matches, row_count, to_idx, norm);
    trans_normalize(cur, norm[3], norm[7], norm, norm + 4);
    double x[MAXDOF], x_old[MAXDOF];
    em_get_param(cur, x);
    memcpy(x_old, x, sizeof x);
    lrr_lbfgs_result r = lrr_lbfgs_minimize_stop(dof, x, em_fg_par, &E, 5, LRR_CDCL_LBFGS_STOP_FRACTION);  /* cdcl/cdcl_lbfgs.h:28-36 */
    parameters_converged = (r.f_start - r.f_end) < 1e-4;
    if (r.code < 0) memcpy(x, x_old, sizeof x);
    trans_set_param(cur, x);
    trans_unnormalize(cur, norm[3], norm[7], norm, norm + 4);
    info[0] = r.f_start; info[1] = r.f_end; info[4] = (double)r.code; info[6] = (double)r.evals;
    if (rigid) {
      trans_recompose_increment(&G, cur);
      em_compute_weights(&S, allowed, &G, cur);   /* J and map of the global transform, covariance of trans_ (:153) */
      em_premap(moving[S.res], &G, &pm, pm_loc, pm_cov);
    } else {
      em_compute_weights(&S, allowed, cur, cur);
    }
  }
  {
    E.mv = rigid ? &pm : moving[S.res]; E.fx = fixed[S.res];
    lrr_oracle_cdc_normalize_stats(E.mv, E.fx, max_matches, row_count, to_idx, norm);
    trans_normalize(cur, norm[3], norm[7], norm, norm + 4);
    double U[MAXDOF * MAXDOF], x[MAXDOF * (MAXDOF + 1) / 2], x_old[MAXDOF * (MAXDOF + 1) / 2];
    if (!chol_upper(dof, cur->covar, U)) { rcode = -1; goto done; }
    { int ind = 0; for (int d1 = 0; d1 < dof; ++d1) for (int d2 = d1; d2 < dof; ++d2) x[ind++] = U[d1 * dof + d2]; }
    memcpy(x_old, x, sizeof(double) * (size_t)ncov);
    lrr_lbfgs_result r = lrr_lbfgs_minimize_stop(ncov, x, em_fg_cov, &E, 5, LRR_CDCL_LBFGS_STOP_FRACTION);
    covariance_converged = (r.f_start - r.f_end) < 1e-5;
    if (r.code < 0) memcpy(x, x_old, sizeof(double) * (size_t)ncov);
    memset(U, 0, sizeof U);
    { int ind = 0; for (int d1 = 0; d1 < dof; ++d1) for (int d2 = d1; d2 < dof; ++d2) U[d1 * dof + d2] = x[ind++]; }
    for (int a2 = 0; a2 < dof; ++a2)
      for (int b2 = 0; b2 < dof; ++b2) {
        double s2 = 0.0;
        for (int k = 0; k < dof; ++k) s2 += U[k * dof + a2] * U[k * dof + b2];
        cur->covar[a2 * dof + b2] = s2;
      }
    trans_unnormalize(cur, norm[3], norm[7], norm, norm + 4);
    info[2] = r.f_start; info[3] = r.f_end; info[5] = (double)r.code; info[7] = (double)r.evals;
  }
  if (rigid) {  /* (:178-182) */
    memcpy(G.covar, cur->covar, sizeof(double) * (size_t)dof * dof);
    T = G;
  }
  memcpy(A, T.A, sizeof T.A); memcpy(t, T.t, sizeof T.t); memcpy(c, T.c, sizeof T.c);
  memcpy(covar, T.covar, sizeof(double) * (size_t)dof * dof);
  {
    double fro = 0.0;
    for (int i = 0; i < dof * dof; ++i) fro += T.covar[i] * T.covar[i];
    rcode = parameters_converged && covariance_converged && sqrt(fro) < 1e-3;
  }
done:
  *res_inout = S.res;
  if (state) { state[0] = (double)S.res; state[1] = (double)S.matches; state[2] = (double)S.fell_back; state[3] = S.avg; state[4] = (double)S.downs; state[5] = (double)S.ups; }
  free(pm_loc); free(pm_cov);
  return rcode;
}
int lrr_oracle_cdc_em_iteration(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, double *A, double *t, double *c,
                                double *covar, unsigned iteration, uint64_t *seed_state, int max_matches,
                                int64_t *row_count, int64_t *to_idx, double *w, double *info /* 8: f_par start/end, f_cov start/end, codes, evals */) {
  int res = 0;
  return lrr_oracle_cdc_em_iteration_pyramid(12, 1, &moving, &fixed, NULL, A, t, c, covar, iteration, seed_state, max_matches, &res, row_count, to_idx, w, info, NULL);
}

/* ------------------------------------------------------------------------- */
/* forward estimator: cdcl_estimation_ICP_matching_all / rrl_estimation_ICP_matching_all */
/* ------------------------------------------------------------------------- */

typedef struct {
  int rrl;              /* 0: cdcl class + cdcl_feature_with_shape; 1: rrl class + cdcl_feature_ICP */
  int dof;
  cloud moving;         /* moving_[res_] */
  cloud fixed;          /* fixed_[res_]  */
  kdtree *kd;           /* kd_tree_[res_] (built from the un-normalised locations, once) */
  voronoi_map *vor;     /* fixed_voronoi_map_ (rrl .txx:496-498), when the backend is the Voronoi lookup */
  int nn_backend;
  int64_t *order;       /* current order of the moving_ vector */
  int64_t number_points;
  trans tr;             /* trans_ */
  int normalize_matches, weight_by_strength, scale_estimated;
  /* matches_: exactly one to_ per from_ on this path */
  int64_t nmatch, cap;
  int64_t *m_from, *m_to;
  double *m_w;
  int from_is_copy;     /* rigid pre-map replaced from_ by new feature objects */
  double premap_R[9];   /* ... whose covariance is R C R^T: rotation of that pre-map */
  double *m_from_loc;   /* their locations (3 per match) */
  uint8_t *flag;        /* scratch: distinct-to marker, size fixed.n */
  double last_scale, injected_scale;
  double scale_constant; /* 1 in the matching_all classes (cdcl .txx:730-732), 2 in their base class (cdcl_estimation_ICP.txx:580-582) */
  uint64_t rng;
  int64_t *explicit_order;
  int64_t explicit_n;
} fwd_est;

static fwd_est *fwd_create(int rrl, int dof, const lrr_oracle_features *moving,
                           const lrr_oracle_features *fixed, const trans *T, uint64_t seed,
                           int nn_backend) {
  fwd_est *f = (fwd_est *)calloc(1, sizeof *f);
  f->rrl = rrl; f->dof = dof; f->nn_backend = nn_backend;
  cloud_copy_in(&f->moving, moving, rrl);
  cloud_copy_in(&f->fixed, fixed, rrl);
  if (nn_backend == LRR_ORACLE_NN_KDTREE) f->kd = kd_build(f->fixed.loc, f->fixed.n);
  if (nn_backend == LRR_ORACLE_NN_VORONOI) f->vor = vor_build(f->fixed.loc, f->fixed.n);
  f->order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(f->moving.n > 0 ? f->moving.n : 1));
  for (int64_t i = 0; i < f->moving.n; ++i) f->order[i] = i;
  f->tr = *T;
  /* ctor: number_points_ = min(1000, |moving|)  (cdcl .txx:73-91; rrl .txx:89-113) */
  f->number_points = 1000;
  if (f->number_points > f->moving.n) f->number_points = f->moving.n;
  f->flag = (uint8_t *)calloc((size_t)(f->fixed.n > 0 ? f->fixed.n : 1), 1);
  f->rng = seed;
  f->injected_scale = -1.0;
  f->scale_constant = 1.0;
  return f;
}

static void fwd_free(fwd_est *f) {
  if (!f) return;
  cloud_free(&f->moving); cloud_free(&f->fixed); kd_free(f->kd); vor_free(f->vor);
  free(f->order); free(f->m_from); free(f->m_to); free(f->m_w); free(f->m_from_loc);
  free(f->flag); free(f->explicit_order); free(f);
}

static void fwd_reserve(fwd_est *f, int64_t n) {
  if (n <= f->cap) return;
  f->cap = n;
  f->m_from = (int64_t *)realloc(f->m_from, sizeof(int64_t) * (size_t)n);
  f->m_to = (int64_t *)realloc(f->m_to, sizeof(int64_t) * (size_t)n);
  f->m_w = (double *)realloc(f->m_w, sizeof(double) * (size_t)n);
  f->m_from_loc = (double *)realloc(f->m_from_loc, sizeof(double) * 3 * (size_t)n);
}

static inline const double *fwd_from_loc(const fwd_est *f, int64_t i) {
  return f->from_is_copy ? f->m_from_loc + 3 * i : f->moving.loc + 3 * f->m_from[i];
}
static inline double *fwd_from_loc_mut(fwd_est *f, int64_t i) {
  return f->from_is_copy ? f->m_from_loc + 3 * i : f->moving.loc + 3 * f->m_from[i];
}

/* set_transform (cdcl .txx:139-153; rrl .txx:156-171) */
static void fwd_set_transform(fwd_est *f, const trans *T) {
  f->tr = *T;
  f->scale_estimated = 0;
  f->number_points = 1000;
  if (f->number_points > f->moving.n) f->number_points = f->moving.n;
}

/* set_moving_data (cdcl .h:96: matches_.clear(); moving_ = moving) */
static void fwd_set_moving_data(fwd_est *f, const lrr_oracle_features *moving) {
  cloud_free(&f->moving);
  cloud_copy_in(&f->moving, moving, f->rrl);
  f->order = (int64_t *)realloc(f->order, sizeof(int64_t) * (size_t)(f->moving.n > 0 ? f->moving.n : 1));
  for (int64_t i = 0; i < f->moving.n; ++i) f->order[i] = i;
  f->nmatch = 0;
  f->from_is_copy = 0;
}

/* set_number_points (cdcl .txx:94-108; rrl .txx:116-130) */
static int fwd_set_number_points(fwd_est *f, int64_t np) {
  if (np >= f->moving.n) { f->number_points = f->moving.n; return 1; }
  f->number_points = np;
  return 0;
}

/* squared projected residual e^T P e, clamped at 0 (cdcl .txx:214-220, 687-692; rrl .txx:227-233).
 * Canonical order: v = e^T P (sum over rows of P), then v . e. */
static inline double proj_residual2(const double *y, const double *q, const double *P) {
  double e0 = y[0] - q[0], e1 = y[1] - q[1], e2 = y[2] - q[2];
  double v0 = (e0 * P[0] + e1 * P[3]) + e2 * P[6];
  double v1 = (e0 * P[1] + e1 * P[4]) + e2 * P[7];
  double v2 = (e0 * P[2] + e1 * P[5]) + e2 * P[8];
  double r2 = (v0 * e0 + v1 * e1) + v2 * e2;
  if (r2 < 0.0) r2 = 0.0;
  return r2;
}

static inline double fwd_residual2(const fwd_est *f, int64_t i) {
  double y[3];
  trans_map(&f->tr, fwd_from_loc(f, i), y);
  int64_t q = f->m_to[i];
  return proj_residual2(y, f->fixed.loc + 3 * q, f->fixed.proj + 9 * q);
}

/* find_closest_euclidean (cdcl .txx:541-654; rrl .txx:494-571 with exact NN instead of the
 * Voronoi volume).  Shuffle moving_, take the first number_points_, map, nearest fixed, w = 1. */
static void fwd_find_closest_euclidean(fwd_est *f) {
  if (f->explicit_order) {
    memcpy(f->order, f->explicit_order, sizeof(int64_t) * (size_t)f->explicit_n);
    free(f->explicit_order);
    f->explicit_order = NULL;
  } else {
    lrr_oracle_shuffle(f->order, f->moving.n, &f->rng);
  }
  fwd_reserve(f, f->number_points);
  f->nmatch = 0;
  f->from_is_copy = 0;
  for (int64_t i = 0; i < f->number_points; ++i) {
    int64_t p = f->order[i];
    double y[3];
    trans_map(&f->tr, f->moving.loc + 3 * p, y);
    int64_t q = f->vor ? vor_lookup(f->vor, y) : (f->kd ? nn_kd(f->kd, y) : nn_brute(f->fixed.loc, f->fixed.n, y));
    if (q < 0) continue; /* indices.size() > 0 guard (cdcl .txx:574) */
    f->m_from[f->nmatch] = p;
    f->m_to[f->nmatch] = q;
    f->m_w[f->nmatch] = 1.0;
    ++f->nmatch;
  }
}

/* RMS_error (cdcl .txx:196-230).  rrl .txx:211-241 never increments `count`, so it returns
 * the raw sum of squared residuals. */
static double fwd_RMS_error(const fwd_est *f) {
  double error = 0.0;
  int64_t count = 0;
  for (int64_t i = 0; i < f->nmatch; ++i) {
    error += fwd_residual2(f, i);
    if (!f->rrl) count += 1;
  }
  if (count != 0) error = sqrt(error / (double)count);
  return error;
}

/* median_error (cdcl .txx:159-190; rrl .txx:177-203): element (n-1)/2 + 1 of the sorted sqrt
 * residuals (vcl_nth_element at begin + (n-1)/2 + 1).  With fewer than two matches the reference
 * dereferences end(): 0 is returned here. */
static double fwd_median_error(const fwd_est *f) {
  int64_t n = f->nmatch;
  if (n < 2) return 0.0;
  double *errors = (double *)malloc(sizeof(double) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) errors[i] = sqrt(fwd_residual2(f, i));
  qsort(errors, (size_t)n, sizeof(double), cmp_double);
  double med = errors[(n - 1) / 2 + 1];
  free(errors);
  return med;
}

/* weighted_error (cdcl .txx:236-268; rrl .txx:247-276) */
static double fwd_weighted_error(const fwd_est *f) {
  double error = 0.0, sum_weight = 0.0;
  for (int64_t i = 0; i < f->nmatch; ++i) {
    double r2 = fwd_residual2(f, i);
    double w = f->m_w[i];
    error += sqrt(r2) * w;
    sum_weight += w;
  }
  if (sum_weight != 0) error /= sum_weight;
  return error;
}

/* estimate_scale_and_assign_weight (cdcl .txx:660-822; rrl .txx:577-718) */
static double fwd_estimate_scale_and_assign_weight(fwd_est *f) {
  int64_t n = f->nmatch;
  double *distances = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  double sum_weighted_error = 0.0, sum_weights = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    double r2 = fwd_residual2(f, i);
    if (f->scale_estimated) {
      sum_weighted_error += f->m_w[i] * r2;
      sum_weights += f->m_w[i];
    }
    f->m_w[i] = r2; /* temporarily store the squared residual */
    distances[i] = sqrt(r2);
  }
  double scale;
  if (!f->scale_estimated) {
    /* the cdcl class also computes a MAD scale here and overwrites it (:716-725) */
    if (f->injected_scale > 0.0) scale = f->injected_scale;
    else scale = lrr_oracle_muse_scale(distances, n);
    f->scale_estimated = 1;
  } else {
    scale = sqrt(sum_weighted_error / sum_weights);
  }
  scale = f->scale_constant * scale;
  for (int64_t i = 0; i < n; ++i) {
    double scale2 = scale * scale;
    f->m_w[i] = lrr_oracle_weight_BT(f->m_w[i] / scale2) / scale2;
  }
  free(distances);
  f->last_scale = scale;
  return scale;
}

/* a flat list of matches on which the re-weighting and accumulation steps operate;
 * needed because the cdcl symmetric class appends role-reversed clones of the
 * backward matches to the forward list (add_constraints). */
typedef struct {
  int64_t n;
  double *from_loc, *to_loc, *to_proj, *s_from, *s_to, *w;
} flat_list;

static void flat_alloc(flat_list *L, int64_t n) {
  size_t m = (size_t)(n > 0 ? n : 1);
  L->n = 0;
  L->from_loc = (double *)malloc(sizeof(double) * 3 * m);
  L->to_loc = (double *)malloc(sizeof(double) * 3 * m);
  L->to_proj = (double *)malloc(sizeof(double) * 9 * m);
  L->s_from = (double *)malloc(sizeof(double) * m);
  L->s_to = (double *)malloc(sizeof(double) * m);
  L->w = (double *)malloc(sizeof(double) * m);
}
static void flat_free(flat_list *L) {
  free(L->from_loc); free(L->to_loc); free(L->to_proj); free(L->s_from); free(L->s_to); free(L->w);
}
/* append f's matches (current, possibly normalised, in-place state); reversed = add_constraints role swap */
static void flat_append(flat_list *L, const fwd_est *f, int reversed) {
  for (int64_t i = 0; i < f->nmatch; ++i) {
    int64_t k = L->n++;
    int64_t p = f->m_from[i], q = f->m_to[i];
    const double *pl = fwd_from_loc(f, i), *ql = f->fixed.loc + 3 * q;
    if (!reversed) {
      memcpy(L->from_loc + 3 * k, pl, 24); memcpy(L->to_loc + 3 * k, ql, 24);
      memcpy(L->to_proj + 9 * k, f->fixed.proj + 9 * q, 72);
      L->s_from[k] = f->moving.strength[p]; L->s_to[k] = f->fixed.strength[q];
    } else {
      memcpy(L->from_loc + 3 * k, ql, 24); memcpy(L->to_loc + 3 * k, pl, 24);
      memcpy(L->to_proj + 9 * k, f->moving.proj + 9 * p, 72);
      L->s_from[k] = f->fixed.strength[q]; L->s_to[k] = f->moving.strength[p];
    }
    L->w[k] = f->m_w[i];
  }
}

/* weight_by_strength() (cdcl .txx:826-862; rrl .txx:722-752).  For cdcl_feature_ICP the
 * strengths are float and the product is a float multiply (cdcl_feature_ICP.h:83). */
static void flat_weight_by_strength(flat_list *L, int float_product) {
  int64_t n = L->n, m = 0;
  double *g = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; ++i) {
    if (L->w[i] != 0) {
      double grad_w;
      if (float_product) grad_w = (double)((float)L->s_from[i] * (float)L->s_to[i]);
      else grad_w = L->s_from[i] * L->s_to[i];
      if (grad_w > 90000.0) grad_w = 90000.0;
      L->w[i] *= grad_w;
      g[m++] = grad_w;
    }
  }
  double grad_scale = lrr_oracle_mad_scale(g, m);
  for (int64_t i = 0; i < n; ++i) L->w[i] *= (1.0 / grad_scale);
  free(g);
}

/* weight_spatially() (cdcl .txx:973-1004; rrl .txx:756-778); to-locations are normalised */
static void flat_weight_spatially(flat_list *L) {
  int64_t n = L->n;
  double *sw = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; ++i) {
    const double *q = L->to_loc + 3 * i;
    double dist_to_center = sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
    double spatial_weight = 1.0 / (1.0 + 5.0 * dist_to_center * dist_to_center);
    L->w[i] *= spatial_weight;
    sw[i] = spatial_weight;
  }
  double spatial_scale = lrr_oracle_mad_scale(sw, n);
  for (int64_t i = 0; i < n; ++i) L->w[i] *= (1.0 / spatial_scale);
  free(sw);
}

/* AtA += J^T P J w ; Atb += J^T P b w, generic fixed-matrix products as the reference writes
 * them (cdcl .txx:1039-1069; cdcl sym .txx:575-628).  b = q - p for the rigid increment
 * (:1051-1054), q otherwise. */
static void flat_accumulate_generic(const flat_list *L, const trans *T, int rigid_increment, double *AtA,
                                    double *Atb) {
  int dof = T->dof;
  double J[3 * MAXDOF], M[MAXDOF * 3];
  for (int64_t i = 0; i < L->n; ++i) {
    const double *p = L->from_loc + 3 * i, *q = L->to_loc + 3 * i, *P = L->to_proj + 9 * i;
    double w = L->w[i], b[3];
    trans_jac(T, p, J);
    if (rigid_increment) { b[0] = q[0] - p[0]; b[1] = q[1] - p[1]; b[2] = q[2] - p[2]; }
    else { b[0] = q[0]; b[1] = q[1]; b[2] = q[2]; }
    for (int a = 0; a < dof; ++a)
      for (int c = 0; c < 3; ++c)
        M[a * 3 + c] = (J[0 * dof + a] * P[0 * 3 + c] + J[1 * dof + a] * P[1 * 3 + c]) + J[2 * dof + a] * P[2 * 3 + c];
    for (int a = 0; a < dof; ++a) {
      for (int c = 0; c < dof; ++c) {
        double v = (M[a * 3] * J[c] + M[a * 3 + 1] * J[dof + c]) + M[a * 3 + 2] * J[2 * dof + c];
        AtA[a * dof + c] += v * w;
      }
      double v = (M[a * 3] * b[0] + M[a * 3 + 1] * b[1]) + M[a * 3 + 2] * b[2];
      Atb[a] += v * w;
    }
  }
}

/* rrl symmetric accumulation (rrl/rrl_estimation_symmetric_ICP_matching_all.txx:441-542, 546-655):
 * projector pre-multiplied by w, matches with w < 1e-4 skipped, upper triangle only. */
static void flat_accumulate_rrl_upper(const flat_list *L, const trans *T, double *AtA, double *Atb) {
  int dof = T->dof;
  double J[3 * MAXDOF], M[MAXDOF * 3], Pw[9];
  for (int64_t i = 0; i < L->n; ++i) {
    double w = L->w[i];
    if (w < 1e-4) continue;
    const double *p = L->from_loc + 3 * i, *q = L->to_loc + 3 * i, *P = L->to_proj + 9 * i;
    trans_jac(T, p, J);
    for (int k = 0; k < 9; ++k) Pw[k] = P[k] * w;
    for (int a = 0; a < dof; ++a)
      for (int c = 0; c < 3; ++c) {
        /* J^T has one non-zero per row for the affine Jacobian; the products with the zeros
         * are exact zeros, so the generic form equals the hand-unrolled one bit for bit */
        double s = 0.0;
        int first = 1;
        for (int r = 0; r < 3; ++r) {
          double j = J[r * dof + a];
          if (j == 0.0 && dof == 12) continue;
          double term = j * Pw[r * 3 + c];
          s = first ? term : s + term;
          first = 0;
        }
        M[a * 3 + c] = s;
      }
    for (int a = 0; a < dof; ++a) {
      for (int c = a; c < dof; ++c) {
        double accum = M[a * 3] * J[c];
        for (int k = 1; k < 3; ++k) accum += M[a * 3 + k] * J[k * dof + c];
        AtA[a * dof + c] += accum;
      }
      double accum = M[a * 3] * q[0];
      for (int k = 1; k < 3; ++k) accum += M[a * 3 + k] * q[k];
      Atb[a] += accum;
    }
  }
}

/* cdcl_normalize_matches (cdcl .txx:1159-1240; rrl .txx:864-943): centre/radius of from_
 * over matches and of the DISTINCT matched to_; applied in place on the shared objects. */
static void fwd_normalize_matches(fwd_est *f, double *cm, double *rm, double *cf, double *rf) {
  int64_t n = f->nmatch, nd = 0;
  cm[0] = cm[1] = cm[2] = cf[0] = cf[1] = cf[2] = 0.0;
  *rm = *rf = 0.0;
  memset(f->flag, 0, (size_t)f->fixed.n);
  for (int64_t i = 0; i < n; ++i) {
    const double *p = fwd_from_loc(f, i);
    cm[0] += p[0]; cm[1] += p[1]; cm[2] += p[2];
    int64_t q = f->m_to[i];
    if (!f->flag[q]) {
      f->flag[q] = 1; ++nd;
      const double *ql = f->fixed.loc + 3 * q;
      cf[0] += ql[0]; cf[1] += ql[1]; cf[2] += ql[2];
    }
  }
  for (int d = 0; d < 3; ++d) { cm[d] /= (double)n; cf[d] /= (double)nd; }
  for (int64_t i = 0; i < n; ++i) {
    double *p = fwd_from_loc_mut(f, i);
    p[0] -= cm[0]; p[1] -= cm[1]; p[2] -= cm[2];
    *rm += sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
  }
  *rm /= (double)n;
  for (int64_t q = 0; q < f->fixed.n; ++q) {
    if (!f->flag[q]) continue;
    double *ql = f->fixed.loc + 3 * q;
    ql[0] -= cf[0]; ql[1] -= cf[1]; ql[2] -= cf[2];
    *rf += sqrt((ql[0] * ql[0] + ql[1] * ql[1]) + ql[2] * ql[2]);
  }
  *rf /= (double)nd;
  for (int64_t i = 0; i < n; ++i) {
    double *p = fwd_from_loc_mut(f, i);
    p[0] /= *rm; p[1] /= *rm; p[2] /= *rm;
  }
  for (int64_t q = 0; q < f->fixed.n; ++q) {
    if (!f->flag[q]) continue;
    double *ql = f->fixed.loc + 3 * q;
    ql[0] /= *rf; ql[1] /= *rf; ql[2] /= *rf;
  }
}

/* cdcl_normalize_matches_known (cdcl sym .txx:869-908; rrl sym .txx:1093-1129) */
static void fwd_normalize_matches_known(fwd_est *f, const double *cm, double rm, const double *cf, double rf) {
  memset(f->flag, 0, (size_t)f->fixed.n);
  for (int64_t i = 0; i < f->nmatch; ++i) {
    double *p = fwd_from_loc_mut(f, i);
    for (int d = 0; d < 3; ++d) { p[d] -= cm[d]; p[d] /= rm; }
    int64_t q = f->m_to[i];
    if (!f->flag[q]) {
      f->flag[q] = 1;
      double *ql = f->fixed.loc + 3 * q;
      for (int d = 0; d < 3; ++d) { ql[d] -= cf[d]; ql[d] /= rf; }
    }
  }
}

/* cdcl_unnormalize_matches (cdcl .txx:1245-1286; rrl .txx:948-986) */
static void fwd_unnormalize_matches(fwd_est *f, const double *cm, double rm, const double *cf, double rf) {
  memset(f->flag, 0, (size_t)f->fixed.n);
  for (int64_t i = 0; i < f->nmatch; ++i) {
    double *p = fwd_from_loc_mut(f, i);
    for (int d = 0; d < 3; ++d) { p[d] *= rm; p[d] += cm[d]; }
    int64_t q = f->m_to[i];
    if (!f->flag[q]) {
      f->flag[q] = 1;
      double *ql = f->fixed.loc + 3 * q;
      for (int d = 0; d < 3; ++d) { ql[d] *= rf; ql[d] += cf[d]; }
    }
  }
}

/* rigid pre-map (cdcl .txx:455-484; cdcl sym .txx:400-456): replace every from_ by a new
 * feature at G(p); trans_ becomes the identity increment carrying G's covariance. */
static void fwd_rigid_premap(fwd_est *f, trans *global_out) {
  *global_out = f->tr;
  trans inc;
  trans_identity(&inc, 6);
  memcpy(inc.covar, global_out->covar, sizeof inc.covar);
  for (int64_t i = 0; i < f->nmatch; ++i) {
    double y[3];
    trans_map(global_out, fwd_from_loc(f, i), y);
    memcpy(f->m_from_loc + 3 * i, y, 24);
  }
  if (f->from_is_copy) { /* copies pre-mapped again: their covariance carries both rotations */
    double both[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        both[3 * r + c] = (global_out->A[3 * r] * f->premap_R[c] + global_out->A[3 * r + 1] * f->premap_R[3 + c]) + global_out->A[3 * r + 2] * f->premap_R[6 + c];
    memcpy(f->premap_R, both, sizeof both);
  } else {
    memcpy(f->premap_R, global_out->A, sizeof f->premap_R);
  }
  f->from_is_copy = 1;
  double zero6[6] = { 0, 0, 0, 0, 0, 0 };
  trans_set_param(&inc, zero6);
  f->tr = inc;
}

typedef struct {
  double AtA[MAXDOF * MAXDOF], Atb[MAXDOF];
} normal_eq;

static void solve_and_set(trans *T, const double *AtA, const double *Atb, int estimate_parameters) {
  int dof = T->dof;
  double inv[MAXDOF * MAXDOF], par[MAXDOF];
  lrr_oracle_sym_pinv(dof, AtA, inv);
  memcpy(T->covar, inv, sizeof(double) * (size_t)(dof * dof)); /* set_covariance(invAA) */
  if (estimate_parameters) {
    for (int i = 0; i < dof; ++i) {
      double s = 0.0;
      for (int k = 0; k < dof; ++k) s += inv[i * dof + k] * Atb[k];
      par[i] = s;
    }
    trans_set_param(T, par);
  }
}

/* estimate_LS of the forward classes (cdcl .txx:1009-1093; rrl .txx:783-859) */
static void fwd_estimate_LS(fwd_est *f, int estimate_parameters, normal_eq *ne) {
  double cm[3] = { 0, 0, 0 }, cf[3] = { 0, 0, 0 }, rm = 0.0, rf = 0.0;
  if (f->normalize_matches) {
    fwd_normalize_matches(f, cm, &rm, cf, &rf);
    trans_normalize(&f->tr, rm, rf, cm, cf);
  }
  fwd_estimate_scale_and_assign_weight(f); /* in normalised coordinates here */
  flat_list L;
  flat_alloc(&L, f->nmatch);
  flat_append(&L, f, 0);
  if (f->weight_by_strength) flat_weight_by_strength(&L, f->rrl);
  memset(ne, 0, sizeof *ne);
  /* the rrl class has no rigid branch: b = q always (rrl .txx:822-829) */
  flat_accumulate_generic(&L, &f->tr, !f->rrl && f->dof == 6, ne->AtA, ne->Atb);
  memcpy(f->m_w, L.w, sizeof(double) * (size_t)f->nmatch);
  flat_free(&L);
  solve_and_set(&f->tr, ne->AtA, ne->Atb, estimate_parameters);
  if (f->normalize_matches) {
    trans_unnormalize(&f->tr, rm, rf, cm, cf);
    fwd_unnormalize_matches(f, cm, rm, cf, rf);
  }
}

/* cdcl_estimation::initialize (cdcl/cdcl_estimation.txx:70-84) at one resolution: the base class's find_closest_euclidean over ALL moving
 * features (cdcl_estimation_ICP.txx:451-489: one nearest fixed feature each, w = 1) and its estimate_LS() (:646-750, parameters not
 * estimated): residuals along the normal of the fixed feature, the eigenvector of the smallest eigenvalue of its covariance
 * (NORMAL_DISTANCES_FOR_ICP, :523-537 and :696-709), scale = 2 x MUSE (:557-582), Beaton-Tukey weights (:603-604), parameter
 * covariance = svd inverse of AtA (:727-731).  weight_by_strength_ stays false: nothing in the reference sets it on this class.
 * With normalize the reference takes the eigenvectors of covariance / r_f^2: the same vectors.
 * A, t, c: trans_, in and out — normalize() / unnormalize() around the fit hand the same map back about another centre (the moving
 * centroid, cdcl_trans_affine.txx:259-286, cdcl_trans_rigid3d.cxx:284-305); covar_out: dof x dof, belonging to that centre;
 * scale_out: the scale the weights were computed with. */
int lrr_oracle_cdc_em_initialize(int dof, const lrr_oracle_features *moving, const lrr_oracle_features *fixed, double *A,
                                 double *t, double *c, int normalize, double *covar_out, double *scale_out) {
  if ((dof != 6 && dof != 12) || !moving || !fixed || !fixed->cov || moving->n < 1 || fixed->n < 1) return -1;
  double *proj = (double *)malloc(sizeof(double) * 9 * (size_t)fixed->n);
  for (int64_t j = 0; j < fixed->n; ++j) {
    double nrm[3], big[3];
    sym_eig_ends(fixed->cov + 9 * j, nrm, big);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) proj[9 * j + 3 * a + b] = nrm[a] * nrm[b];
  }
  lrr_oracle_features fx = *fixed;
  fx.proj = proj;
  trans T;
  trans_from_Atc(&T, dof, A, t, c);
  fwd_est *f = fwd_create(0, dof, moving, &fx, &T, 1, LRR_ORACLE_NN_KDTREE);
  f->number_points = moving->n;
  f->normalize_matches = normalize;
  f->scale_constant = 2.0;
  f->explicit_order = (int64_t *)malloc(sizeof(int64_t) * (size_t)moving->n);  /* moving_[res_] in its own order */
  for (int64_t i = 0; i < moving->n; ++i) f->explicit_order[i] = i;
  f->explicit_n = moving->n;
  fwd_find_closest_euclidean(f);
  normal_eq ne;
  fwd_estimate_LS(f, 0, &ne);
  memcpy(covar_out, f->tr.covar, sizeof(double) * (size_t)(dof * dof));
  memcpy(A, f->tr.A, sizeof f->tr.A);
  memcpy(t, f->tr.t, sizeof f->tr.t);
  memcpy(c, f->tr.c, sizeof f->tr.c);
  if (scale_out) *scale_out = f->last_scale;
  fwd_free(f);
  free(proj);
  return 0;
}

/* one_iteration of the forward classes (cdcl .txx:434-509; rrl .txx:427-462) */
static int fwd_one_iteration(fwd_est *f, normal_eq *ne) {
  fwd_find_closest_euclidean(f);
  double RMS_before = fwd_RMS_error(f);
  trans global;
  int rigid = (!f->rrl && f->dof == 6); /* is_type(cdcl_trans_rigid3d), cdcl class only */
  if (rigid) fwd_rigid_premap(f, &global);
  fwd_estimate_LS(f, 1, ne);
  if (rigid) {
    trans_recompose_increment(&global, &f->tr);
    memcpy(global.covar, f->tr.covar, sizeof global.covar);
    f->tr = global;
  }
  /* NB rigid: matches_[i]->from_ are still the pre-mapped copies -> mapped twice (cdcl .txx:499 vs :477) */
  double RMS_after = fwd_RMS_error(f);
  return fabs(RMS_before - RMS_after) < 1e-4;
}

/* sheet_angles / tube_angles of the rrl class (rrl .txx:281-375): mapped directions_ */
static double fwd_angles_rrl(const fwd_est *f, int tube) {
  double avg = 0.0, sum_weight = 0.0;
  if (!f->moving.dirs || !f->fixed.dirs) return NAN;
  for (int64_t i = 0; i < f->nmatch; ++i) {
    int64_t p = f->m_from[i], q = f->m_to[i];
    uint8_t want = tube ? 1 : 2;
    if (f->moving.shape[p] != want) continue;
    if (f->fixed.shape[q] != want) continue;
    const double *pl = fwd_from_loc(f, i);
    const double *pd = f->moving.dirs + 6 * p, *qd = f->fixed.dirs + 6 * q;
    double dirp[3], dirq[3];
    if (tube) { /* vnl_cross_3d(directions_[0], directions_[1]) */
      dirp[0] = pd[1] * pd[5] - pd[2] * pd[4]; dirp[1] = pd[2] * pd[3] - pd[0] * pd[5]; dirp[2] = pd[0] * pd[4] - pd[1] * pd[3];
      dirq[0] = qd[1] * qd[5] - qd[2] * qd[4]; dirq[1] = qd[2] * qd[3] - qd[0] * qd[5]; dirq[2] = qd[0] * qd[4] - qd[1] * qd[3];
    } else {
      dirp[0] = pd[0]; dirp[1] = pd[1]; dirp[2] = pd[2];
      dirq[0] = qd[0]; dirq[1] = qd[1]; dirq[2] = qd[2];
    }
    double y[3], x2[3] = { pl[0] + dirp[0], pl[1] + dirp[1], pl[2] + dirp[2] }, y2[3];
    trans_map(&f->tr, pl, y);
    trans_map(&f->tr, x2, y2);
    double m[3] = { y2[0] - y[0], y2[1] - y[1], y2[2] - y[2] };
    double nrm = sqrt((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]);
    m[0] /= nrm; m[1] /= nrm; m[2] /= nrm;
    double a = acos((m[0] * dirq[0] + m[1] * dirq[1]) + m[2] * dirq[2]);
    if (a > M_PI / 2) a = M_PI - a;
    double w = f->m_w[i];
    avg += w * a;
    sum_weight += w;
  }
  return avg / sum_weight;
}

/* sheet_angles / tube_angles of the cdcl class (cdcl .txx:273-382): angle between the covariance eigenvectors of the
 * smallest eigenvalue (sheets, column 0) / the largest (tubes, column 2), no mapping by the transform; a pre-mapped
 * from_ carries R C R^T (cdcl .txx:473-479), whose eigenvector is R v */
static double fwd_angles_cdcl(const fwd_est *f, int tube) {
  double avg = 0.0, sum_weight = 0.0;
  if (!f->moving.evec || !f->fixed.evec) return NAN;
  for (int64_t i = 0; i < f->nmatch; ++i) {
    int64_t p = f->m_from[i], q = f->m_to[i];
    uint8_t want = tube ? 1 : 2;
    if (f->moving.shape[p] != want) continue;
    if (f->fixed.shape[q] != want) continue;
    const double *vp = f->moving.evec + 6 * p + (tube ? 3 : 0), *vq = f->fixed.evec + 6 * q + (tube ? 3 : 0);
    double d[3] = { vp[0], vp[1], vp[2] };
    if (f->from_is_copy) {
      const double *R = f->premap_R;
      d[0] = (R[0] * vp[0] + R[1] * vp[1]) + R[2] * vp[2];
      d[1] = (R[3] * vp[0] + R[4] * vp[1]) + R[5] * vp[2];
      d[2] = (R[6] * vp[0] + R[7] * vp[1]) + R[8] * vp[2];
    }
    double a = acos((d[0] * vq[0] + d[1] * vq[1]) + d[2] * vq[2]);
    if (a > M_PI / 2) a = M_PI - a;
    double w = f->m_w[i];
    avg += w * a;
    sum_weight += w;
  }
  return avg / sum_weight;
}
static double fwd_angles(const fwd_est *f, int tube) {
  /* the cdcl classes on cdcl_feature_with_shape compare covariance eigenvectors, the rrl classes mapped directions_ */
  return (!f->rrl && f->moving.evec && f->fixed.evec) ? fwd_angles_cdcl(f, tube) : fwd_angles_rrl(f, tube);
}

/* compute_transfer_error_covariance (cdcl .txx:388-404; rrl .txx:381-397) */
static void fwd_transfer_error_covariance(fwd_est *f) {
  int dof = f->dof;
  double acc[9] = { 0 }, J[3 * MAXDOF], JC[3 * MAXDOF];
  for (int64_t i = 0; i < f->nmatch; ++i) {
    trans_jac(&f->tr, fwd_from_loc(f, i), J);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < dof; ++c) {
        double s = 0.0;
        for (int k = 0; k < dof; ++k) s += J[r * dof + k] * f->tr.covar[k * dof + c];
        JC[r * dof + c] = s;
      }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        for (int k = 0; k < dof; ++k) s += JC[r * dof + k] * J[c * dof + k];
        acc[r * 3 + c] += s;
      }
  }
  for (int k = 0; k < 9; ++k) f->tr.covarJ[k] = acc[k] * (1.0 / (double)f->nmatch);
}

/* ------------------------------------------------------------------------- */
/* estimator handle (forward or symmetric)                                   */
/* ------------------------------------------------------------------------- */

struct lrr_oracle_est {
  int kind, dof;
  fwd_est *fwd, *bwd;      /* forward_, backward_ (bwd NULL for forward kinds) */
  /* symmetric state (cdcl sym .h / rrl sym .h:208-224) */
  int normalize_matches;
  unsigned oscillation_count;
  double error_difference, weighted_error;
  int64_t number_matches;
  int finest_level;
  normal_eq ne;
  double inj_fwd, inj_bwd;
};

static int is_sym(const lrr_oracle_est *e) { return e->kind >= 2; }
static int is_rrl(const lrr_oracle_est *e) { return e->kind == 1 || e->kind == 3; }

lrr_oracle_est *lrr_oracle_create(int kind, int dof, const lrr_oracle_features *moving,
                                  const lrr_oracle_features *fixed,
                                  const lrr_oracle_features *moving_inside,
                                  const lrr_oracle_features *fixed_inside, const double *A,
                                  const double *t, const double *c, uint64_t seed, int nn_backend) {
  if (dof != 6 && dof != 12) return NULL;
  lrr_oracle_est *e = (lrr_oracle_est *)calloc(1, sizeof *e);
  e->kind = kind; e->dof = dof;
  trans T;
  trans_from_Atc(&T, dof, A, t, c);
  int rrl = (kind == 1 || kind == 3);
  if (kind < 2) {
    e->fwd = fwd_create(rrl, dof, moving, fixed, &T, seed, nn_backend);
  } else {
    /* forward_ = ICP(moving_inside, fixed, trans); backward_ = ICP(fixed_inside, moving, trans_inverse)
     * (rrl sym .txx:211-217; cdcl sym .txx:157-163); both weight_by_strength(true) + normalize_matches() */
    trans Ti;
    trans_inverse(&T, &Ti);
    e->fwd = fwd_create(rrl, dof, moving_inside, fixed, &T, seed, nn_backend);
    e->bwd = fwd_create(rrl, dof, fixed_inside, moving, &Ti, seed ^ 0xD1B54A32D192ED03ULL, nn_backend);
    e->fwd->weight_by_strength = e->bwd->weight_by_strength = 1;
    e->fwd->normalize_matches = e->bwd->normalize_matches = 1;
    e->normalize_matches = 1;
    e->number_matches = 1000;
  }
  e->inj_fwd = e->inj_bwd = -1.0;
  return e;
}

void lrr_oracle_destroy(lrr_oracle_est *e) {
  if (!e) return;
  fwd_free(e->fwd); fwd_free(e->bwd); free(e);
}

/* set_transform (forward: cdcl .txx:139-153; symmetric: rrl sym .txx:303-315, cdcl sym .txx:230-242) */
void lrr_oracle_set_transform(lrr_oracle_est *e, const double *A, const double *t, const double *c) {
  trans T;
  trans_from_Atc(&T, e->dof, A, t, c);
  fwd_set_transform(e->fwd, &T);
  e->fwd->injected_scale = e->inj_fwd;
  if (is_sym(e)) {
    trans Ti;
    trans_inverse(&T, &Ti);
    fwd_set_transform(e->bwd, &Ti);
    e->bwd->injected_scale = e->inj_bwd;
    e->oscillation_count = 0;
    e->error_difference = 0.0;
    e->number_matches = 1000;
    e->finest_level = 0;
  }
}

/* set_features (rrl sym .txx:268-289; cdcl sym .txx:205-226) */
void lrr_oracle_set_features(lrr_oracle_est *e, const lrr_oracle_features *moving_inside,
                             const lrr_oracle_features *fixed_inside) {
  fwd_set_moving_data(e->fwd, moving_inside);
  if (is_sym(e)) {
    fwd_set_moving_data(e->bwd, fixed_inside);
    e->oscillation_count = 0;
    e->error_difference = 0.0;
    e->number_matches = 1000;
    e->finest_level = 0;
  }
}

/* test hook (as the product's set_subset_seed): restarts the generators of the subset permutations, so that one estimator
 * object re-used over many query locations (set_features + set_transform, rrl/itkLocationRegistration.cxx:1635-1637) gives
 * every location the keys a freshly built estimator with that seed would use */
void lrr_oracle_set_subset_seed(lrr_oracle_est *e, uint64_t seed) {
  e->fwd->rng = seed;
  if (e->bwd) e->bwd->rng = seed ^ 0xD1B54A32D192ED03ULL;
}

void lrr_oracle_weight_by_strength(lrr_oracle_est *e, int flag) {
  e->fwd->weight_by_strength = flag;
  if (e->bwd) e->bwd->weight_by_strength = flag;
}
void lrr_oracle_normalize_matches(lrr_oracle_est *e) {
  e->fwd->normalize_matches = 1;
  if (e->bwd) { e->bwd->normalize_matches = 1; e->normalize_matches = 1; }
}
void lrr_oracle_set_number_points(lrr_oracle_est *e, int64_t n) {
  int a = fwd_set_number_points(e->fwd, n), b = 1;
  if (e->bwd) b = fwd_set_number_points(e->bwd, n);
  if (is_sym(e)) { e->number_matches = n; e->finest_level = a && b; }
}
void lrr_oracle_inject_initial_scale(lrr_oracle_est *e, double fwd, double bwd) {
  e->inj_fwd = fwd; e->inj_bwd = bwd;
  e->fwd->injected_scale = fwd;
  if (e->bwd) e->bwd->injected_scale = bwd;
}
void lrr_oracle_set_subset_order(lrr_oracle_est *e, int dir, const int64_t *order, int64_t n) {
  fwd_est *f = dir ? e->bwd : e->fwd;
  if (!f || n != f->moving.n) return;
  free(f->explicit_order);
  f->explicit_order = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  memcpy(f->explicit_order, order, sizeof(int64_t) * (size_t)n);
  f->explicit_n = n;
}

void lrr_oracle_find_closest_euclidean(lrr_oracle_est *e) {
  fwd_find_closest_euclidean(e->fwd);
  if (e->bwd) fwd_find_closest_euclidean(e->bwd);
}

/* rrl symmetric estimate_LS (rrl/rrl_estimation_symmetric_ICP_matching_all.txx:699-918) */
static void sym_estimate_LS_rrl(lrr_oracle_est *e, fwd_est *F, fwd_est *B, int estimate_parameters) {
  fwd_estimate_scale_and_assign_weight(F); /* un-normalised coordinates */
  fwd_estimate_scale_and_assign_weight(B);
  double cm[3] = { 0, 0, 0 }, cf[3] = { 0, 0, 0 }, rm = 0.0, rf = 0.0;
  if (e->normalize_matches) {
    fwd_normalize_matches(F, cm, &rm, cf, &rf);
    trans_normalize(&F->tr, rm, rf, cm, cf);
    fwd_normalize_matches_known(B, cf, rf, cm, rm);
    trans_normalize(&B->tr, rf, rm, cf, cm);
  }
  flat_list LF, LB;
  flat_alloc(&LF, F->nmatch); flat_append(&LF, F, 0);
  flat_alloc(&LB, B->nmatch); flat_append(&LB, B, 0);
  if (F->weight_by_strength) flat_weight_by_strength(&LF, F->rrl);
  if (B->weight_by_strength) flat_weight_by_strength(&LB, B->rrl);
  flat_weight_spatially(&LF);
  flat_weight_spatially(&LB);
  memcpy(F->m_w, LF.w, sizeof(double) * (size_t)F->nmatch);
  memcpy(B->m_w, LB.w, sizeof(double) * (size_t)B->nmatch);
  memset(&e->ne, 0, sizeof e->ne);
  flat_accumulate_rrl_upper(&LF, &F->tr, e->ne.AtA, e->ne.Atb);
  /* backward term: J of the FORWARD transform at q_b (backward to_), P of p_b (backward from_),
   * b = p_b  (:557-646).  That is the role-reversed list. */
  flat_list LR;
  flat_alloc(&LR, B->nmatch); flat_append(&LR, B, 1);
  memcpy(LR.w, LB.w, sizeof(double) * (size_t)B->nmatch);
  flat_accumulate_rrl_upper(&LR, &F->tr, e->ne.AtA, e->ne.Atb);
  flat_free(&LR); flat_free(&LF); flat_free(&LB);
  int dof = e->dof;
  for (int i = 0; i < dof; ++i)
    for (int j = i + 1; j < dof; ++j) e->ne.AtA[j * dof + i] = e->ne.AtA[i * dof + j];
  solve_and_set(&F->tr, e->ne.AtA, e->ne.Atb, estimate_parameters);
  if (e->normalize_matches) {
    trans_unnormalize(&F->tr, rm, rf, cm, cf);
    fwd_unnormalize_matches(F, cm, rm, cf, rf);
    trans_unnormalize(&B->tr, rf, rm, cf, cm);
    fwd_unnormalize_matches(B, cf, rf, cm, rm);
  }
  if (estimate_parameters) trans_inverse(&F->tr, &B->tr);
}

/* cdcl symmetric estimate_LS (cdcl/cdcl_estimation_symmetric_ICP_matching_all.txx:398-701) */
static void sym_estimate_LS_cdcl(lrr_oracle_est *e, fwd_est *F, fwd_est *B, int estimate_parameters) {
  trans Gf, Gb;
  int rigid = (e->dof == 6);
  if (rigid) { fwd_rigid_premap(F, &Gf); fwd_rigid_premap(B, &Gb); }
  fwd_estimate_scale_and_assign_weight(F);
  fwd_estimate_scale_and_assign_weight(B);
  double cm[3] = { 0, 0, 0 }, cf[3] = { 0, 0, 0 }, rm = 0.0, rf = 0.0;
  if (e->normalize_matches) {
    fwd_normalize_matches(F, cm, &rm, cf, &rf);
    trans_normalize(&F->tr, rm, rf, cm, cf);
    fwd_normalize_matches_known(B, cf, rf, cm, rm);
    trans_normalize(&B->tr, rf, rm, cf, cm);
  }
  /* add_constraints (:367-393, 541-542): append role-reversed clones of the backward matches */
  flat_list L;
  flat_alloc(&L, F->nmatch + B->nmatch);
  flat_append(&L, F, 0);
  flat_append(&L, B, 1);
  if (F->weight_by_strength) flat_weight_by_strength(&L, F->rrl);
  flat_weight_spatially(&L);
  memset(&e->ne, 0, sizeof e->ne);
  flat_accumulate_generic(&L, &F->tr, rigid, e->ne.AtA, e->ne.Atb);
  memcpy(F->m_w, L.w, sizeof(double) * (size_t)F->nmatch); /* appended matches are erased (:665) */
  flat_free(&L);
  solve_and_set(&F->tr, e->ne.AtA, e->ne.Atb, estimate_parameters);
  if (e->normalize_matches) {
    trans_unnormalize(&F->tr, rm, rf, cm, cf);
    fwd_unnormalize_matches(F, cm, rm, cf, rf);
    trans_unnormalize(&B->tr, rf, rm, cf, cm);
    fwd_unnormalize_matches(B, cf, rf, cm, rm);
  }
  if (rigid) {
    trans_recompose_increment(&Gf, &F->tr);
    memcpy(Gf.covar, F->tr.covar, sizeof Gf.covar);
    F->tr = Gf;
  }
  if (estimate_parameters) trans_inverse(&F->tr, &B->tr);
  (void)Gb;
}

void lrr_oracle_estimate_LS(lrr_oracle_est *e, int estimate_parameters) {
  if (!is_sym(e)) fwd_estimate_LS(e->fwd, estimate_parameters, &e->ne);
  else if (is_rrl(e)) sym_estimate_LS_rrl(e, e->fwd, e->bwd, estimate_parameters);
  else sym_estimate_LS_cdcl(e, e->fwd, e->bwd, estimate_parameters);
}

/* cdcl_estimation_symmetric_ICP_matching_all::estimate_LS_backward (cdcl sym .txx:704-863): NOT estimate_LS with the
 * roles swapped.  Only the backward list is pre-mapped (rigid, :707-734); scale and Beaton-Tukey weights of both lists
 * (:737-741); the forward matches are appended to the backward list as role-reversed clones (:743-745) BEFORE
 * cdcl_normalize_matches runs on it (:753-757), so the statistics are those of the merged list and every clone counts as a
 * distinct to_; weight_by_strength of the merged list (:765), no spatial weighting; generic accumulation under the backward
 * transform (:778-812); covariance / parameters of the backward transform (:823-838); the clones are erased (:841);
 * un-normalise (:843-851); forward = inverse(backward) (:853-855) and only then, rigid, the increment is recomposed
 * (:860-866). */
static void sym_estimate_LS_backward_cdcl(lrr_oracle_est *e, fwd_est *F, fwd_est *B, int estimate_parameters) {
  trans Gb;
  int rigid = (e->dof == 6);
  if (rigid) fwd_rigid_premap(B, &Gb);
  fwd_estimate_scale_and_assign_weight(F);
  fwd_estimate_scale_and_assign_weight(B);
  flat_list L;
  flat_alloc(&L, F->nmatch + B->nmatch);
  flat_append(&L, B, 0);
  flat_append(&L, F, 1);
  double cm[3] = { 0, 0, 0 }, cf[3] = { 0, 0, 0 }, rm = 0.0, rf = 0.0;
  if (e->normalize_matches) {
    /* from_: every match; to_: the distinct objects among the backward list's own to_, every clone */
    int64_t nB = B->nmatch, n = L.n, nd = 0;
    memset(B->flag, 0, (size_t)B->fixed.n);
    for (int64_t i = 0; i < n; ++i) {
      const double *p = L.from_loc + 3 * i, *q = L.to_loc + 3 * i;
      cm[0] += p[0]; cm[1] += p[1]; cm[2] += p[2];
      int first = 1;
      if (i < nB) { first = !B->flag[B->m_to[i]]; B->flag[B->m_to[i]] = 1; }
      if (first) { ++nd; cf[0] += q[0]; cf[1] += q[1]; cf[2] += q[2]; }
    }
    for (int d = 0; d < 3; ++d) { cm[d] /= (double)n; cf[d] /= (double)nd; }
    memset(B->flag, 0, (size_t)B->fixed.n);
    for (int64_t i = 0; i < n; ++i) {
      double *p = L.from_loc + 3 * i, *q = L.to_loc + 3 * i;
      for (int d = 0; d < 3; ++d) { p[d] -= cm[d]; q[d] -= cf[d]; }
      rm += sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
      int first = 1;
      if (i < nB) { first = !B->flag[B->m_to[i]]; B->flag[B->m_to[i]] = 1; }
      if (first) rf += sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
    }
    rm /= (double)n;
    rf /= (double)nd;
    for (int64_t i = 0; i < n; ++i) {
      double *p = L.from_loc + 3 * i, *q = L.to_loc + 3 * i;
      for (int d = 0; d < 3; ++d) { p[d] /= rm; q[d] /= rf; }
    }
    trans_normalize(&B->tr, rm, rf, cm, cf);
  }
  if (B->weight_by_strength) flat_weight_by_strength(&L, B->rrl);
  memset(&e->ne, 0, sizeof e->ne);
  flat_accumulate_generic(&L, &B->tr, rigid, e->ne.AtA, e->ne.Atb);
  memcpy(B->m_w, L.w, sizeof(double) * (size_t)B->nmatch); /* the clones are erased (:841) */
  flat_free(&L);
  solve_and_set(&B->tr, e->ne.AtA, e->ne.Atb, estimate_parameters);
  if (e->normalize_matches) trans_unnormalize(&B->tr, rm, rf, cm, cf);
  if (estimate_parameters) trans_inverse(&B->tr, &F->tr);
  if (rigid) {
    trans_recompose_increment(&Gb, &B->tr);
    memcpy(Gb.covar, B->tr.covar, sizeof Gb.covar);
    B->tr = Gb;
  }
}

/* estimate_LS_backward: the rrl class swaps forward_/backward_, calls estimate_LS and swaps back (rrl sym .txx:921-934);
 * the cdcl class has its own variant (above) */
void lrr_oracle_estimate_LS_backward(lrr_oracle_est *e, int estimate_parameters) {
  if (e->kind == LRR_ORACLE_RRL_SYMMETRIC) sym_estimate_LS_rrl(e, e->bwd, e->fwd, estimate_parameters);
  else if (e->kind == LRR_ORACLE_CDCL_SYMMETRIC) sym_estimate_LS_backward_cdcl(e, e->fwd, e->bwd, estimate_parameters);
}

/* symmetric one_iteration (rrl sym .txx:341-438; cdcl sym .txx:268-363) */
static int sym_one_iteration(lrr_oracle_est *e, unsigned iteration) {
  double weighted_error_before = fwd_weighted_error(e->fwd);
  lrr_oracle_find_closest_euclidean(e);
  lrr_oracle_estimate_LS(e, 1);
  double weighted_error_after = fwd_weighted_error(e->fwd);
  double curr_error_difference = weighted_error_after - weighted_error_before;
  double diff = (weighted_error_after - weighted_error_before) / weighted_error_after;
  int converged = fabs(diff) < 1e-4;
  e->weighted_error = weighted_error_after;
  /* the rrl class applies the oscillation logic only when iteration > 0 (:387) */
  if (!is_rrl(e) || iteration > 0) {
    if (!e->finest_level && (e->oscillation_count > 1 || (fabs(diff) < 0.01))) converged = 1;
    else if (!converged) {
      if (weighted_error_after > 1.5) {
        if (curr_error_difference * e->error_difference < 0.0) ++e->oscillation_count;
        else if (e->oscillation_count > 0) --e->oscillation_count;
        if (e->oscillation_count > 10) converged = 1; /* oscillated() */
      } else {
        if (curr_error_difference > 0.0) ++e->oscillation_count;
        if (e->oscillation_count > 3) converged = 1;
      }
    }
  }
  e->error_difference = curr_error_difference;
  if (converged) {
    if (!e->finest_level) {
      converged = 0;
      e->number_matches *= 2;
      int fmax = fwd_set_number_points(e->fwd, e->number_matches);
      int bmax = fwd_set_number_points(e->bwd, e->number_matches);
      e->finest_level = fmax && bmax;
      e->oscillation_count = 0;
      e->error_difference = 0.0;
    }
  }
  return converged && e->finest_level;
}

int lrr_oracle_one_iteration(lrr_oracle_est *e, unsigned iteration) {
  if (is_sym(e)) return sym_one_iteration(e, iteration);
  return fwd_one_iteration(e->fwd, &e->ne);
}

/* run(): forward classes loop <= 30 iterations (cdcl .txx:413-428); the symmetric classes'
 * run() is a stub in the reference, the caller loops <= 70 (itkLocationRegistration.cxx:2418). */
int lrr_oracle_run(lrr_oracle_est *e, unsigned max_iterations, unsigned *iterations_done) {
  int converged = 0;
  unsigned it = 0;
  while (!converged && it < max_iterations) {
    converged = lrr_oracle_one_iteration(e, it);
    ++it;
  }
  if (iterations_done) *iterations_done = it;
  return converged;
}

void lrr_oracle_get_transform(lrr_oracle_est *e, int dir, double *A, double *t, double *c, double *covar) {
  fwd_est *f = dir ? e->bwd : e->fwd;
  if (!f) return;
  if (A) memcpy(A, f->tr.A, sizeof f->tr.A);
  if (t) memcpy(t, f->tr.t, sizeof f->tr.t);
  if (c) memcpy(c, f->tr.c, sizeof f->tr.c);
  if (covar) memcpy(covar, f->tr.covar, sizeof(double) * (size_t)(e->dof * e->dof));
}

double lrr_oracle_RMS_error(lrr_oracle_est *e) { return fwd_RMS_error(e->fwd); }
double lrr_oracle_median_error(lrr_oracle_est *e) { return fwd_median_error(e->fwd); }
double lrr_oracle_weighted_error(lrr_oracle_est *e, int dir) {
  fwd_est *f = dir ? e->bwd : e->fwd;
  return f ? fwd_weighted_error(f) : NAN;
}
double lrr_oracle_sheet_angles(lrr_oracle_est *e, int dir) {
  fwd_est *f = dir ? e->bwd : e->fwd;
  return f ? fwd_angles(f, 0) : NAN;
}
double lrr_oracle_tube_angles(lrr_oracle_est *e, int dir) {
  fwd_est *f = dir ? e->bwd : e->fwd;
  return f ? fwd_angles(f, 1) : NAN;
}
void lrr_oracle_compute_transfer_error_covariance(lrr_oracle_est *e, double *cf, double *cb) {
  fwd_transfer_error_covariance(e->fwd);
  if (cf) memcpy(cf, e->fwd->tr.covarJ, sizeof e->fwd->tr.covarJ);
  if (e->bwd) {
    fwd_transfer_error_covariance(e->bwd);
    if (cb) memcpy(cb, e->bwd->tr.covarJ, sizeof e->bwd->tr.covarJ);
  }
}
int lrr_oracle_oscillated(lrr_oracle_est *e) { return e->oscillation_count > 10; }

int64_t lrr_oracle_num_matches(lrr_oracle_est *e, int dir) {
  fwd_est *f = dir ? e->bwd : e->fwd;
  return f ? f->nmatch : 0;
}
void lrr_oracle_get_matches(lrr_oracle_est *e, int dir, int64_t *from_idx, int64_t *to_idx, double *w) {
  fwd_est *f = dir ? e->bwd : e->fwd;
  if (!f) return;
  for (int64_t i = 0; i < f->nmatch; ++i) {
    if (from_idx) from_idx[i] = f->m_from[i];
    if (to_idx) to_idx[i] = f->m_to[i];
    if (w) w[i] = f->m_w[i];
  }
}
int64_t lrr_oracle_get_number_points(lrr_oracle_est *e, int dir) {
  fwd_est *f = dir ? e->bwd : e->fwd;
  return f ? f->number_points : 0;
}
double lrr_oracle_last_scale(lrr_oracle_est *e, int dir) {
  fwd_est *f = dir ? e->bwd : e->fwd;
  return f ? f->last_scale : NAN;
}
void lrr_oracle_last_normal_equations(lrr_oracle_est *e, double *AtA, double *Atb) {
  if (AtA) memcpy(AtA, e->ne.AtA, sizeof(double) * (size_t)(e->dof * e->dof));
  if (Atb) memcpy(Atb, e->ne.Atb, sizeof(double) * (size_t)e->dof);
}
int lrr_oracle_finest_level(lrr_oracle_est *e) { return e->finest_level; }
