/* lbfgs_standin.h — a small limited-memory BFGS minimiser in plain C (header-only), the stand-in for the optimiser the
 * reference calls through cdcl_lbfgs (cdcl/cdcl_lbfgs.h: a thin wrapper of vnl_lbfgs, i.e. netlib's LBFGS — third party,
 * absent here, "parity unpinned").  It is NOT a restatement of that routine: same family (two-loop recursion over the
 * last m = 5 curvature pairs, initial step 1 / |g|, Armijo backtracking with safeguarded quadratic interpolation), same contract as the caller uses it
 * (cdcl/cdcl_estimation.txx:230-248, 298-311): a cap on function evaluations, the function value at the start and at the
 * end, a return code that is negative only when no descent step was found at all, and (lrr_lbfgs_minimize_stop) the 5 %
 * stopping rule that is the one thing cdcl_lbfgs adds to vnl_lbfgs.
 *
 * Shared by the drop-in classes (include/lrr_amd/cdcl_estimation.h) and by the CPU checker under oracle/ (which includes this
 * header; nothing here depends on it): both
 * sides drive THEIR OWN objective with the same deterministic optimiser, so an EM iteration of the device path can be
 * compared with the oracle's; what is being checked is the objective and its gradient, not the optimiser. */
#ifndef LRR_AMD_LBFGS_STANDIN_H_
#define LRR_AMD_LBFGS_STANDIN_H_

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* value and gradient at x (n entries); g is never NULL */
typedef void (*lrr_lbfgs_fg)(void *user, const double *x, double *f, double *g);

/* USER_STOP: the stopping rule of cdcl_lbfgs::report_iter (vnl_nonlinear_minimizer::FAILED_USER_REQUEST = 9: positive, so the
 * estimator keeps the point reached, cdcl/cdcl_estimation.txx:242-248) */
enum { LRR_LBFGS_CONVERGED = 1, LRR_LBFGS_TOO_MANY_EVALS = 5, LRR_LBFGS_USER_STOP = 9, LRR_LBFGS_NO_DESCENT = -1 };

typedef struct {
  double f_start, f_end; /* get_start_error(), get_end_error() */
  int evals;             /* function evaluations used */
  int code;              /* > 0: x holds the best point found; < 0: failure, x is left as it was */
} lrr_lbfgs_result;

/* minimises f over x[0..n) in place; max_evals >= 1 function evaluations (the reference: 5).
 * stop_fraction > 0: the one behaviour cdcl_lbfgs adds to vnl_lbfgs (cdcl/cdcl_lbfgs.h:28-36, report_iter): after every
 * iteration (accepted iterate) the minimisation ends once |f_start - f| > stop_fraction |f_start| — 0.05 in the reference:
 * "stop when the objective function has been sufficiently decreased".  0 = plain minimiser. */
static inline lrr_lbfgs_result lrr_lbfgs_minimize_stop(int n, double *x, lrr_lbfgs_fg fg, void *user, int max_evals, double stop_fraction) {
  enum { M = 5 };
  lrr_lbfgs_result R;
  const double c1 = 1e-4, gtol = 1e-12, ftol = 1e-14;
  double *g = (double *)malloc(sizeof(double) * (size_t)n * (4 + 2 * M));
  double *d = g + n, *xt = d + n, *gt = xt + n, *S = gt + n, *Y = S + (size_t)M * n;
  double rho[M], alpha[M];
  int hist = 0, head = 0; /* pairs stored, position of the next pair */
  double f = 0.0;
  fg(user, x, &f, g);
  R.f_start = R.f_end = f;
  R.evals = 1;
  R.code = LRR_LBFGS_TOO_MANY_EVALS;
  int moved = 0, first = 1;
  while (R.evals < max_evals) {
    double gnorm2 = 0.0;
    for (int i = 0; i < n; ++i) gnorm2 += g[i] * g[i];
    if (!(gnorm2 > gtol * gtol)) { R.code = LRR_LBFGS_CONVERGED; break; }
    /* direction: two-loop recursion over the stored pairs (newest first) */
    for (int i = 0; i < n; ++i) d[i] = -g[i];
    for (int k = 0; k < hist; ++k) {
      const int j = (head - 1 - k + 2 * M) % M;
      double a = 0.0;
      for (int i = 0; i < n; ++i) a += S[(size_t)j * n + i] * d[i];
      a *= rho[j];
      alpha[j] = a;
      for (int i = 0; i < n; ++i) d[i] -= a * Y[(size_t)j * n + i];
    }
    if (hist > 0) {
      const int j = (head - 1 + M) % M;
      double yy = 0.0, sy = 0.0;
      for (int i = 0; i < n; ++i) { yy += Y[(size_t)j * n + i] * Y[(size_t)j * n + i]; sy += S[(size_t)j * n + i] * Y[(size_t)j * n + i]; }
      const double gamma = sy / yy;
      for (int i = 0; i < n; ++i) d[i] *= gamma;
    }
    for (int k = hist - 1; k >= 0; --k) {
      const int j = (head - 1 - k + 2 * M) % M;
      double b = 0.0;
      for (int i = 0; i < n; ++i) b += Y[(size_t)j * n + i] * d[i];
      b *= rho[j];
      for (int i = 0; i < n; ++i) d[i] += (alpha[j] - b) * S[(size_t)j * n + i];
    }
    double gd = 0.0;
    for (int i = 0; i < n; ++i) gd += g[i] * d[i];
    if (!(gd < 0.0)) { /* not a descent direction: restart from steepest descent */
      hist = 0; head = 0;
      for (int i = 0; i < n; ++i) d[i] = -g[i];
      gd = -gnorm2;
    }
    double step = first ? 1.0 / sqrt(gnorm2) : 1.0;
    first = 0;
    /* Armijo backtracking */
    int accepted = 0;
    double ft = f;
    while (R.evals < max_evals) {
      for (int i = 0; i < n; ++i) xt[i] = x[i] + step * d[i];
      fg(user, xt, &ft, gt);
      ++R.evals;
      if (ft == ft && ft <= f + c1 * step * gd) { accepted = 1; break; }
      /* next trial: the minimiser of the parabola through f, its slope gd and ft, kept inside [0.05, 0.5] step (the
       * safeguarded interpolation every line search of this family uses; a plain halving needs more trials than the five
       * evaluations the caller allows when the first step 1 / |g| overshoots by orders of magnitude) */
      double shrink = 0.5;
      if (ft == ft) {
        const double curv = ft - f - gd * step;
        if (curv > 0.0) { shrink = -gd * step / (2.0 * curv); if (shrink < 0.05) shrink = 0.05; if (shrink > 0.5) shrink = 0.5; }
      } else {
        shrink = 0.05;
      }
      step *= shrink;
    }
    if (!accepted) break;
    /* curvature pair */
    double sy = 0.0;
    for (int i = 0; i < n; ++i) {
      S[(size_t)head * n + i] = xt[i] - x[i];
      Y[(size_t)head * n + i] = gt[i] - g[i];
      sy += S[(size_t)head * n + i] * Y[(size_t)head * n + i];
    }
    if (sy > 0.0) { rho[head] = 1.0 / sy; head = (head + 1) % M; if (hist < M) ++hist; }
    const double df = f - ft;
    memcpy(x, xt, sizeof(double) * (size_t)n);
    memcpy(g, gt, sizeof(double) * (size_t)n);
    f = ft;
    moved = 1;
    if (stop_fraction > 0.0 && fabs(R.f_start - f) > stop_fraction * fabs(R.f_start)) { R.code = LRR_LBFGS_USER_STOP; break; }
    if (df <= ftol * fabs(f)) { R.code = LRR_LBFGS_CONVERGED; break; }
  }
  R.f_end = f;
  if (!moved && R.code == LRR_LBFGS_TOO_MANY_EVALS && max_evals > 1) R.code = LRR_LBFGS_NO_DESCENT;
  free(g);
  return R;
}

static inline lrr_lbfgs_result lrr_lbfgs_minimize(int n, double *x, lrr_lbfgs_fg fg, void *user, int max_evals) {
  return lrr_lbfgs_minimize_stop(n, x, fg, user, max_evals, 0.0);
}
/* the fraction cdcl_lbfgs::report_iter stops at */
#define LRR_CDCL_LBFGS_STOP_FRACTION 0.05

#ifdef __cplusplus
}
#endif
#endif
