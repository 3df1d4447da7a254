/*
 * lrr_oracle.h — flat C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lrr_amd/ (the product) may include,
 * link, load or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker / CPU baseline.
 *
 * PARITY UNPINNED: the reference (msofka/LRR) ships no tests, fixtures or
 * golden vectors for this path and cannot be built here (VXL/ITK absent), so
 * this restatement is pinned only by known-answer tests (tests/test_oracle_*.py).
 */
#ifndef LRR_ORACLE_H_
#define LRR_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A feature set as flat arrays (cdcl_feature_with_shape / cdcl_feature_ICP fields,
 * cdcl/cdcl_feature_with_shape.h:22-89, cdcl/cdcl_feature_ICP.h:22-96). */
typedef struct {
  int64_t        n;
  const double  *loc;       /* n x 3, location_                                  */
  const double  *proj;      /* n x 9 row-major, error_projector_                 */
  const double  *strength;  /* n, strength_ (rounded to float for the rrl kinds) */
  const uint8_t *shape;     /* n, 0 CORNER 1 TUBE 2 SHEET (may be NULL)          */
  const double  *dirs;      /* n x 6, directions_[0], directions_[1] (may be NULL) */
  const double  *cov;       /* n x 9 row-major, covariance_ of cdcl_feature_with_shape (may be NULL)  */
} lrr_oracle_features;

/* estimator kinds */
enum {
  LRR_ORACLE_CDCL_FORWARD   = 0, /* cdcl_estimation_ICP_matching_all<3,dof>            */
  LRR_ORACLE_RRL_FORWARD    = 1, /* rrl_estimation_ICP_matching_all<3,dof>             */
  LRR_ORACLE_CDCL_SYMMETRIC = 2, /* cdcl_estimation_symmetric_ICP_matching_all<3,dof>  */
  LRR_ORACLE_RRL_SYMMETRIC  = 3  /* rrl_estimation_symmetric_ICP_matching_all<3,dof>   */
};

/* nearest-neighbour back-ends: brute force and kd-tree are exact (ties -> lowest index); VORONOI is the
 * production rrl matcher's 1 mm index-volume lookup (rrl/rrl_estimation_ICP_matching_all.txx:530-543,
 * rrl/examples/compute_distance_map.cxx:55-151) */
enum { LRR_ORACLE_NN_BRUTE = 0, LRR_ORACLE_NN_KDTREE = 1, LRR_ORACLE_NN_VORONOI = 2 };

typedef struct lrr_oracle_est lrr_oracle_est;

/* dof = 6 (cdcl_trans_rigid3d) or 12 (cdcl_trans_affine<3,12>).
 * Forward kinds: moving/fixed are the two sets; *_inside must be NULL.
 * Symmetric kinds: moving/fixed are the full volumes, *_inside the ROI copies
 *   (ctor at rrl_estimation_symmetric_ICP_matching_all.txx:198-220).
 * A (3x3 row-major), t, c: initial transform  x -> A (x - c) + t.
 * seed: seed of the deterministic stand-in for std::random_shuffle. */
lrr_oracle_est *lrr_oracle_create(int kind, int dof,
                                  const lrr_oracle_features *moving,
                                  const lrr_oracle_features *fixed,
                                  const lrr_oracle_features *moving_inside,
                                  const lrr_oracle_features *fixed_inside,
                                  const double *A, const double *t, const double *c,
                                  uint64_t seed, int nn_backend);
void lrr_oracle_destroy(lrr_oracle_est *e);

void lrr_oracle_set_transform(lrr_oracle_est *e, const double *A, const double *t, const double *c);
void lrr_oracle_set_features(lrr_oracle_est *e, const lrr_oracle_features *moving_inside,
                             const lrr_oracle_features *fixed_inside);
void lrr_oracle_weight_by_strength(lrr_oracle_est *e, int flag);
void lrr_oracle_normalize_matches(lrr_oracle_est *e);
/* test hooks: not in the reference */
void lrr_oracle_set_number_points(lrr_oracle_est *e, int64_t n);          /* protected in the reference */
void lrr_oracle_set_subset_seed(lrr_oracle_est *e, uint64_t seed);        /* restart the subset-permutation generators */
void lrr_oracle_inject_initial_scale(lrr_oracle_est *e, double fwd, double bwd); /* bypass MUSE; <=0 disables */
void lrr_oracle_set_subset_order(lrr_oracle_est *e, int dir, const int64_t *order, int64_t n); /* explicit permutation for the next match */

void lrr_oracle_find_closest_euclidean(lrr_oracle_est *e);
int  lrr_oracle_one_iteration(lrr_oracle_est *e, unsigned iteration);
int  lrr_oracle_run(lrr_oracle_est *e, unsigned max_iterations, unsigned *iterations_done);
void lrr_oracle_estimate_LS(lrr_oracle_est *e, int estimate_parameters);
void lrr_oracle_estimate_LS_backward(lrr_oracle_est *e, int estimate_parameters);

/* dir: 0 forward, 1 backward */
void   lrr_oracle_get_transform(lrr_oracle_est *e, int dir, double *A, double *t, double *c, double *covar /*dof*dof*/);
double lrr_oracle_RMS_error(lrr_oracle_est *e);
double lrr_oracle_median_error(lrr_oracle_est *e);
double lrr_oracle_weighted_error(lrr_oracle_est *e, int dir);
double lrr_oracle_sheet_angles(lrr_oracle_est *e, int dir);
double lrr_oracle_tube_angles(lrr_oracle_est *e, int dir);
void   lrr_oracle_compute_transfer_error_covariance(lrr_oracle_est *e, double *covarJ_fwd, double *covarJ_bwd);
int    lrr_oracle_oscillated(lrr_oracle_est *e);

/* introspection for parity tests */
int64_t lrr_oracle_num_matches(lrr_oracle_est *e, int dir);
void    lrr_oracle_get_matches(lrr_oracle_est *e, int dir, int64_t *from_idx, int64_t *to_idx, double *w);
int64_t lrr_oracle_get_number_points(lrr_oracle_est *e, int dir);
double  lrr_oracle_last_scale(lrr_oracle_est *e, int dir);
void    lrr_oracle_last_normal_equations(lrr_oracle_est *e, double *AtA /*dof*dof*/, double *Atb /*dof*/);
int     lrr_oracle_finest_level(lrr_oracle_est *e);

/* SURVEY 8(f)4, first step: cdcl_estimation::compute_weights at one resolution (cdcl/cdcl_estimation.txx:342-545):
 * radius search under the transfer-error covariance, <= max_matches candidates per moving feature, Mahalanobis
 * weights, two-sided normalisation.  Both feature sets need `cov`.  Rows of max_matches entries, ascending fixed index. */
int64_t lrr_oracle_cdc_compute_weights(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, int dof,
                                       const double *A, const double *t, const double *c, const double *param_covar,
                                       const uint32_t keys[4], int max_matches, int64_t *row_count, int64_t *to_idx, double *w);
/* cdcl_obj_fun_par (CDC-EM M-step, parameters): objective and gradient over the match rows (cdcl/cdcl_obj_fun_par.txx:20-142) */
int64_t lrr_oracle_cdc_obj_par(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, int dof, const double *A,
                               const double *t, const double *c, const double *param_covar, int max_matches,
                               const int64_t *row_count, const int64_t *to_idx, const double *w, const double *norm,
                               double *f_out, double *g_out);
/* cdcl_obj_fun_cov (CDC-EM M-step, covariance): objective and gradient in the Cholesky parameters (cdcl/cdcl_obj_fun_cov.txx:17-134) */
int64_t lrr_oracle_cdc_obj_cov(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, int dof, const double *A,
                               const double *t, const double *c, const double *x_upper, int max_matches,
                               const int64_t *row_count, const int64_t *to_idx, const double *w, const double *norm,
                               double *f_out, double *g_out);
/* cdcl_normalize_matches over the one-to-many rows -> {cm[3], rm, cf[3], rf} (cdcl/cdcl_estimation_ICP.txx:751-795) */
void lrr_oracle_cdc_normalize_stats(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, int max_matches,
                                    const int64_t *row_count, const int64_t *to_idx, double *norm);
/* one EM iteration of cdcl_estimation, affine transform, no resolution switching (cdcl/cdcl_estimation.txx:95-187, 210-338) */
/* the same for dof 6 (rigid, small-angle increment) or 12 over a pyramid of resolutions ([0] the finest; *res_inout: the current one,
 * coarsest at the start); spacing[r]: point spacing of fixed[r] (NULL with one level); rows of the resolution left in *res_inout */
int lrr_oracle_cdc_em_iteration_pyramid(int dof, int levels, const lrr_oracle_features *const *moving, const lrr_oracle_features *const *fixed,
                                        const double *spacing, double *A, double *t, double *c, double *covar, unsigned iteration,
                                        uint64_t *seed_state, int max_matches, int *res_inout, int64_t *row_count, int64_t *to_idx, double *w,
                                        double *info, double *state);
int lrr_oracle_cdc_em_iteration(const lrr_oracle_features *moving, const lrr_oracle_features *fixed, double *A, double *t, double *c,
                                double *covar, unsigned iteration, uint64_t *seed_state, int max_matches,
                                int64_t *row_count, int64_t *to_idx, double *w, double *info);

/* cdcl_estimation::initialize at one resolution (cdcl/cdcl_estimation.txx:70-84 -> cdcl_estimation_ICP.txx:451-489, 497-750): closest
 * fixed feature of every moving feature, residuals along the normal of the fixed feature's covariance, 2 x MUSE scale, the parameter
 * covariance (dof x dof) = inverse of AtA; A, t, c come back as normalize() / unnormalize() around the fit leave them (the same map
 * about the moving centroid) */
int lrr_oracle_cdc_em_initialize(int dof, const lrr_oracle_features *moving, const lrr_oracle_features *fixed, double *A,
                                 double *t, double *c, int normalize, double *covar_out, double *scale_out);

/* stand-alone pieces (each is also used inside the estimators) */
void   lrr_oracle_nn(const double *fixed_loc, int64_t nf, const double *query, int64_t nq,
                     int backend, int64_t *idx_out);
double lrr_oracle_muse_scale(const double *abs_res, int64_t n);
double lrr_oracle_mad_scale(double *v, int64_t n);            /* permutes v */
double lrr_oracle_weight_BT(double u);
void   lrr_oracle_shuffle(int64_t *order, int64_t n, uint64_t *state);
void   lrr_oracle_perm_keys(uint64_t *state, uint32_t keys[4]);
int64_t lrr_oracle_subset_perm(int64_t i, int64_t n, const uint32_t keys[4]);
void   lrr_oracle_sym_pinv(int n, const double *M, double *out);
void   lrr_oracle_svd3(const double *M, double *U, double *s, double *V);

#ifdef __cplusplus
}
#endif
#endif
