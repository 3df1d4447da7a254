/*
 * lrr_icp.h — C-ABI of the MI355X (gfx950) robust-ICP hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI: its seam is the C++
 * class family cdcl_estimation_ICP_matching_all / rrl_estimation_symmetric_ICP_matching_all.
 * The host-side classes in include/lrr_amd/ keep those names and methods and call ONLY the
 * functions below; every function names the reference loop it replaces (paths relative to the
 * msofka/LRR repo root).  Plain pointers and sizes, no C++ / torch types, no exceptions: every
 * function returns 0 on success or a negative code and records a message readable through
 * lrr_icp_last_error().  One context per host thread; all work of a context is issued on its
 * own HIP stream.
 *
 * Coordinates: "xf" is x -> A (x - c) + t (cdcl_trans_rigid3d.cxx:171-174,
 * cdcl_trans_affine.txx:112-115).  "norm" carries the normalisation of
 * cdcl_normalize_matches (cdcl/cdcl_estimation_ICP_matching_all.txx:1159-1240): the reference
 * rewrites the shared feature objects in place and restores them afterwards; the device clouds
 * are never mutated, kernels apply (x - centre) / radius on the fly.
 */
#ifndef LRR_ICP_H_
#define LRR_ICP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRR_ICP_OK 0
#define LRR_ICP_ERR_ARG (-1)
#define LRR_ICP_ERR_HIP (-2)
#define LRR_ICP_ERR_STATE (-3)

typedef struct lrr_icp_ctx lrr_icp_ctx;
typedef struct lrr_icp_cloud lrr_icp_cloud; /* a feature vector resident in HBM + its search grid */
typedef struct lrr_icp_dir lrr_icp_dir;     /* device state of one (forward) estimation object: matches_, w_ */

typedef struct { double A[9]; double t[3]; double c[3]; } lrr_icp_xform;
/* from-side points are normalised by (cm, rm), to-side points by (cf, rf); identity = {0,1,0,1} */
typedef struct { double cm[3]; double rm; double cf[3]; double rf; } lrr_icp_norm;

/* nearest-neighbour methods.  BRUTE and GRID are exact (ties -> lowest index).  VORONOI reproduces the lookup
 * of the production rrl matcher (rrl/rrl_estimation_ICP_matching_all.txx:530-543): the mapped point is rounded
 * to its voxel of a 1 mm volume over the to-cloud's bounding box, clamped to the volume, and matched to the
 * feature the nearest occupied voxel stands for (rrl/examples/compute_distance_map.cxx:55-151; exact distances
 * between voxel centres where the reference uses Danielsson's approximation; ties -> lowest feature index) */
enum { LRR_ICP_NN_AUTO = 0, LRR_ICP_NN_BRUTE = 1, LRR_ICP_NN_GRID = 2, LRR_ICP_NN_VORONOI = 3 };

/* accumulate flags */
enum {
  LRR_ICP_ACC_RIGID_INCREMENT = 1, /* b = q - p   (cdcl .txx:1051-1054)                            */
  LRR_ICP_ACC_REVERSED = 2,        /* role-reversed term: J at to_, projector of from_, b = from_
                                      (rrl sym .txx:546-655; cdcl sym add_constraints .txx:367-393)  */
  LRR_ICP_ACC_PREMUL_SKIP = 4,     /* rrl: projector pre-multiplied by w, skip w < 1e-4
                                      (rrl sym .txx:463-468)                                         */
  LRR_ICP_ACC_CLONE_W = 8          /* use the clone weights written by the *_merged weighting calls   */
};

/* ---- context ---------------------------------------------------------------------------- */
int  lrr_icp_create(int device, lrr_icp_ctx **out);
void lrr_icp_destroy(lrr_icp_ctx *ctx);
const char *lrr_icp_last_error(const lrr_icp_ctx *ctx);
int  lrr_icp_synchronize(lrr_icp_ctx *ctx);
/* HIP-event stopwatch on the context's stream (bench.py's kernel timing) */
int  lrr_icp_timer_start(lrr_icp_ctx *ctx);
int  lrr_icp_timer_stop(lrr_icp_ctx *ctx, double *elapsed_ms);
/* accumulated device time of the matching kernels since the last reset (HIP events around
 * every K1 launch when profiling is enabled) */
int  lrr_icp_match_profile(lrr_icp_ctx *ctx, int enable, double *total_ms, int64_t *launches, int reset);
/* the elapsed time of every timed K1 submission since the last reset, in submission order (a submission serves one
 * direction, or both directions of a symmetric estimator in the same launches): *count = how many there are, the first
 * min(cap, *count) are copied to ms.  bench.py's median / spread of the K1 kernel time. */
int  lrr_icp_match_profile_samples(lrr_icp_ctx *ctx, double *ms, int64_t cap, int64_t *count);
/* how many searches (directions) each state of the grid matcher has served on this context since the last reset:
 * certificate kernel that searches its own failures / certificate + list + wave-per-query kernels / ball search of a
 * settled level / ball search right after a level grew / tile kernel (first dense search) / coarse pair path
 * (wave-per-query) / thread-per-slot ring search.  Diagnostics and tests; no reference counterpart. */
enum { LRR_ICP_K1_CERT_INLINE = 0, LRR_ICP_K1_CERT = 1, LRR_ICP_K1_BALL = 2, LRR_ICP_K1_BALL_GROWTH = 3, LRR_ICP_K1_TILE = 4,
       LRR_ICP_K1_COARSE_PAIR = 5, LRR_ICP_K1_DIRECT = 6, LRR_ICP_K1_PATHS = 7 };
int  lrr_icp_k1_path_counts(lrr_icp_ctx *ctx, int64_t counts[LRR_ICP_K1_PATHS], int reset);
/* elapsed time of an EMPTY event pair on the context's stream, mean of `pairs` (the markers' own cost inside a timed submission) */
int  lrr_icp_event_overhead(lrr_icp_ctx *ctx, int pairs, double *ms_per_pair);

/* K1 at the finest level skips the search of a source whose previous match is provably still its unique
 * nearest neighbour (second-nearest bound + triangle inequality, k_match_cert in lrr_icp.hip): sources seen by
 * that kernel and sources it had to search since the last reset */
int  lrr_icp_cert_stats(lrr_icp_ctx *ctx, int64_t *queries, int64_t *searched, int reset);

/* ---- feature clouds ---------------------------------------------------------------------
 * Replaces the vcl_vector<feature_sptr> the estimators hold plus build_kd_trees()
 * (cdcl/cdcl_estimation_ICP_matching_all.txx:113-135) / set_voronoi_maps()
 * (rrl/rrl_estimation_symmetric_ICP_matching_all.txx:293-299): uploads location_,
 * error_projector_ (must be symmetric; 9 doubles row-major) and strength_ once and builds the
 * uniform search grid.  float_strength != 0 rounds strengths to float and makes the strength
 * product a float multiply (cdcl_feature_ICP.h:83). */
int  lrr_icp_cloud_create(lrr_icp_ctx *ctx, int64_t n, const double *loc, const double *proj,
                          const double *strength, int float_strength, lrr_icp_cloud **out);
void lrr_icp_cloud_destroy(lrr_icp_cloud *cloud);
int64_t lrr_icp_cloud_size(const lrr_icp_cloud *cloud);
/* the occupied voxels of the 1 mm index volume compute_distance_map builds for this feature set (what
 * set_voronoi_maps hands to the rrl classes); called on demand by the first LRR_ICP_NN_VORONOI search */
int  lrr_icp_cloud_build_voronoi(lrr_icp_cloud *cloud);

/* ---- one estimation direction ------------------------------------------------------------ */
int  lrr_icp_dir_create(lrr_icp_ctx *ctx, lrr_icp_cloud *from, lrr_icp_cloud *to, lrr_icp_dir **out);
void lrr_icp_dir_destroy(lrr_icp_dir *dir);
/* set_moving_data (cdcl/cdcl_estimation_ICP_matching_all.h:96): new from-cloud, matches cleared */
int  lrr_icp_dir_set_from(lrr_icp_dir *dir, lrr_icp_cloud *from);
int64_t lrr_icp_dir_num_matches(const lrr_icp_dir *dir);

/* keys of the seeded subset permutation (stand-in for std::random_shuffle, cdcl .txx:557):
 * advances *state and fills 4 round keys; perm(i) is evaluated inside the match kernel. */
void lrr_icp_perm_keys(uint64_t *state, uint32_t keys[4]);
int64_t lrr_icp_subset_perm(int64_t i, int64_t n, const uint32_t keys[4]);

/* K1 — find_closest_euclidean (cdcl .txx:541-654; rrl .txx:494-571): for i < n_points:
 * p = from[perm(i)], y = xf(p), q = argmin_j |y - to_j| (exact, ties -> lowest j), w = 1.
 * order: optional explicit host permutation (n_points indices into the from-cloud); when NULL the
 * keyed permutation is used. */
int  lrr_icp_match(lrr_icp_dir *dir, const lrr_icp_xform *xf, const uint32_t keys[4],
                   const int64_t *order, int64_t n_points, int method);
/* The symmetric classes search both lists back to back (forward_->find_closest_euclidean();
 * backward_->find_closest_euclidean(): rrl/rrl_estimation_symmetric_ICP_matching_all.txx:341-356,
 * cdcl/cdcl_estimation_symmetric_ICP_matching_all.txx:268-283).  Same results as two lrr_icp_match
 * calls; the two searches are submitted together and run concurrently on the device. */
int  lrr_icp_match_pair(lrr_icp_dir *fwd, const lrr_icp_xform *xf_fwd, const uint32_t keys_fwd[4],
                        const int64_t *order_fwd, int64_t n_fwd,
                        lrr_icp_dir *bwd, const lrr_icp_xform *xf_bwd, const uint32_t keys_bwd[4],
                        const int64_t *order_bwd, int64_t n_bwd, int method);

/* rigid pre-map (cdcl .txx:455-484; cdcl sym .txx:400-456): every match's from_ becomes a new
 * feature located at G(p) */
int  lrr_icp_premap(lrr_icp_dir *dir, const lrr_icp_xform *G);

/* K2 pass 1 — estimate_scale_and_assign_weight, first loop (cdcl .txx:677-708; rrl .txx:594-617):
 * r2_i = max(0, e^T P_q e), e = xf(p~) - q~; stashes r2; returns S = sum w_prev r2, W = sum w_prev */
int  lrr_icp_residual_pass(lrr_icp_dir *dir, const lrr_icp_xform *xf, const lrr_icp_norm *norm,
                           double *S, double *W);
/* rrel_muset_obj(0,false).scale on sqrt(r2) (call sites cdcl .txx:721-725; rrl .txx:630-634) */
int  lrr_icp_muse_scale(lrr_icp_dir *dir, double *sigma);
/* K2 pass 2 (cdcl .txx:744-765; rrl .txx:652-671): w_i = BT(r2_i / sigma^2) / sigma^2 */
int  lrr_icp_assign_weights(lrr_icp_dir *dir, double sigma);

/* K5 — cdcl_normalize_matches statistics (cdcl .txx:1159-1213; rrl .txx:864-916): centroid and
 * mean radius of from_ over matches and of the DISTINCT matched to_ */
int  lrr_icp_normalize_stats(lrr_icp_dir *dir, lrr_icp_norm *out);

/* the same statistics over a merged list: own's matches plus the role-reversed clones of rev's matches (clones are new
 * objects: each counts among the distinct to_).  cdcl sym estimate_LS_backward adds the constraints BEFORE it normalises
 * (cdcl/cdcl_estimation_symmetric_ICP_matching_all.txx:742-755) */
int  lrr_icp_normalize_stats_merged(lrr_icp_dir *own, lrr_icp_dir *rev, lrr_icp_norm *out);

/* add_constraints (cdcl sym .txx:367-393): the role-reversed clones of this direction's matches
 * start with a copy of its weights; the merged weighting calls below then work on the copy. */
int  lrr_icp_clone_weights(lrr_icp_dir *dir);
/* weight_by_strength() (cdcl .txx:826-862; rrl .txx:722-752).  b may be NULL.  With b != NULL the
 * two lists are weighted as ONE merged list (cdcl sym .txx:541-544): a's own weights and b's
 * clone weights. */
int  lrr_icp_weight_by_strength(lrr_icp_dir *a, lrr_icp_dir *b_merged, double *grad_scale);
/* weight_spatially() (cdcl .txx:973-1004; rrl .txx:756-778); to-locations normalised by norm.cf/rf.
 * Merged form as above: b's clones are role-reversed, their to_ is b's from_ (normalised by
 * norm_b.cm/rm). */
int  lrr_icp_weight_spatially(lrr_icp_dir *a, const lrr_icp_norm *norm_a, lrr_icp_dir *b_merged,
                              const lrr_icp_norm *norm_b, double *spatial_scale);

/* K3 — normal equations (cdcl .txx:1039-1069; rrl sym .txx:441-655): AtA (dof x dof row-major,
 * both triangles filled) and Atb of this list; jac_centre is trans_->center_moving_ of the
 * transform whose Jacobian is used. */
int  lrr_icp_accumulate(lrr_icp_dir *dir, int dof, const double jac_centre[3],
                        const lrr_icp_norm *norm, int flags, double *AtA, double *Atb);

/* K4 — RMS_error / weighted_error in one pass (cdcl .txx:196-268; rrl .txx:211-276):
 * sum_r2 = sum r2_i, sum_wr = sum sqrt(r2_i) w_i, sum_w = sum w_i */
int  lrr_icp_errors(lrr_icp_dir *dir, const lrr_icp_xform *xf, double *sum_r2, double *sum_wr,
                    double *sum_w);
/* the same in two halves: launch queues the kernel and returns, collect waits for its sums; other work may be queued on the
 * context in between (one launch outstanding per context: a second launch first completes the first) */
int  lrr_icp_errors_launch(lrr_icp_dir *dir, const lrr_icp_xform *xf);
int  lrr_icp_errors_collect(lrr_icp_dir *dir, double *sum_r2, double *sum_wr, double *sum_w);

/* ---- post-convergence diagnostics (the verifier's inputs, rrl/itkLocationRegistration.cxx:1816-1871) --
 * shapes and directions of a cloud's features: shape[i] in {0 CORNER, 1 TUBE, 2 SHEET}
 * (cdcl/cdcl_feature_with_shape.h:28), ndirs[i] = directions_.size(), dirs6 = directions_[0], directions_[1] */
int  lrr_icp_cloud_set_shapes(lrr_icp_cloud *cloud, const uint8_t *shape, const uint8_t *ndirs, const double *dirs6);
/* the cdcl classes compare eigenvectors of the features' covariances instead (cdcl .txx:273-382): per feature the
 * eigenvector of the smallest eigenvalue (sheets) and of the largest (tubes), vnl_symmetric_eigensystem_compute order */
int  lrr_icp_cloud_set_shape_vectors(lrr_icp_cloud *cloud, const uint8_t *shape, const double *sheet3, const double *tube3);
/* ... compared as they are; R (3x3 row-major) rotates the from-feature's vector: the rotation of the rigid pre-map
 * (its covariance became R C R^T, cdcl .txx:473-479), identity otherwise */
int  lrr_icp_angles_unmapped(lrr_icp_dir *dir, const double R[9], int tube, double *sum_wa, double *sum_w);
/* sheet_angles (tube = 0) / tube_angles (tube = 1) (rrl/rrl_estimation_ICP_matching_all.txx:281-375;
 * cdcl .txx:273-382): sum of w_i * angle_i and sum of w_i over the sheet-sheet / tube-tube matches */
int  lrr_icp_angles(lrr_icp_dir *dir, const lrr_icp_xform *xf, int tube, double *sum_wa, double *sum_w);
/* compute_transfer_error_covariance (cdcl .txx:388-404): sum_i J(p_i) C J(p_i)^T, 3x3 row-major, C = the
 * dof x dof parameter covariance, J the Jacobian with respect to the parameters at the from-location */
int  lrr_icp_transfer_covariance(lrr_icp_dir *dir, int dof, const double jac_centre[3], const double *covar,
                                 double out9[9]);

/* median_error (cdcl .txx:159-190; rrl .txx:177-203): element (n-1)/2 + 1 of the sorted sqrt(r2_i)
 * (vcl_nth_element at begin + (n-1)/2 + 1); 0 for fewer than two matches, where the reference reads
 * past the end of its list */
int  lrr_icp_median_error(lrr_icp_dir *dir, const lrr_icp_xform *xf, double *median);

/* ---- fused symmetric estimate ----------------------------------------------------------------
 * The device part of {cdcl,rrl}_estimation_symmetric_ICP_matching_all::estimate_LS
 * (rrl sym .txx:699-878; cdcl sym .txx:458-628) as ONE asynchronous submission with a single
 * device->host copy at the end: residuals and scales of both directions, normalisation
 * statistics of the forward list, strength / spatial re-weighting with their MAD scales
 * (order statistics on the device), and the summed normal equations of the forward list and
 * the role-reversed backward list.  Same arithmetic as the granular calls above; scalars that
 * the granular path hands back to the host between kernels stay in HBM here. */
enum { LRR_ICP_SCALE_RMS = 0, LRR_ICP_SCALE_MUSE = 1, LRR_ICP_SCALE_INJECTED = 2 };
typedef struct {
  int dof;              /* 6 or 12 */
  int rrl;              /* 1: rrl flavour (per-list weighting, pre-multiplied projector, skip w<1e-4);
                           0: cdcl flavour (merged list weighting, clone weights for the backward list) */
  int normalize;        /* normalize_matches_ */
  int strength_f, strength_b; /* weight_by_strength_ of the two directions */
  int rigid_increment;  /* b = q - p */
  int scale_mode_f, scale_mode_b;
  double injected_f, injected_b;
  double jac_centre[3]; /* center_moving_ of the forward transform as the accumulation sees it */
} lrr_icp_sym_opts;
typedef struct {
  lrr_icp_norm norm;    /* (cm, rm, cf, rf) of the forward list; identity when !normalize */
  double sigma_f, sigma_b;
  double grad_scale[2], spatial_scale[2]; /* [0] forward (or merged), [1] backward */
  double AtA[144], Atb[12];
} lrr_icp_sym_out;
int  lrr_icp_estimate_symmetric(lrr_icp_dir *fwd, lrr_icp_dir *bwd, const lrr_icp_xform *xf_fwd,
                                const lrr_icp_xform *xf_bwd, const lrr_icp_sym_opts *opts,
                                lrr_icp_sym_out *out);

/* estimate_LS of the forward classes in one submission (cdcl/cdcl_estimation_ICP_matching_all.txx:1009-1093,
 * rrl/rrl_estimation_ICP_matching_all.txx:783-859): cdcl_normalize_matches statistics (.txx:1159-1240) and
 * trans_->normalize on the device, estimate_scale_and_assign_weight (.txx:660-822), weight_by_strength
 * (.txx:826-862), AtA / Atb (.txx:1039-1069).  xf is the transform BEFORE normalisation; the host repeats
 * normalize / unnormalize on its own transform object around the solve with the returned statistics.
 * Same results as the granular calls above, one host round trip instead of eight. */
typedef struct {
  int dof;               /* 6 or 12 */
  int normalize;         /* normalize_matches_ */
  int strength;          /* weight_by_strength_ */
  int rigid_transform;   /* cdcl_trans_rigid3d: normalize() keeps center_moving_ (cdcl_trans_rigid3d.cxx:260-280) */
  int rigid_increment;   /* right-hand side q - p (cdcl .txx:1051-1057) */
  int scale_mode;        /* LRR_ICP_SCALE_* */
  double injected;       /* scale for LRR_ICP_SCALE_INJECTED */
  double jac_centre[3];  /* center_moving_ of the transform as the accumulation sees it */
} lrr_icp_fwd_opts;
int  lrr_icp_estimate_forward(lrr_icp_dir *dir, const lrr_icp_xform *xf, const lrr_icp_fwd_opts *opts,
                              lrr_icp_sym_out *out);

/* matches_ on demand (callbacks, sheet/tube angles): from/to indices into the clouds, weights */
int  lrr_icp_get_matches(lrr_icp_dir *dir, int64_t *from_idx, int64_t *to_idx, double *w);

/* ---- batched query locations (config 5) -------------------------------------------------------------
 * The caller's outer loop over query locations (rrl/itkLocationRegistration.cxx:2134-2138) runs one
 * rrl_estimation_symmetric_ICP_matching_all<3,12> per location on the features of two 50 mm boxes.  A
 * batch keeps the two full volumes resident and runs EVERY phase of an iteration for B locations in one
 * submission; a location is two index lists into the volumes, selected on the device.  Per-location
 * results equal those of the one-at-a-time calls above (dof 12, rrl flavour: per-list weighting,
 * pre-multiplied projector, skip w < 1e-4). */
typedef struct lrr_icp_batch lrr_icp_batch;
int  lrr_icp_batch_create(lrr_icp_ctx *ctx, lrr_icp_cloud *moving_all, lrr_icp_cloud *fixed_all, lrr_icp_batch **out);
void lrr_icp_batch_destroy(lrr_icp_batch *batch);
int64_t lrr_icp_batch_size(const lrr_icp_batch *batch);
/* the matcher of every location: LRR_ICP_NN_AUTO / _GRID = exact nearest neighbour (default); LRR_ICP_NN_VORONOI = the lookup
 * semantics of the 1 mm Voronoi maps the caller hands to every estimator (set_voronoi_maps, rrl sym .txx:293-299; the lookup
 * rrl/rrl_estimation_ICP_matching_all.txx:530-543).  Call before lrr_icp_batch_set_rois. */
int  lrr_icp_batch_set_nn_method(lrr_icp_batch *batch, int method);
/* FeaturesInROIs (rrl/itkLocationRegistration.cxx:1324-1360) for B locations at once, on the device:
 * box b = {x0[3], x1[3]}; a feature is inside when x0[d] <= p[d] <= x1[d] for every d; the ROI list keeps
 * the volume's original order.  Also set_features: matches and schedule state of every location are
 * reset.  n_*_inside[b]: list sizes (the caller's >= 1000 test, :1632). */
int  lrr_icp_batch_set_rois(lrr_icp_batch *batch, int64_t B, const double *moving_box, const double *fixed_box,
                            int64_t *n_moving_inside, int64_t *n_fixed_inside);
int  lrr_icp_batch_get_roi(lrr_icp_batch *batch, int64_t b, int dir, int64_t *idx);
typedef struct {
  lrr_icp_xform xf_fwd, xf_bwd;      /* forward (moving -> fixed) and backward transforms of the location */
  uint32_t keys_fwd[4], keys_bwd[4]; /* subset permutations (lrr_icp_perm_keys) */
  int64_t n_fwd, n_bwd;              /* number_points_ of the two lists */
  int active;                        /* 0: the location is skipped by the call */
  int scale_mode_f, scale_mode_b;    /* LRR_ICP_SCALE_* of the forward-role and backward-role list */
  double injected_f, injected_b;
  double jac_centre[3];              /* as lrr_icp_sym_opts.jac_centre */
} lrr_icp_batch_loc;
typedef struct {
  int normalize, strength_f, strength_b;
  int swapped;                       /* estimate_LS_backward: the backward list plays the forward role
                                        (rrl/rrl_estimation_symmetric_ICP_matching_all.txx:921-934) */
} lrr_icp_batch_opts;
/* find_closest_euclidean of both lists of every active location (rrl sym .txx:327-331) */
int  lrr_icp_batch_match(lrr_icp_batch *batch, const lrr_icp_batch_loc *loc);
/* the device part of estimate_LS of every active location (as lrr_icp_estimate_symmetric); out[b] */
int  lrr_icp_batch_estimate(lrr_icp_batch *batch, const lrr_icp_batch_loc *loc, const lrr_icp_batch_opts *opts,
                            lrr_icp_sym_out *out);
/* K4 of both lists of every active location: sums[b][dir] = {sum r2, sum sqrt(r2) w, sum w} */
int  lrr_icp_batch_errors(lrr_icp_batch *batch, const lrr_icp_batch_loc *loc, double *sums);
/* sheet / tube angles of both lists: sums[b][dir] = {sheet sum w a, sheet sum w, tube sum w a, tube sum w} */
int  lrr_icp_batch_angles(lrr_icp_batch *batch, const lrr_icp_batch_loc *loc, double *sums);
/* The caller's loop for every location of the batch in one call (rrl/itkLocationRegistration.cxx:2418-2482: one_iteration
 * until converged, max_iterations or an invalid transform), dof 12 / rrl flavour / affine transforms.  The first iteration
 * runs through the calls above (MUSE); from its solve on, the host half of estimate_LS (solve, un-normalise, inverse:
 * rrl sym .txx:880-912) and the convergence test with its schedule (rrl sym .txx:383-437) run as kernels over the
 * per-location tables, so no result travels to the host between iterations.  fin[b] holds the schedule's outcome and the
 * normal equations and normalisation of the last estimate: the caller rebuilds the final transform and its covariance
 * from them exactly as after lrr_icp_batch_estimate. */
typedef struct {
  int32_t run;                       /* 0: the location is left alone (too few features) */
  lrr_icp_xform xf_fwd, xf_bwd;      /* transforms at the start */
  uint64_t rng_fwd, rng_bwd;         /* states of the subset-permutation generators (lrr_icp_perm_keys) */
} lrr_icp_batch_start;
typedef struct {
  int32_t converged, oscillated, valid_transform, iterations;
  int64_t matched_pairs;
  int64_t n_fwd, n_bwd;              /* number_points_ of the two lists after the last schedule step ... */
  int64_t searched_fwd, searched_bwd;/* ... and at the last search */
  double initial_rms, rms, weighted_error;
  double AtA[144], Atb[12];          /* normal equations of the last estimate (iterations > 0) */
  lrr_icp_norm norm;                 /* and its normalisation */
  uint64_t rng_fwd, rng_bwd;         /* generator states after the last search */
} lrr_icp_batch_final;
int  lrr_icp_batch_iterate(lrr_icp_batch *batch, const lrr_icp_batch_start *start, unsigned max_iterations, int check_valid,
                           lrr_icp_batch_final *fin);
int64_t lrr_icp_batch_num_matches(const lrr_icp_batch *batch, int64_t b, int dir);
/* matches_ of location b: from = index in the ROI list, to = index in the full volume */
int  lrr_icp_batch_get_matches(lrr_icp_batch *batch, int64_t b, int dir, int64_t *from_idx, int64_t *to_idx, double *w);

/* ---- covariance-driven EM, E-step (first step) ---------------------------------------
 * covariance_ of every feature of a cloud (n x 9 row-major, original order): cdcl/cdcl_feature.h:21-64 */
int  lrr_icp_cloud_set_covariances(lrr_icp_cloud *cloud, const double *cov9);
/* cdcl_estimation::compute_weights at one resolution (cdcl/cdcl_estimation.txx:342-545): for every moving feature the
 * fixed features within 5 sqrt(lambda_max(J C_theta J^T + J_p C_p J_p^T)) of its mapped location, at most max_matches
 * of them (the reference keeps the first 100 of a shuffled list: here the candidates with the smallest rank under the
 * keyed permutation `keys` of the fixed set), Mahalanobis weights f (1 - r/16)^2 and the two-sided normalisation
 * w_ij^2 / (sum_j w_ij sum_i w_ij).  Rows of max_matches slots per moving feature (original order): row_count[i]
 * matches, to_idx = original fixed indices (-1 in unused slots), in no particular order inside a row.
 * overflow_rows: moving features with more candidates inside the radius than one wavefront holds (1536): their
 * selection is taken among the first 1536 found.  NULL: such a row is an error (LRR_ICP_ERR_STATE) — the reference
 * has no cap, a caller that does not look at the count must not get a silently different match set. */
int  lrr_icp_cdc_compute_weights(lrr_icp_cloud *moving, lrr_icp_cloud *fixed, const lrr_icp_xform *xf, int dof,
                                 const double *param_covar, const uint32_t keys[4], int max_matches,
                                 int64_t *row_count, int64_t *to_idx, double *w, int64_t *overflow_rows);

/* The same with the rows RESIDENT on the device (what the M-step reads): create for a cloud pair, compute_weights fills
 * the rows, get copies them out. */
typedef struct lrr_icp_cdc_set lrr_icp_cdc_set;
int  lrr_icp_cdc_set_create(lrr_icp_cloud *moving, lrr_icp_cloud *fixed, int max_matches, lrr_icp_cdc_set **out);
void lrr_icp_cdc_set_destroy(lrr_icp_cdc_set *set);
int  lrr_icp_cdc_set_compute_weights(lrr_icp_cdc_set *set, const lrr_icp_xform *xf, int dof, const double *param_covar,
                                     const uint32_t keys[4], int64_t *overflow_rows);
int  lrr_icp_cdc_set_get(lrr_icp_cdc_set *set, int64_t *row_count, int64_t *to_idx, double *w);
/* what the resolution switch inside compute_weights reads (cdcl/cdcl_estimation.txx:384, 491-517) after an E-step: the sum over
 * ALL moving features of sqrt(lambda_max(J_th C_theta J_th^T)), and matches_.size() = the moving features whose row is not empty */
int  lrr_icp_cdc_set_stats(lrr_icp_cdc_set *set, double *eval_sum, int64_t *rows_with_matches);
/* from_weights of the last E-step: sum_j w_ij before the normalisation, per moving feature (original order); row_count may be NULL */
int  lrr_icp_cdc_set_row_sums(lrr_icp_cdc_set *set, double *row_sum, int64_t *row_count);
/* w_ij^2 / (wi wj) again with the caller's from-side sums wi: after a resolution switch the reference's restart appends to
 * from_weights instead of clearing it (:356-358), so match i of the second pass is normalised by entry i of the first pass */
int  lrr_icp_cdc_set_renormalize(lrr_icp_cdc_set *set, const double *row_sum);
/* the rigid increment of the EM loop (cdcl/cdcl_estimation.txx:118-141, 157-166): until the next compute_weights every
 * matches_[i]->from_ is the copy of the moving feature at G(p) with covariance R C_p R^T — the objectives and the normalisation
 * statistics below see those copies.  G = NULL: the features themselves. */
int  lrr_icp_cdc_set_premap(lrr_icp_cdc_set *set, const lrr_icp_xform *G);
/* `if( matches_.size() == 0 ) find_closest_euclidean()` (cdcl/cdcl_estimation.txx:543-544; cdcl/cdcl_estimation_ICP.txx:451-489):
 * every moving feature gets its Euclidean-nearest fixed feature under xf, weight 1 */
int  lrr_icp_cdc_set_closest_euclidean(lrr_icp_cdc_set *set, const lrr_icp_xform *xf);
/* M-step, parameters: cdcl_obj_fun_par::compute (cdcl/cdcl_obj_fun_par.txx:44-142; constructor :20-41) at the
 * transform x over the resident rows: f = (1/R) sum w (rho(r) + ln det C_ij) and, when g != NULL, its gradient
 * (dof values: 2 J_th^T C_ij^-1 e and the two correction terms of d C_ij / d theta), R = rows that are not empty.
 * param_covar: the dof x dof parameter covariance the transfer-error term J_th C_theta J_th^T is built from (the
 * objective's constructor takes it from the transform).  norm (may be NULL): the features as cdcl_normalize_matches
 * leaves them (cdcl/cdcl_estimation_ICP.txx:751-833): locations (x - c) / r, covariances C / (r r); x is taken as
 * given (the caller normalises the transform, cdcl/cdcl_estimation.txx:218-221).  The L-BFGS driver
 * (cdcl_lbfgs / vnl_lbfgs, five evaluations per call, cdcl/cdcl_estimation.txx:230-239) stays with the caller. */
int  lrr_icp_cdc_objective_par(lrr_icp_cdc_set *set, const lrr_icp_xform *x, int dof, const double *param_covar,
                               const lrr_icp_norm *norm, double *f, double *g, int64_t *rows);

/* cdcl_normalize_matches for the one-to-many rows (cdcl/cdcl_estimation_ICP.txx:751-795): centroid of the moving features
 * whose row is not empty, centroid of the DISTINCT matched fixed features, mean radii about them */
int  lrr_icp_cdc_normalize_stats(lrr_icp_cdc_set *set, lrr_icp_norm *out);
/* M-step, covariance: cdcl_obj_fun_cov::compute (cdcl/cdcl_obj_fun_cov.txx:42-134; constructor :17-38) over the resident
 * rows.  x_upper: the dof (dof + 1) / 2 entries of the upper-triangular Cholesky factor U of the parameter covariance,
 * row by row (the optimiser's variables: cdcl/cdcl_estimation.txx:283-293); C_theta = U^T U.  f = (1/N) sum w (rho(r) +
 * ln det C_ij) over all N pairs, g (may be NULL): its gradient in those entries.  x: the transform (not varied here). */
int  lrr_icp_cdc_objective_cov(lrr_icp_cdc_set *set, const lrr_icp_xform *x, int dof, const double *x_upper,
                               const lrr_icp_norm *norm, double *f, double *g, int64_t *pairs);

/* test hook (no reference counterpart): overwrites the stored match of from-feature from_idx (original index) with `value`
 * as it is — also a value outside the to-cloud, which K1 never writes: the residual passes of the estimate that follows
 * must answer with LRR_ICP_ERR_STATE ("match index out of range") instead of gathering from a wild address */
int  lrr_icp_debug_set_match(lrr_icp_dir *dir, int64_t from_idx, int64_t value);

/* stand-alone exact nearest neighbour of host query points (tests, diagnostics) */
int  lrr_icp_nearest(lrr_icp_cloud *cloud, const double *query, int64_t nq, int method, int64_t *idx);

#ifdef __cplusplus
}
#endif
#endif
