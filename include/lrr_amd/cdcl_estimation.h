// cdcl_estimation.h — the covariance-driven EM estimator of cdcl/cdcl_estimation.{h,txx} on the device pieces of
// include/lrr_icp.h: compute_weights (E-step), estimate_parameters and estimate_covariance
// (M-step) and the one_iteration that strings them together, for the affine transform.
//
//   cdcl_obj_fun_par<3, dof> / cdcl_obj_fun_cov<3, dof>   compute(x, f, g): the two cost functions the reference hands to its
//       optimiser (cdcl/cdcl_obj_fun_par.txx:44-142, cdcl/cdcl_obj_fun_cov.txx:42-134), evaluated over match rows that
//       stay on the device.  With VXL at hand they plug into vnl_lbfgs as the reference's do (same compute() signature).
//   cdcl_lbfgs            stand-in for the reference's wrapper of vnl_lbfgs (third party, absent): lbfgs_standin.h
//   cdcl_estimation_em    compute_weights / estimate_parameters / estimate_covariance / one_iteration of
//       cdcl_estimation<3,12> (cdcl/cdcl_estimation.txx:95-187, 210-338, 342-545)
//
// Affine (dof 12) and rigid with the small-angle increment (dof 6: pre-mapped feature copies, recompose_increment, the second
// compute_weights on the global transform, :111-182), resolution switching inside compute_weights with the statistic and
// the restart's from_weights quirk (:356-358, 491-517), the closest-point fallback (:543-544), cdcl_lbfgs' 5 % stop.
// Not provided: the ICP base class that initialises the reference's estimator (cdcl_estimation_ICP::estimate_LS for the
// initial covariance, the subsampled pyramid itself): the caller passes the clouds of every resolution and an initial
// transform with its covariance.
#ifndef LRR_AMD_CDCL_ESTIMATION_H_
#define LRR_AMD_CDCL_ESTIMATION_H_

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "lrr_amd/icp_engine.h"
#include "lrr_amd/lbfgs_standin.h"

namespace lrr_amd {

// match rows of a moving / fixed pair, resident on the device
// The device cloud of a vector of cdcl_feature objects (location_, covariance_, strength_: cdcl/cdcl_feature.h:21-64), n x 3 / n x 9 / n
// row-major.  Such a feature has no error projector of its own: the cloud's projector slot takes the one the EM class's base derives from
// the covariance wherever it needs one, the outer product of the eigenvector of the smallest eigenvalue — the point normal
// (NORMAL_DISTANCES_FOR_ICP, cdcl/cdcl_estimation_ICP.txx:523-537 and :696-709).  cdcl_estimation_em_t::initialize() relies on it.
inline lrr_icp_cloud *cdcl_feature_cloud(int64_t n, const double *loc, const double *cov9, const double *strength) {
  if (n < 0 || (n > 0 && (!loc || !cov9 || !strength))) throw std::runtime_error("cdcl_feature_cloud: missing locations, covariances or strengths");
  std::vector<double> proj(9 * (size_t)std::max<int64_t>(n, 1));
  for (int64_t i = 0; i < n; ++i) {
    double U[9], sv[3], V[9];
    linalg::svd3(cov9 + 9 * i, U, sv, V);  // symmetric positive semi-definite: singular vectors = eigenvectors
    int lo = 0;
    for (int j = 1; j < 3; ++j) if (sv[j] < sv[lo]) lo = j;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) proj[9 * (size_t)i + 3 * a + b] = V[3 * a + lo] * V[3 * b + lo];
  }
  lrr_icp_cloud *c = nullptr;
  check(lrr_icp_cloud_create(device::ctx(), n, loc, proj.data(), strength, 0, &c), "cloud upload");
  if (lrr_icp_cloud_set_covariances(c, cov9) != LRR_ICP_OK) {
    lrr_icp_cloud_destroy(c);
    throw std::runtime_error("cdcl_feature_cloud: covariance upload failed");
  }
  return c;
}

class cdc_match_rows {
 public:
  cdc_match_rows(lrr_icp_cloud *moving, lrr_icp_cloud *fixed, int max_matches = 100) : set_(nullptr), moving_(moving), fixed_(fixed) {
    check(lrr_icp_cdc_set_create(moving, fixed, max_matches, &set_), "cdc_set_create");
  }
  ~cdc_match_rows() { if (set_) lrr_icp_cdc_set_destroy(set_); }
  cdc_match_rows(const cdc_match_rows &) = delete;
  cdc_match_rows &operator=(const cdc_match_rows &) = delete;
  lrr_icp_cdc_set *handle() const { return set_; }
  lrr_icp_cloud *moving() const { return moving_; }
  lrr_icp_cloud *fixed() const { return fixed_; }

 private:
  lrr_icp_cdc_set *set_;
  lrr_icp_cloud *moving_, *fixed_;
};

// (cdcl/cdcl_obj_fun_par.h:24-52) constructor: matches + transform (its centre and covariance enter J_th C_theta J_th^T);
// compute(): value and gradient at the parameter vector x (A row-major, then t)
template <unsigned dim, unsigned dof>
class cdcl_obj_fun_par {
 public:
  cdcl_obj_fun_par(cdc_match_rows const &rows, typename cdcl_trans<dim, dof>::sptr const &trans, const lrr_icp_norm *norm)
      : rows_(rows), trans_(trans), norm_(norm ? *norm : lrr_icp_norm()), has_norm_(norm != nullptr) {
    static_assert(dim == 3 && (dof == 12 || dof == 6), "affine or rigid transform in 3-D");
    vnl_matrix_fixed<double, dof, dof> const &C = trans_->get_covariance();
    for (unsigned i = 0; i < dof; ++i) for (unsigned j = 0; j < dof; ++j) covar_[i * dof + j] = C(i, j);
  }
  void compute(const vnl_vector<double> &x, double *f, vnl_vector<double> *g) {
    trans_->set_parameterization(x);  // (:50) update transformation
    lrr_icp_xform xf;
    vnl_matrix_fixed<double, 3, 3> &A = trans_->get_A();
    vnl_vector_fixed<double, 3> &t = trans_->get_translation();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) xf.A[3 * r + c] = A(r, c); xf.t[r] = t[r]; xf.c[r] = trans_->center_moving_[r]; }
    double fv = 0.0, gv[dof];
    check(lrr_icp_cdc_objective_par(rows_.handle(), &xf, (int)dof, covar_, has_norm_ ? &norm_ : nullptr, &fv, g ? gv : nullptr, nullptr),
          "cdc objective (parameters)");
    if (f) *f = fv;
    if (g) for (unsigned i = 0; i < dof; ++i) (*g)[i] = gv[i];
  }

 private:
  cdc_match_rows const &rows_;
  typename cdcl_trans<dim, dof>::sptr trans_;
  lrr_icp_norm norm_;
  bool has_norm_;
  double covar_[dof * dof];
};

// (cdcl/cdcl_obj_fun_cov.h) compute(): value and gradient at the dof (dof + 1) / 2 entries of the upper-triangular
// Cholesky factor of the parameter covariance; sets the transform's covariance to U^T U (:57)
template <unsigned dim, unsigned dof>
class cdcl_obj_fun_cov {
 public:
  cdcl_obj_fun_cov(cdc_match_rows const &rows, typename cdcl_trans<dim, dof>::sptr const &trans, const lrr_icp_norm *norm)
      : rows_(rows), trans_(trans), norm_(norm ? *norm : lrr_icp_norm()), has_norm_(norm != nullptr) {
    static_assert(dim == 3 && (dof == 12 || dof == 6), "affine or rigid transform in 3-D");
  }
  void compute(const vnl_vector<double> &x, double *f, vnl_vector<double> *g) {
    const unsigned ncov = dof * (dof + 1) / 2;
    vnl_matrix_fixed<double, dof, dof> U(0.0);
    unsigned ind = 0;
    for (unsigned d1 = 0; d1 < dof; ++d1) for (unsigned d2 = d1; d2 < dof; ++d2) U(d1, d2) = x[ind++];
    trans_->set_covariance(U.transpose() * U);
    lrr_icp_xform xf;
    vnl_matrix_fixed<double, 3, 3> &A = trans_->get_A();
    vnl_vector_fixed<double, 3> &t = trans_->get_translation();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) xf.A[3 * r + c] = A(r, c); xf.t[r] = t[r]; xf.c[r] = trans_->center_moving_[r]; }
    double fv = 0.0, gv[ncov], xv[ncov];
    for (unsigned i = 0; i < ncov; ++i) xv[i] = x[i];
    check(lrr_icp_cdc_objective_cov(rows_.handle(), &xf, (int)dof, xv, has_norm_ ? &norm_ : nullptr, &fv, g ? gv : nullptr, nullptr),
          "cdc objective (covariance)");
    if (f) *f = fv;
    if (g) for (unsigned i = 0; i < ncov; ++i) (*g)[i] = gv[i];
  }

 private:
  cdc_match_rows const &rows_;
  typename cdcl_trans<dim, dof>::sptr trans_;
  lrr_icp_norm norm_;
  bool has_norm_;
};

// The interface of cdcl_lbfgs the estimator uses (cdcl/cdcl_lbfgs.h; cdcl/cdcl_estimation.txx:230-248), over the stand-in
// optimiser.  F: a class with compute(const vnl_vector<double> &, double *, vnl_vector<double> *).
template <class F>
class cdcl_lbfgs {
 public:
  explicit cdcl_lbfgs(F &f) : f_(f), max_evals_(5) { res_.f_start = res_.f_end = 0.0; res_.evals = 0; res_.code = 0; }
  void set_max_function_evals(int n) { max_evals_ = n; }
  void set_verbose(bool) {}
  void set_check_derivatives(bool) {}
  void set_trace(bool) {}
  bool minimize(vnl_vector<double> &x) {
    n_ = (int)x.size();
    std::vector<double> v((size_t)n_);
    for (int i = 0; i < n_; ++i) v[(size_t)i] = x[(unsigned)i];
    // (cdcl/cdcl_lbfgs.h:28-36: report_iter() ends the minimisation once the objective has changed by 5 % of its start value)
    res_ = lrr_lbfgs_minimize_stop(n_, v.data(), &cdcl_lbfgs::thunk, this, max_evals_, LRR_CDCL_LBFGS_STOP_FRACTION);
    if (res_.code > 0) for (int i = 0; i < n_; ++i) x[(unsigned)i] = v[(size_t)i];
    return res_.code == LRR_LBFGS_CONVERGED;
  }
  double get_start_error() const { return res_.f_start; }
  double get_end_error() const { return res_.f_end; }
  double obj_fun_change() const { return std::fabs(res_.f_start - res_.f_end); }  // (cdcl/cdcl_lbfgs.h:25)
  int get_failure_code() const { return res_.code; }
  int get_num_evaluations() const { return res_.evals; }

 private:
  static void thunk(void *self, const double *x, double *f, double *g) {
    cdcl_lbfgs *me = static_cast<cdcl_lbfgs *>(self);
    vnl_vector<double> xv((unsigned)me->n_, 0.0), gv((unsigned)me->n_, 0.0);
    for (int i = 0; i < me->n_; ++i) xv[(unsigned)i] = x[i];
    me->f_.compute(xv, f, &gv);
    for (int i = 0; i < me->n_; ++i) g[i] = gv[(unsigned)i];
  }
  F &f_;
  int max_evals_, n_ = 0;
  lrr_lbfgs_result res_;
};

// The EM loop of cdcl_estimation<3, dof> around the device pieces (cdcl/cdcl_estimation.txx:95-187, 210-338, 342-545), for the
// affine transform (dof 12) and for the rigid one with its small-angle increment (dof 6), at one resolution or over a
// multi-resolution pyramid (the second constructor of the reference, cdcl/cdcl_estimation.txx:54-63).
//   moving / fixed: device clouds with covariances (lrr_icp_cloud_set_covariances), [0] the finest resolution;
//   fixed_spacing[r]: the point spacing of fixed[r] the resolution switch compares with (cdcl_estimation.txx:503-517);
//   trans: the transform with its covariance, modified in place as the reference's estimator modifies the caller's object.
// The estimator starts at the coarsest resolution (cdcl/cdcl_estimation_ICP.txx:118).
template <unsigned dof>
class cdcl_estimation_em_t {
 public:
  static_assert(dof == 6 || dof == 12, "rigid3d (6) or affine (12)");
  typedef typename cdcl_trans<3, dof>::sptr trans_sptr_type;
  cdcl_estimation_em_t(lrr_icp_cloud *moving, lrr_icp_cloud *fixed, trans_sptr_type const &trans, uint64_t seed = 1, int max_matches = 100)
      : trans_(trans), seed_(seed), normalize_matches_(true), overflow_rows_(0), res_(0), max_res_(0) {
    rows_.push_back(std::unique_ptr<cdc_match_rows>(new cdc_match_rows(moving, fixed, max_matches)));
    fixed_spacing_.push_back(1.0);  // (cdcl/cdcl_estimation_ICP.txx:84)
    std::memset(&last_, 0, sizeof last_);
  }
  cdcl_estimation_em_t(std::vector<lrr_icp_cloud *> const &moving, std::vector<lrr_icp_cloud *> const &fixed, trans_sptr_type const &trans,
                       std::vector<double> const &fixed_spacing, uint64_t seed = 1, int max_matches = 100)
      : trans_(trans), seed_(seed), normalize_matches_(true), overflow_rows_(0), fixed_spacing_(fixed_spacing) {
    const size_t levels = std::min(moving.size(), fixed.size());
    if (levels == 0 || fixed_spacing.size() < levels) throw std::runtime_error("cdcl_estimation_em: empty pyramid or missing spacings");
    for (size_t r = 0; r < levels; ++r) rows_.push_back(std::unique_ptr<cdc_match_rows>(new cdc_match_rows(moving[r], fixed[r], max_matches)));
    res_ = max_res_ = (unsigned)levels - 1;  // set the coarsest resolution (cdcl/cdcl_estimation_ICP.txx:118)
    std::memset(&last_, 0, sizeof last_);
  }
  void normalize_matches(bool on = true) { normalize_matches_ = on; }
  trans_sptr_type get_transform() const { return trans_; }
  cdc_match_rows const &rows() const { return *rows_[res_]; }
  unsigned resolution() const { return res_; }
  int64_t overflow_rows() const { return overflow_rows_; }
  int64_t matches_size() const { return matches_size_; }
  bool used_closest_euclidean() const { return fell_back_; }
  double avg_largest_eval() const { return avg_largest_eval_; }  // of the last pass of compute_weights (:491)
  // resolution switches (to a finer, to a coarser level) since one_iteration() began
  unsigned switches_down() const { return switches_down_; }
  unsigned switches_up() const { return switches_up_; }
  struct step_info { double par_start, par_end, cov_start, cov_end; int par_code, cov_code, par_evals, cov_evals; };
  step_info const &last_step() const { return last_; }

  // (cdcl/cdcl_estimation.txx:342-545)  trans: the transform the features are mapped and the Jacobians taken with (default:
  // trans_); the parameter covariance is trans_'s either way (transfer_covar, cdcl_estimation.h:96-98)
  void compute_weights(bool resolution_switch_allowed = false, trans_sptr_type trans = trans_sptr_type()) {
    if (!trans) trans = trans_;
    lrr_icp_xform xf = xform(trans);
    double C[dof * dof];
    covariance(C);
    std::vector<double> from_weights;  // (:356-358: declared outside the restart loop — a restart appends, it does not clear)
    std::vector<int64_t> first_counts;
    bool stop = false, not_restarted = true;
    fell_back_ = false;
    while (!stop) {
      uint32_t keys[4];
      lrr_icp_perm_keys(&seed_, keys);
      // (rows with more candidates inside the radius than the device holds are an error: the reference has no cap)
      check(lrr_icp_cdc_set_compute_weights(rows_[res_]->handle(), &xf, (int)dof, C, keys, nullptr), "cdc compute_weights");
      overflow_rows_ = 0;
      double eval_sum = 0.0;
      check(lrr_icp_cdc_set_stats(rows_[res_]->handle(), &eval_sum, &matches_size_), "cdc statistics");
      const double avg_largest_eval = eval_sum / (double)matches_size_;  // (:491; 0 matches: inf / nan, as the reference)
      avg_largest_eval_ = avg_largest_eval;
      const bool restart_pass = !not_restarted;
      if (!restart_pass && rows_.size() > 1 && resolution_switch_allowed) {
        // keep this pass's from_weights in match order: the second pass is normalised with them (see below)
        const int64_t nm = lrr_icp_cloud_size(moving_of(res_));
        std::vector<double> rs((size_t)nm);
        std::vector<int64_t> rc((size_t)nm);
        check(lrr_icp_cdc_set_row_sums(rows_[res_]->handle(), rs.data(), rc.data()), "cdc row sums");
        for (int64_t i = 0; i < nm; ++i) if (rc[(size_t)i] > 0) from_weights.push_back(rs[(size_t)i]);
      }
      stop = true;
      if (avg_largest_eval < 2.0 * fixed_spacing_[res_] && resolution_switch_allowed && not_restarted) {
        if (res_ > 0) { --res_; stop = false; not_restarted = false; ++switches_down_; }
      }
      if (res_ + 1 <= max_res_ && avg_largest_eval > 2.0 * fixed_spacing_[res_ + 1] && resolution_switch_allowed && not_restarted) {
        ++res_; stop = false; not_restarted = false; ++switches_up_;
      }
      if (stop && restart_pass && !from_weights.empty()) {
        // match i of this pass takes from_weights[i]: an entry of the FIRST pass while there are any (:524-530 index the
        // vector by the match's position; the restart appended this pass's sums behind the first pass's)
        const int64_t nm = lrr_icp_cloud_size(moving_of(res_));
        std::vector<double> rs((size_t)nm);
        std::vector<int64_t> rc((size_t)nm);
        check(lrr_icp_cdc_set_row_sums(rows_[res_]->handle(), rs.data(), rc.data()), "cdc row sums");
        std::vector<double> all = from_weights;
        for (int64_t i = 0; i < nm; ++i) if (rc[(size_t)i] > 0) all.push_back(rs[(size_t)i]);
        size_t k = 0;
        for (int64_t i = 0; i < nm; ++i) if (rc[(size_t)i] > 0) rs[(size_t)i] = all[k++];
        check(lrr_icp_cdc_set_renormalize(rows_[res_]->handle(), rs.data()), "cdc renormalise");
      }
    }
    // covariance is probably too small for any matches to be found, use closest points then (:543-544)
    if (matches_size_ == 0) {
      lrr_icp_xform own = xform(trans_);
      check(lrr_icp_cdc_set_closest_euclidean(rows_[res_]->handle(), &own), "cdc closest points");
      matches_size_ = lrr_icp_cloud_size(moving_of(res_));
      fell_back_ = true;
    }
  }
  // (:210-260)
  bool estimate_parameters() {
    lrr_icp_norm nm;
    normalize(nm);
    cdcl_obj_fun_par<3, dof> f_par(*rows_[res_], trans_, normalize_matches_ ? &nm : nullptr);
    vnl_vector<double> x = trans_->get_parameterization(), x_old = x;
    cdcl_lbfgs<cdcl_obj_fun_par<3, dof> > minimizer(f_par);
    minimizer.set_max_function_evals(5);
    bool converged = minimizer.minimize(x);
    bool parameters_converged = (minimizer.get_start_error() - minimizer.get_end_error()) < 1e-4;
    if ((!converged && minimizer.get_failure_code() < 0) || x.size() == 0) trans_->set_parameterization(x_old);
    else trans_->set_parameterization(x);
    unnormalize(nm);
    last_.par_start = minimizer.get_start_error(); last_.par_end = minimizer.get_end_error();
    last_.par_code = minimizer.get_failure_code(); last_.par_evals = minimizer.get_num_evaluations();
    return parameters_converged;
  }
  // (:264-338)
  bool estimate_covariance() {
    lrr_icp_norm nm;
    normalize(nm);
    cdcl_obj_fun_cov<3, dof> f_cov(*rows_[res_], trans_, normalize_matches_ ? &nm : nullptr);
    const unsigned cov_dof = dof * (dof + 1) / 2;
    // upper triangular Cholesky factor of the covariance -> x (:283-293)
    double C[dof * dof], U[dof * dof];
    covariance(C);
    if (!chol_upper(C, U)) {
      // (the reference's vnl_cholesky runs in quiet mode and carries on with whatever it produced; here the estimate fails —
      // after the transform has been put back into un-normalised coordinates, so the caller's object stays usable)
      unnormalize(nm);
      throw std::runtime_error("cdcl_estimation_em: the parameter covariance is not positive definite");
    }
    vnl_vector<double> x(cov_dof, 0.0);
    unsigned ind = 0;
    for (unsigned d1 = 0; d1 < dof; ++d1) for (unsigned d2 = d1; d2 < dof; ++d2) x[ind++] = U[d1 * dof + d2];
    vnl_vector<double> x_old = x;
    cdcl_lbfgs<cdcl_obj_fun_cov<3, dof> > minimizer(f_cov);
    minimizer.set_max_function_evals(5);
    bool converged = minimizer.minimize(x);
    bool covariance_converged = (minimizer.get_start_error() - minimizer.get_end_error()) < 1e-5;
    if ((!converged && minimizer.get_failure_code() < 0) || x.size() == 0) x = x_old;
    // the covariance from its factor (:313-323)
    vnl_matrix_fixed<double, dof, dof> upper(0.0);
    ind = 0;
    for (unsigned d1 = 0; d1 < dof; ++d1) for (unsigned d2 = d1; d2 < dof; ++d2) upper(d1, d2) = x[ind++];
    trans_->set_covariance(upper.transpose() * upper);
    unnormalize(nm);
    last_.cov_start = minimizer.get_start_error(); last_.cov_end = minimizer.get_end_error();
    last_.cov_code = minimizer.get_failure_code(); last_.cov_evals = minimizer.get_num_evaluations();
    return covariance_converged;
  }
  // (:95-187)
  bool one_iteration(unsigned iteration) {
    const unsigned startEM = 5;
    const bool resolution_switch_allowed = iteration > startEM;
    switches_down_ = switches_up_ = 0;
    this->compute_weights(resolution_switch_allowed);
    trans_sptr_type global_transform;
    const bool rigid = is_rigid();
    if (rigid) {  // from now on the transform is incremental (:111-141)
      global_transform = trans_;
      trans_ = trans_->create_new();  // identity
      trans_->set_covariance(global_transform->get_covariance());
      premap(global_transform);       // every matches_[i]->from_ becomes the copy at G(p) with covariance Jp C Jp^T
      trans_->set_parameterization(vnl_vector<double>(dof, 0.0));
    }
    bool parameters_converged = false, covariance_converged = false;
    if (iteration > startEM) {
      parameters_converged = this->estimate_parameters();
      if (rigid) {
        global_transform->recompose_increment(trans_);                        // (:152)
        this->compute_weights(resolution_switch_allowed, global_transform);   // (:153)
        premap(global_transform);                                             // (:156-170)
      } else {
        this->compute_weights(resolution_switch_allowed);
      }
    }
    covariance_converged = this->estimate_covariance();
    if (rigid) {  // (:178-182)
      global_transform->set_covariance(trans_->get_covariance());
      trans_ = global_transform;
      check(lrr_icp_cdc_set_premap(rows_[res_]->handle(), nullptr), "cdc premap");
    }
    double fro = 0.0;
    vnl_matrix_fixed<double, dof, dof> const &C = trans_->get_covariance();
    for (unsigned i = 0; i < dof; ++i) for (unsigned j = 0; j < dof; ++j) fro += C(i, j) * C(i, j);
    return parameters_converged && covariance_converged && std::sqrt(fro) < 1e-3;
  }

  // (cdcl/cdcl_estimation.txx:70-84) initial weights and covariance from the base class cdcl_estimation_ICP at the current resolution:
  // find_closest_euclidean() over ALL moving features (cdcl_estimation_ICP.txx:451-489) and estimate_LS() without the parameters
  // (:646-750): residuals along the normal of the fixed feature's covariance (the projector cdcl_feature_cloud() stored), scale =
  // 2 x MUSE (:557-582), Beaton-Tukey weights, trans_'s covariance = inverse of AtA.  weight_by_strength_ of the base is false and
  // nothing in the reference sets it on this class: not offered.  The clouds must come from cdcl_feature_cloud().
  void initialize() {
    lrr_icp_dir *dir = closest_point_list(res_);
    const int64_t nm = lrr_icp_cloud_size(moving_of(res_));
    if (nm < 1) throw std::runtime_error("cdcl_estimation_em: initialize() without moving features");
    std::vector<int64_t> order((size_t)nm);
    for (int64_t i = 0; i < nm; ++i) order[(size_t)i] = i;  // moving_[res_] in its own order: no draw from the random stream
    const uint32_t no_keys[4] = {0, 0, 0, 0};
    lrr_icp_xform xf = xform(trans_);
    check(lrr_icp_match(dir, &xf, no_keys, order.data(), nm, LRR_ICP_NN_GRID), "closest points");
    lrr_icp_norm nm_stats;
    for (int d = 0; d < 3; ++d) nm_stats.cm[d] = nm_stats.cf[d] = 0.0;
    nm_stats.rm = nm_stats.rf = 1.0;
    vnl_vector_fixed<double, 3> cm(0.0), cf(0.0);
    if (normalize_matches_) {  // (:654-657)
      check(lrr_icp_normalize_stats(dir, &nm_stats), "normalize statistics");
      for (int d = 0; d < 3; ++d) { cm[d] = nm_stats.cm[d]; cf[d] = nm_stats.cf[d]; }
      trans_->normalize(nm_stats.rm, nm_stats.rf, cm, cf);
    }
    xf = xform(trans_);
    double S = 0.0, W = 0.0, sigma = 0.0;
    check(lrr_icp_residual_pass(dir, &xf, &nm_stats, &S, &W), "residual pass");
    check(lrr_icp_muse_scale(dir, &sigma), "MUSE scale");  // (the MAD scale computed first is overwritten, :562-572)
    initial_scale_ = 2.0 * sigma;                           // (:580-582)
    check(lrr_icp_assign_weights(dir, initial_scale_), "assign weights");
    double AtA[dof * dof], Atb[dof];
    const double centre[3] = {trans_->center_moving_[0], trans_->center_moving_[1], trans_->center_moving_[2]};
    check(lrr_icp_accumulate(dir, (int)dof, centre, &nm_stats, is_rigid() ? LRR_ICP_ACC_RIGID_INCREMENT : 0, AtA, Atb), "normal equations");
    solve_normal_equations<dof>(trans_, AtA, Atb, false);
    if (normalize_matches_) trans_->unnormalize(nm_stats.rm, nm_stats.rf, cm, cf);  // (:738-742)
  }
  double initial_scale() const { return initial_scale_; }
  // (cdcl/cdcl_estimation.txx:190-204)
  bool run(unsigned *iterations_done = nullptr) {
    this->initialize();
    bool converged = false;
    const unsigned max_iterations = 85;
    unsigned iteration = 0;
    while (!converged && iteration < max_iterations) {
      converged = this->one_iteration(iteration);
      ++iteration;
    }
    if (iterations_done) *iterations_done = iteration;
    return converged;
  }
  ~cdcl_estimation_em_t() { for (lrr_icp_dir *d : closest_) if (d) lrr_icp_dir_destroy(d); }
  cdcl_estimation_em_t(const cdcl_estimation_em_t &) = delete;
  cdcl_estimation_em_t &operator=(const cdcl_estimation_em_t &) = delete;

 private:
  bool is_rigid() const { return dof == 6 && trans_->is_type(cdcl_trans_rigid3d::type_id()); }
  lrr_icp_cloud *moving_of(unsigned r) const { return rows_[r]->moving(); }
  // the one-to-one match list of the base class at resolution r (created on first use)
  lrr_icp_dir *closest_point_list(unsigned r) {
    if (closest_.size() < rows_.size()) closest_.resize(rows_.size(), nullptr);
    if (!closest_[r]) check(lrr_icp_dir_create(device::ctx(), rows_[r]->moving(), rows_[r]->fixed(), &closest_[r]), "dir create");
    return closest_[r];
  }
  void premap(trans_sptr_type const &G) {
    lrr_icp_xform g = xform(G);
    check(lrr_icp_cdc_set_premap(rows_[res_]->handle(), &g), "cdc premap");
  }
  static lrr_icp_xform xform(trans_sptr_type const &T) {
    lrr_icp_xform x;
    vnl_matrix_fixed<double, 3, 3> &A = T->get_A();
    vnl_vector_fixed<double, 3> &t = T->get_translation();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) x.A[3 * r + c] = A(r, c); x.t[r] = t[r]; x.c[r] = T->center_moving_[r]; }
    return x;
  }
  void covariance(double *C) const {
    vnl_matrix_fixed<double, dof, dof> const &M = trans_->get_covariance();
    for (unsigned i = 0; i < dof; ++i) for (unsigned j = 0; j < dof; ++j) C[i * dof + j] = M(i, j);
  }
  // cdcl_normalize_matches + trans_->normalize (:218-221); the features themselves are normalised on the fly by the kernels
  void normalize(lrr_icp_norm &nm) {
    if (!normalize_matches_) return;
    check(lrr_icp_cdc_normalize_stats(rows_[res_]->handle(), &nm), "cdc normalize statistics");
    vnl_vector_fixed<double, 3> cm, cf;
    for (int d = 0; d < 3; ++d) { cm[d] = nm.cm[d]; cf[d] = nm.cf[d]; }
    trans_->normalize(nm.rm, nm.rf, cm, cf);
  }
  void unnormalize(lrr_icp_norm const &nm) {
    if (!normalize_matches_) return;
    vnl_vector_fixed<double, 3> cm, cf;
    for (int d = 0; d < 3; ++d) { cm[d] = nm.cm[d]; cf[d] = nm.cf[d]; }
    trans_->unnormalize(nm.rm, nm.rf, cm, cf);
  }
  static bool chol_upper(const double *C, double *U) {
    const int n = (int)dof;
    std::memset(U, 0, sizeof(double) * dof * dof);
    for (int i = 0; i < n; ++i)
      for (int j = i; j < n; ++j) {
        double s = C[i * n + j];
        for (int k = 0; k < i; ++k) s -= U[k * n + i] * U[k * n + j];
        if (i == j) { if (!(s > 0.0)) return false; U[i * n + i] = std::sqrt(s); }
        else U[i * n + j] = s / U[i * n + i];
      }
    return true;
  }

  std::vector<std::unique_ptr<cdc_match_rows> > rows_;  // one set of match rows per resolution
  trans_sptr_type trans_;
  uint64_t seed_;
  bool normalize_matches_;
  int64_t overflow_rows_;
  std::vector<double> fixed_spacing_;
  unsigned res_, max_res_;
  int64_t matches_size_ = 0;
  bool fell_back_ = false;
  double avg_largest_eval_ = 0.0;
  unsigned switches_down_ = 0, switches_up_ = 0;
  step_info last_;
  std::vector<lrr_icp_dir *> closest_;
  double initial_scale_ = 0.0;
};
typedef cdcl_estimation_em_t<12> cdcl_estimation_em;
typedef cdcl_estimation_em_t<6> cdcl_estimation_em_rigid;

}  // namespace lrr_amd
#endif
