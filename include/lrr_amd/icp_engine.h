// icp_engine.h — host side of the MI355X robust-ICP estimators.
//
// Two class templates carry the logic of the reference's four estimator classes:
//   icp_direction<dof, Feature, RRL>  <-> cdcl_estimation_ICP_matching_all (RRL = false,
//        cdcl/cdcl_estimation_ICP_matching_all.{h,txx}) and rrl_estimation_ICP_matching_all
//        (RRL = true, rrl/rrl_estimation_ICP_matching_all.{h,txx})
//   icp_symmetric<dof, Feature, RRL>  <-> cdcl_/rrl_estimation_symmetric_ICP_matching_all
// Same method names, argument meaning and state machine (number_points_ schedule,
// scale_estimated_, oscillation counters); every per-match loop of the reference is one call
// into the C-ABI of include/lrr_icp.h.  The host keeps only O(dof^2) arithmetic: transform
// bookkeeping, the dof x dof solve, convergence logic.
//
// What differs from the reference by design:
//   * feature objects are never mutated: cdcl_normalize_matches() computes the centres/radii
//     and the kernels normalise on the fly (the reference rewrites location_ in place and
//     restores it, cdcl .txx:1159-1286);
//   * std::random_shuffle(rand()) is a seeded keyed permutation evaluated on the device;
//   * matches_ lives in HBM and is materialised as cdcl_match objects only on demand
//     (callbacks, sheet/tube angles);
//   * failures of the device layer throw std::runtime_error (the reference prints and goes on).
#ifndef LRR_AMD_ICP_ENGINE_H_
#define LRR_AMD_ICP_ENGINE_H_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "lrr_amd/cdcl_types.h"
#include "lrr_amd/linalg.h"
#include "lrr_icp.h"

// ---- abstract interface (cdcl/cdcl_estimation_abs.h:20-65) -----------------------------------
#ifdef LRR_AMD_USE_REFERENCE_TYPES
#include "cdcl_estimation_abs.h"
#else
template <unsigned int dim, unsigned int dof, class feature_typeT = cdcl_feature<dim> >
class cdcl_estimation_abs : public vbl_ref_count {
 public:
  typedef vbl_smart_ptr<cdcl_estimation_abs> sptr;
  typedef feature_typeT feature_type;
  typedef typename feature_type::sptr feature_sptr_type;
  typedef cdcl_match<dim, feature_type> match_type;
  typedef cdcl_trans<dim, dof> trans_type;
  typedef typename match_type::sptr match_sptr_type;
  typedef typename trans_type::sptr trans_sptr_type;
  typedef void (*display_callback_type)(void *caller, vcl_vector<feature_sptr_type> const &moving_,
                                        vcl_vector<feature_sptr_type> const &fixed_,
                                        vcl_vector<match_sptr_type> const &matches_,
                                        trans_sptr_type const &transform, unsigned int iteration);
  cdcl_estimation_abs() {}
  virtual void initialize() = 0;
  virtual bool run(void *caller = 0, display_callback_type display_points = 0) = 0;
};
#endif

namespace lrr_amd {

// One device context per host thread, created on first use on the device selected by
// set_device() or the LRR_ICP_DEVICE environment variable (default 0).
class device {
 public:
  static void set_device(int index) { slot().want = index; }
  static lrr_icp_ctx *ctx() {
    state &s = slot();
    if (!s.ctx) {
      int dev = s.want;
      if (dev < 0) {
        const char *e = std::getenv("LRR_ICP_DEVICE");
        dev = e ? std::atoi(e) : 0;
      }
      if (lrr_icp_create(dev, &s.ctx) != LRR_ICP_OK || !s.ctx)
        throw std::runtime_error("lrr_amd: cannot create a HIP context on device " + std::to_string(dev) +
                                 " (no usable MI355X / HIP runtime): the ICP estimators have no CPU path");
    }
    return s.ctx;
  }
  static void release() {
    state &s = slot();
    if (s.ctx) lrr_icp_destroy(s.ctx);
    s.ctx = nullptr;
  }

 private:
  struct state { lrr_icp_ctx *ctx = nullptr; int want = -1; };
  static state &slot() { static thread_local state s; return s; }
};

// how set_voronoi_maps() / set_fixed_voronoi_map() of the rrl classes are honoured (see icp_symmetric): by default the call
// selects the SEMANTICS of the reference's map lookup (LRR_ICP_NN_VORONOI: what the production binary computes);
// voronoi_lookup::enable(false) — or LRR_ICP_VORONOI=0 in the environment — keeps the exact nearest-neighbour search
// (faster on the device and exact where the 1 mm maps approximate, but not the reference's match sets)
struct voronoi_lookup {
  static void enable(bool on) { flag() = on ? 1 : 0; }
  static bool enabled() {
    if (flag() < 0) {
      const char *e = std::getenv("LRR_ICP_VORONOI");
      flag() = (e && e[0] == '0') ? 0 : 1;
    }
    return flag() == 1;
  }
  static void note_exact() {
    static bool said = false;
    if (!said) {
      said = true;
      std::fprintf(stderr, "lrr_amd: set_voronoi_maps(): the lookup semantics of the maps are switched off (LRR_ICP_VORONOI=0 / "
                           "voronoi_lookup::enable(false)) — matches are the exact nearest neighbours, which the 1 mm maps approximate\n");
    }
  }
 private:
  static int &flag() { static int f = -1; return f; }
};

inline void check(int rc, const char *what) {
  if (rc != LRR_ICP_OK) throw std::runtime_error(std::string("lrr_amd: ") + what + " failed: " + lrr_icp_last_error(device::ctx()));
}

// cdcl_feature_with_shape carries a covariance, whose eigenvectors the cdcl classes' sheet / tube angles compare
// (cdcl .txx:273-382); cdcl_feature_ICP has none and the rrl classes compare mapped directions_ (rrl .txx:281-375)
template <class F> struct feature_traits;
template <unsigned d> struct feature_traits<cdcl_feature_with_shape<d> > {
  static const bool float_strength = false;
  static const bool covariance_angles = true;
  // columns 0 and 2 of vnl_symmetric_eigensystem_compute (eigenvalues ascending) of the 3 x 3 covariance: for a
  // symmetric positive semi-definite matrix its singular vectors, picked by singular value
  static void angle_vectors(cdcl_feature_with_shape<d> const &f, double *sheet, double *tube) {
    double U[9], sv[3], V[9];
    linalg::svd3(f.covariance_.data_block(), U, sv, V);
    int lo = 0, hi = 0;
    for (int j = 1; j < 3; ++j) { if (sv[j] < sv[lo]) lo = j; if (sv[j] > sv[hi]) hi = j; }
    for (int k = 0; k < 3; ++k) { sheet[k] = V[3 * k + lo]; tube[k] = V[3 * k + hi]; }
  }
};
template <unsigned d> struct feature_traits<cdcl_feature_ICP<d> > {
  static const bool float_strength = true;
  static const bool covariance_angles = false;
  static void angle_vectors(cdcl_feature_ICP<d> const &, double *, double *) {}
};

// a feature vector resident on the device
template <class FeatureT>
class device_cloud {
 public:
  device_cloud() : h_(nullptr) {}
  ~device_cloud() { reset(); }
  device_cloud(const device_cloud &) = delete;
  device_cloud &operator=(const device_cloud &) = delete;
  void reset() { if (h_) lrr_icp_cloud_destroy(h_); h_ = nullptr; }
  void upload(const std::vector<typename FeatureT::sptr> &f) {
    reset();
    const size_t n = f.size();
    std::vector<double> loc(3 * n + 3), proj(9 * n + 9), st(n + 1);
    for (size_t i = 0; i < n; ++i) {
      for (int d = 0; d < 3; ++d) loc[3 * i + d] = f[i]->location_[d];
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) proj[9 * i + 3 * r + c] = f[i]->error_projector_(r, c);
      st[i] = (double)f[i]->strength_;
    }
    check(lrr_icp_cloud_create(device::ctx(), (int64_t)n, loc.data(), proj.data(), st.data(),
                               feature_traits<FeatureT>::float_strength ? 1 : 0, &h_), "cloud upload");
    if (feature_traits<FeatureT>::covariance_angles) {  // shape_ and the covariance eigenvectors (cdcl angle diagnostics)
      std::vector<uint8_t> shape(n + 1, 0);
      std::vector<double> sheet(3 * n + 3, 0.0), tube(3 * n + 3, 0.0);
      for (size_t i = 0; i < n; ++i) {
        shape[i] = (uint8_t)f[i]->shape_;
        feature_traits<FeatureT>::angle_vectors(*f[i], &sheet[3 * i], &tube[3 * i]);
      }
      check(lrr_icp_cloud_set_shape_vectors(h_, shape.data(), sheet.data(), tube.data()), "shape upload");
      return;
    }
    // shape_ and directions_ for the post-convergence diagnostics (sheet_angles / tube_angles)
    std::vector<uint8_t> shape(n + 1, 0), ndirs(n + 1, 0);
    std::vector<double> dirs(6 * n + 6, 0.0);
    for (size_t i = 0; i < n; ++i) {
      shape[i] = (uint8_t)f[i]->shape_;
      const size_t nd = f[i]->directions_.size();
      ndirs[i] = (uint8_t)(nd > 3 ? 3 : nd);
      for (size_t k = 0; k < nd && k < 2; ++k)
        for (int d = 0; d < 3; ++d) dirs[6 * i + 3 * k + d] = f[i]->directions_[k][d];
    }
    check(lrr_icp_cloud_set_shapes(h_, shape.data(), ndirs.data(), dirs.data()), "shape upload");
  }
  lrr_icp_cloud *handle() const { return h_; }

 private:
  lrr_icp_cloud *h_;
};

// C = svd(AtA).inverse(); covariance = C; theta = C Atb (cdcl .txx:1073-1083; rrl sym .txx:880-894)
template <unsigned dof>
inline void solve_normal_equations(typename cdcl_trans<3, dof>::sptr const &T, const double *AtA, const double *Atb,
                                   bool estimate_parameters) {
  vnl_matrix_fixed<double, dof, dof> invAA;
  linalg::sym_inverse((int)dof, AtA, invAA.data_block());
  T->set_covariance(invAA);
  if (estimate_parameters) {
    vnl_vector<double> p(dof, 0.0);
    for (unsigned i = 0; i < dof; ++i) {
      double s = 0.0;
      for (unsigned k = 0; k < dof; ++k) s += invAA(i, k) * Atb[k];
      p[i] = s;
    }
    T->set_parameterization(p);
  }
}

// symmetric_apply_solution(F, B, r, normalize = true, estimate_parameters = true) for an AFFINE forward transform,
// without the work whose results it overwrites: normalize() only zeroes the centre (its A, t and covariance are
// replaced by the solution), and the caller replaces the backward transform by inverse(forward) right after.  What
// remains: C = inverse(AtA), theta = C Atb (normalised coordinates), then unnormalize() (cdcl_trans_affine.txx:259-286):
// A <- (rf/rm) A, t <- rf t + cf, covar <- D C D with the diagonal D = (rf/rm x 9, rf x 3), centre <- cm.  Same
// operations in the same order: bit-identical results, without the four dense 12 x 12 products.
inline void affine_forward_update(cdcl_trans<3, 12>::sptr const &F, const lrr_icp_sym_out &r) {
  double C[144];
  linalg::sym_inverse(12, r.AtA, C);
  const double rm = r.norm.rm, rf = r.norm.rf, f = rf / rm;
  vnl_vector<double> p(12, 0.0);
  for (unsigned i = 0; i < 12; ++i) {
    double s = 0.0;
    for (unsigned k = 0; k < 12; ++k) s += C[i * 12 + k] * r.Atb[k];
    p[i] = i < 9 ? s * f : rf * s + r.norm.cf[i - 9];
  }
  F->set_parameterization(p);
  vnl_matrix_fixed<double, 12, 12> cov;
  for (unsigned i = 0; i < 12; ++i)
    for (unsigned j = 0; j < 12; ++j) cov(i, j) = ((i < 9 ? f : rf) * C[i * 12 + j]) * (j < 9 ? f : rf);
  F->set_covariance(cov);
  for (int d = 0; d < 3; ++d) F->center_moving_[d] = r.norm.cm[d];
}
template <unsigned dof> struct affine_shortcut {
  static bool apply(typename cdcl_trans<3, dof>::sptr const &, const lrr_icp_sym_out &) { return false; }
};
template <> struct affine_shortcut<12> {
  static bool apply(cdcl_trans<3, 12>::sptr const &F, const lrr_icp_sym_out &r) {
    if (!F->is_type(cdcl_trans_affine<3, 12>::type_id())) return false;
    affine_forward_update(F, r);
    return true;
  }
};

// Host half of the symmetric estimate_LS once the device has returned the statistics and the normal
// equations (rrl sym .txx:713-721, 880-912; cdcl sym .txx:470-478, 643-695): normalise both transforms
// (the backward one with flipped centres and radii), solve, un-normalise, backward = inverse(forward).
// F / B: the transforms in the forward and the backward ROLE (estimate_LS_backward swaps them).
template <unsigned dof>
inline void symmetric_apply_solution(typename cdcl_trans<3, dof>::sptr const &F, typename cdcl_trans<3, dof>::sptr const &B,
                                     const lrr_icp_sym_out &r, bool normalize, bool estimate_parameters,
                                     bool backward_is_replaced = false) {
  // (the symmetric classes replace the backward transform by inverse(forward) after a parameter estimate: then the
  // affine forward transform takes the short way)
  if (normalize && estimate_parameters && backward_is_replaced && affine_shortcut<dof>::apply(F, r)) return;
  vnl_vector_fixed<double, 3> cm(0.0), cf(0.0);
  double rm = 0.0, rf = 0.0;
  if (normalize) {
    for (int d = 0; d < 3; ++d) { cm[d] = r.norm.cm[d]; cf[d] = r.norm.cf[d]; }
    rm = r.norm.rm;
    rf = r.norm.rf;
    F->normalize(rm, rf, cm, cf);
    B->normalize(rf, rm, cf, cm);  // flipped centres and radii
  }
  solve_normal_equations<dof>(F, r.AtA, r.Atb, estimate_parameters);
  if (normalize) {
    F->unnormalize(rm, rf, cm, cf);
    B->unnormalize(rf, rm, cf, cm);
  }
}

// Convergence test, oscillation counters and coarse-to-fine schedule of the symmetric classes
// (rrl sym .txx:383-437; cdcl sym .txx:311-361).  step() takes the forward weighted error before and
// after the iteration; grow(n) sets number_points of both lists and returns true at the finest level.
struct symmetric_schedule {
  unsigned int oscillation_count_;
  double error_difference_, weighted_error_;
  unsigned int number_matches_;
  bool finest_level_;
  symmetric_schedule() { reset(); weighted_error_ = 0.0; }
  void reset() {
    oscillation_count_ = 0;
    error_difference_ = 0.0;
    number_matches_ = 1000;
    finest_level_ = false;
  }
  bool oscillated() const { return oscillation_count_ > 10; }
  template <class Grow>
  bool step(double weighted_error_before, double weighted_error_after, unsigned int iteration, bool rrl, Grow grow) {
    double curr_error_difference = weighted_error_after - weighted_error_before;
    double diff = (weighted_error_after - weighted_error_before) / weighted_error_after;
    bool converged = std::fabs(diff) < 1e-4;
    weighted_error_ = weighted_error_after;
    if (!rrl || iteration > 0) {  // the rrl class guards the oscillation logic with iteration > 0 (:387)
      if (!finest_level_ && (oscillation_count_ > 1 || (std::fabs(diff) < 0.01))) converged = true;
      else if (!converged) {
        if (weighted_error_after > 1.5) {
          if (curr_error_difference * error_difference_ < 0.0) ++oscillation_count_;
          else if (oscillation_count_ > 0) --oscillation_count_;
          if (oscillated()) converged = true;
        } else {
          if (curr_error_difference > 0.0) ++oscillation_count_;
          if (oscillation_count_ > 3) converged = true;
        }
      }
    }
    error_difference_ = curr_error_difference;
    if (converged && !finest_level_) {
      converged = false;
      number_matches_ *= 2;
      finest_level_ = grow(number_matches_);
      oscillation_count_ = 0;
      error_difference_ = 0.0;
    }
    return converged && finest_level_;
  }
};

template <unsigned dof, class FeatureT, bool RRL> class icp_symmetric;

// =============================================================================================
// forward estimator
// =============================================================================================
template <unsigned dof, class FeatureT, bool RRL>
class icp_direction : public cdcl_estimation_abs<3, dof, FeatureT> {
 public:
  static const unsigned dim = 3;
  typedef cdcl_estimation_abs<3, dof, FeatureT> superclass_type;
  typedef FeatureT feature_type;
  typedef typename superclass_type::feature_sptr_type feature_sptr_type;
  typedef typename superclass_type::match_type match_type;
  typedef typename superclass_type::match_sptr_type match_sptr_type;
  typedef typename superclass_type::trans_sptr_type trans_sptr_type;
  typedef typename superclass_type::display_callback_type display_callback_type;
  typedef vnl_vector_fixed<double, 3> vec3;
  friend class icp_symmetric<dof, FeatureT, RRL>;

  // (cdcl .txx:73-91; rrl .txx:89-113)
  icp_direction(vcl_vector<feature_sptr_type> const &moving, vcl_vector<feature_sptr_type> const &fixed,
                trans_sptr_type const &trans)
      : trans_(trans), res_(0), number_points_(1000), normalize_matches_(false), weight_by_strength_(false),
        scale_estimated_(false) {
    moving_.push_back(moving);
    fixed_.push_back(fixed);
    init_state();
    this->build_kd_trees();
    upload_moving();
    if (number_points_ > moving_[res_].size()) number_points_ = (unsigned)moving_[res_].size();
  }
  virtual ~icp_direction() { if (dir_) lrr_icp_dir_destroy(dir_); }

  // ---- reference public interface (cdcl/cdcl_estimation_ICP_matching_all.h:73-135)
  virtual bool run(void *caller = 0, display_callback_type display_points = 0) {
    this->initialize();
    bool converged = false;
    unsigned int max_iterations = 30, iteration = 0;
    while (!converged && iteration < max_iterations) {
      converged = this->one_iteration(caller, display_points, iteration);
      ++iteration;
    }
    iterations_run_ = iteration;
    return converged;
  }
  virtual void initialize() {}

  // (cdcl .txx:434-509; rrl .txx:427-462)
  virtual bool one_iteration(void *caller, display_callback_type display_points, unsigned int iteration) {
    this->find_closest_euclidean();
    if (caller && display_points) display_points(caller, moving_[res_], fixed_[res_], this->matches(), trans_, iteration);
    // RMS_before is needed only for the test at the end: its kernel is queued here, in the reference's place, and its
    // sums are collected after the estimate (the device runs the two back to back, the host waits once)
    lrr_icp_xform xf_before = xform_of(trans_);
    check(lrr_icp_errors_launch(dir_, &xf_before), "error metrics");
    trans_sptr_type global_transform;
    const bool rigid = is_rigid();
    if (rigid) global_transform = begin_rigid_increment();
    this->estimate_LS(true);
    if (rigid) end_rigid_increment(global_transform);
    double RMS_before = 0.0;
    {
      double s2 = 0, swr = 0, sw = 0;
      check(lrr_icp_errors_collect(dir_, &s2, &swr, &sw), "error metrics");
      int64_t count = lrr_icp_dir_num_matches(dir_);
      RMS_before = RRL ? s2 : (count != 0 ? std::sqrt(s2 / double(count)) : s2);
    }
    // NB rigid: from_ are still the pre-mapped copies, so they are mapped twice here, as in
    // the reference (cdcl .txx:499 against :477)
    double RMS_after = this->RMS_error();
    return std::fabs(RMS_before - RMS_after) < 1e-4;
  }

  // (cdcl .txx:139-153)
  void set_transform(trans_sptr_type const &trans) {
    trans_ = trans;
    scale_estimated_ = false;
    number_points_ = 1000;
    if (number_points_ > moving_[res_].size()) number_points_ = (unsigned)moving_[res_].size();
  }
  trans_sptr_type get_transform() { return trans_; }
  void normalize_matches() { normalize_matches_ = true; }
  // (cdcl .h:96)
  void set_moving_data(vcl_vector<vcl_vector<feature_sptr_type> > const &moving) {
    moving_ = moving;
    upload_moving();
    invalidate_matches();
  }

  // (cdcl .txx:196-230; the rrl class never increments `count` and returns the raw sum, rrl .txx:211-241)
  double RMS_error() {
    double s2 = 0, swr = 0, sw = 0;
    errors(s2, swr, sw);
    if (RRL) return s2;
    int64_t count = lrr_icp_dir_num_matches(dir_);
    return count != 0 ? std::sqrt(s2 / double(count)) : s2;
  }
  // (cdcl .txx:159-190; rrl .txx:177-203)
  double median_error() {
    lrr_icp_xform xf = xform_of(trans_);
    double med = 0.0;
    check(lrr_icp_median_error(dir_, &xf, &med), "median error");
    return med;
  }
  // (cdcl .txx:236-268)
  double weighted_error() {
    double s2 = 0, swr = 0, sw = 0;
    errors(s2, swr, sw);
    return sw != 0 ? swr / sw : swr;
  }
  // (cdcl .txx:388-404)
  void compute_transfer_error_covariance() {
    vnl_matrix_fixed<double, dof, dof> covar = trans_->get_covariance();
    double c[3] = {trans_->center_moving_[0], trans_->center_moving_[1], trans_->center_moving_[2]}, sum[9];
    check(lrr_icp_transfer_covariance(dir_, (int)dof, c, covar.data_block(), sum), "transfer error covariance");
    vnl_matrix_fixed<double, 3, 3> covarJ(0.0);
    for (int r = 0; r < 3; ++r) for (int e = 0; e < 3; ++e) covarJ(r, e) = sum[3 * r + e];
    covarJ *= (1.0 / double(lrr_icp_dir_num_matches(dir_)));
    trans_->set_covarianceJ(covarJ);
  }
  double sheet_angles() { return angles(false); }
  double tube_angles() { return angles(true); }
  void weight_by_strength(bool flag) { weight_by_strength_ = flag; }
  // (cdcl .txx:826-862; rrl .txx:722-752)
  void weight_by_strength() {
    double gs = 0;
    check(lrr_icp_weight_by_strength(dir_, nullptr, &gs), "weight_by_strength");
    touch();
  }
  // (cdcl .txx:973-1004; rrl .txx:756-778)
  void weight_spatially() {
    double ss = 0;
    check(lrr_icp_weight_spatially(dir_, &norm_, nullptr, nullptr, &ss), "weight_spatially");
    touch();
  }
  // (cdcl .txx:113-135): the search structure of the fixed set (here: upload + uniform grid)
  void build_kd_trees() {
    fixed_cloud_.upload(fixed_[res_]);
    rebuild_dir();
  }

  // ---- protected in the reference, reachable by the friend symmetric classes -----------------
  // (cdcl .txx:541-654; rrl .txx:494-571)
  void find_closest_euclidean() {
    lrr_icp_xform xf = xform_of(trans_);
    uint32_t keys[4];
    lrr_icp_perm_keys(&rng_, keys);
    const int64_t *order = explicit_order_.empty() ? nullptr : explicit_order_.data();
    check(lrr_icp_match(dir_, &xf, keys, order, (int64_t)number_points_, nn_method_), "find_closest_euclidean");
    matched();
  }
  // the symmetric classes' pair of searches, submitted together (same results as one after the other)
  static void find_closest_euclidean_pair(icp_direction &f, icp_direction &b) {
    lrr_icp_xform xf_f = xform_of(f.trans_), xf_b = xform_of(b.trans_);
    uint32_t keys_f[4], keys_b[4];
    lrr_icp_perm_keys(&f.rng_, keys_f);
    lrr_icp_perm_keys(&b.rng_, keys_b);
    const int64_t *order_f = f.explicit_order_.empty() ? nullptr : f.explicit_order_.data();
    const int64_t *order_b = b.explicit_order_.empty() ? nullptr : b.explicit_order_.data();
    int method = f.nn_method_ == b.nn_method_ ? f.nn_method_ : -1;
    if (method < 0) { f.find_closest_euclidean(); b.find_closest_euclidean(); return; }
    check(lrr_icp_match_pair(f.dir_, &xf_f, keys_f, order_f, (int64_t)f.number_points_, b.dir_, &xf_b, keys_b, order_b,
                             (int64_t)b.number_points_, method), "find_closest_euclidean");
    f.matched();
    b.matched();
  }
  void matched() {
    explicit_order_.clear();
    norm_ = identity_norm();
    touch();
    premapped_ = false;
    premap_G_ = trans_sptr_type();
  }
  // (cdcl .txx:660-822; rrl .txx:577-718)
  double estimate_scale_and_assign_weight() {
    lrr_icp_xform xf = xform_of(trans_);
    double S = 0, W = 0;
    check(lrr_icp_residual_pass(dir_, &xf, &norm_, &S, &W), "residual pass");
    double scale;
    if (!scale_estimated_) {
      if (injected_scale_ > 0.0) scale = injected_scale_;
      else check(lrr_icp_muse_scale(dir_, &scale), "MUSE scale");
      scale_estimated_ = true;
    } else {
      scale = std::sqrt(S / W);
    }
    check(lrr_icp_assign_weights(dir_, scale), "assign weights");
    last_scale_ = scale;
    touch();
    return scale;
  }
  // (cdcl .txx:1009-1093; rrl .txx:783-859)
  // The steps below in one device submission (lrr_icp_estimate_forward): statistics of the matches
  // and trans_->normalize, estimate_scale_and_assign_weight, weight_by_strength, AtA / Atb; the host
  // then normalises its transform object with the returned statistics, solves and un-normalises.
  void estimate_LS(bool estimate_parameters = false) {
    lrr_icp_fwd_opts o;
    o.dof = (int)dof;
    o.normalize = normalize_matches_ ? 1 : 0;
    o.strength = weight_by_strength_ ? 1 : 0;
    o.rigid_transform = is_rigid_transform() ? 1 : 0;  // (cdcl_trans_rigid3d::normalize keeps its centre, whichever class calls it)
    o.rigid_increment = is_rigid() ? 1 : 0;
    o.injected = 0.0;
    if (scale_estimated_) o.scale_mode = LRR_ICP_SCALE_RMS;
    else if (injected_scale_ > 0.0) { o.scale_mode = LRR_ICP_SCALE_INJECTED; o.injected = injected_scale_; }
    else o.scale_mode = LRR_ICP_SCALE_MUSE;
    // centre of the transform as the accumulation sees it: the affine normalize() zeroes it
    // (cdcl_trans_affine.txx:252), the rigid one keeps it (cdcl_trans_rigid3d.cxx:260-280)
    for (int d = 0; d < 3; ++d) o.jac_centre[d] = (normalize_matches_ && !is_rigid_transform()) ? 0.0 : trans_->center_moving_[d];
    lrr_icp_xform xf = xform_of(trans_);
    lrr_icp_sym_out r;
    check(lrr_icp_estimate_forward(dir_, &xf, &o, &r), "fused forward estimate");
    scale_estimated_ = true;
    last_scale_ = r.sigma_f;
    touch();
    norm_ = identity_norm();
    // affine transform, normalised, parameters estimated: normalize() -> solve -> unnormalize() without the dense
    // products whose results the solution overwrites (affine_forward_update: same operations, same bits)
    for (unsigned i = 0; i < dof * dof; ++i) last_AtA_[i] = r.AtA[i];
    for (unsigned i = 0; i < dof; ++i) last_Atb_[i] = r.Atb[i];
    if (normalize_matches_ && estimate_parameters && affine_shortcut<dof>::apply(trans_, r)) return;
    vec3 cm(0.0), cf(0.0);
    double rm = 0.0, rf = 0.0;
    if (normalize_matches_) {
      for (int d = 0; d < 3; ++d) { cm[d] = r.norm.cm[d]; cf[d] = r.norm.cf[d]; }
      rm = r.norm.rm;
      rf = r.norm.rf;
      trans_->normalize(rm, rf, cm, cf);
    }
    solve_and_set(trans_, r.AtA, r.Atb, estimate_parameters);
    if (normalize_matches_) trans_->unnormalize(rm, rf, cm, cf);
  }
  // the same estimate step by step through the granular C-ABI calls (each with its own host round trip)
  void estimate_LS_stepwise(bool estimate_parameters = false) {
    vec3 cm(0.0), cf(0.0);
    double rm = 0.0, rf = 0.0;
    if (normalize_matches_) {
      this->cdcl_normalize_matches(cm, rm, cf, rf);
      trans_->normalize(rm, rf, cm, cf);
    }
    this->estimate_scale_and_assign_weight();
    if (weight_by_strength_) this->weight_by_strength();
    double AtA[dof * dof], Atb[dof];
    accumulate(trans_, (!RRL && is_rigid()) ? LRR_ICP_ACC_RIGID_INCREMENT : 0, AtA, Atb);
    solve_and_set(trans_, AtA, Atb, estimate_parameters);
    if (normalize_matches_) {
      trans_->unnormalize(rm, rf, cm, cf);
      this->cdcl_unnormalize_matches(cm, rm, cf, rf);
    }
  }
  // (cdcl .txx:1159-1240): statistics only; the kernels normalise on the fly from now on
  void cdcl_normalize_matches(vec3 &center_moving, double &avg_radius_moving, vec3 &center_fixed, double &avg_radius_fixed) {
    lrr_icp_norm nm;
    check(lrr_icp_normalize_stats(dir_, &nm), "normalize statistics");
    for (int d = 0; d < 3; ++d) { center_moving[d] = nm.cm[d]; center_fixed[d] = nm.cf[d]; }
    avg_radius_moving = nm.rm;
    avg_radius_fixed = nm.rf;
    norm_ = nm;
  }
  // (cdcl sym .txx:869-908; rrl sym .txx:1093-1129)
  void cdcl_normalize_matches_known(vec3 const &center_moving, double avg_radius_moving, vec3 const &center_fixed, double avg_radius_fixed) {
    for (int d = 0; d < 3; ++d) { norm_.cm[d] = center_moving[d]; norm_.cf[d] = center_fixed[d]; }
    norm_.rm = avg_radius_moving;
    norm_.rf = avg_radius_fixed;
  }
  // (cdcl .txx:1245-1286)
  void cdcl_unnormalize_matches(vec3 const &, double, vec3 const &, double) { norm_ = identity_norm(); }
  // (cdcl .txx:94-108)
  bool set_number_points(unsigned int number_points) {
    if (number_points >= moving_[res_].size()) {
      number_points_ = (unsigned)moving_[res_].size();
      return true;
    }
    number_points_ = number_points;
    return false;
  }
  // matches_ of the reference, materialised from the device on demand
  vcl_vector<match_sptr_type> const &matches() {
    if (matches_dirty_) {
      int64_t n = lrr_icp_dir_num_matches(dir_);
      std::vector<int64_t> fi((size_t)n + 1), ti((size_t)n + 1);
      std::vector<double> w((size_t)n + 1);
      check(lrr_icp_get_matches(dir_, fi.data(), ti.data(), w.data()), "get matches");
      matches_.clear();
      matches_.reserve((size_t)n);
      for (int64_t i = 0; i < n; ++i) {
        match_sptr_type m = new match_type;
        feature_sptr_type p = moving_[res_][(size_t)fi[(size_t)i]];
        if (premapped_) {  // new feature at G(p) (cdcl .txx:473-479)
          feature_sptr_type c = new feature_type(*p);
          c->location_ = premap_G_->map_loc(p->location_);
          p = c;
        }
        m->from_ = p;
        m->to_.push_back(fixed_[res_][(size_t)ti[(size_t)i]]);
        m->w_.push_back(w[(size_t)i]);
        matches_.push_back(m);
      }
      matches_dirty_ = false;
    }
    return matches_;
  }

  // ---- hooks that do not exist in the reference (tests, benchmarks) --------------------------
  void set_subset_seed(uint64_t seed) { rng_ = seed; }
  void inject_initial_scale(double s) { injected_scale_ = s; }
  void set_subset_order(std::vector<int64_t> const &order) { explicit_order_ = order; }
  void set_nn_method(int method) { nn_method_ = method; }
  double last_scale() const { return last_scale_; }
  unsigned number_points() const { return number_points_; }
  unsigned iterations_run() const { return iterations_run_; }
  lrr_icp_dir *device_dir() const { return dir_; }
  void last_normal_equations(double *AtA, double *Atb) const {
    for (unsigned i = 0; i < dof * dof; ++i) AtA[i] = last_AtA_[i];
    for (unsigned i = 0; i < dof; ++i) Atb[i] = last_Atb_[i];
  }

 protected:
  static lrr_icp_norm identity_norm() {
    lrr_icp_norm n;
    for (int d = 0; d < 3; ++d) n.cm[d] = n.cf[d] = 0.0;
    n.rm = n.rf = 1.0;
    return n;
  }
  static lrr_icp_xform xform_of(trans_sptr_type const &T) {
    lrr_icp_xform x;
    vnl_matrix_fixed<double, 3, 3> &A = T->get_A();
    vnl_vector_fixed<double, 3> &t = T->get_translation();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) x.A[3 * r + c] = A(r, c); x.t[r] = t[r]; x.c[r] = T->center_moving_[r]; }
    return x;
  }
  // the rigid INCREMENT (pre-mapped from_ copies, recompose_increment) is the cdcl classes' (cdcl .txx:434-509); the rrl classes
  // estimate a cdcl_trans_rigid3d like any other transform
  bool is_rigid() const { return !RRL && is_rigid_transform(); }
  bool is_rigid_transform() const { return dof == 6 && trans_->is_type(cdcl_trans_rigid3d::type_id()); }
  void init_state() {
    dir_ = nullptr;
    rng_ = 1;
    injected_scale_ = -1.0;
    last_scale_ = 0.0;
    nn_method_ = LRR_ICP_NN_AUTO;
    norm_ = identity_norm();
    matches_dirty_ = true;
    premapped_ = false;
    iterations_run_ = 0;
    state_version_ = 1;
    err_version_ = 0;
    err_queued_version_ = 0;
    for (unsigned i = 0; i < dof * dof; ++i) last_AtA_[i] = 0.0;
    for (unsigned i = 0; i < dof; ++i) last_Atb_[i] = 0.0;
  }
  void upload_moving() {
    moving_cloud_.upload(moving_[res_]);
    rebuild_dir();
  }
  void rebuild_dir() {
    if (!moving_cloud_.handle() || !fixed_cloud_.handle()) return;
    if (dir_) lrr_icp_dir_destroy(dir_);
    dir_ = nullptr;
    check(lrr_icp_dir_create(device::ctx(), moving_cloud_.handle(), fixed_cloud_.handle(), &dir_), "dir create");
    invalidate_matches();
  }
  void invalidate_matches() {
    matches_.clear();
    touch();
    premapped_ = false;
    norm_ = identity_norm();
  }
  // RMS_error / weighted_error share one device pass; its three sums are kept until the
  // matches, the weights or the transform values change (the symmetric driver asks for the same
  // weighted error at the end of one iteration and at the start of the next)
  void errors(double &s2, double &swr, double &sw) {
    lrr_icp_xform xf = xform_of(trans_);
    if (err_version_ == state_version_ && std::memcmp(&xf, &err_xf_, sizeof xf) == 0) {
      s2 = err_sums_[0]; swr = err_sums_[1]; sw = err_sums_[2];
      return;
    }
    if (err_queued_version_ == state_version_ && std::memcmp(&xf, &err_queued_xf_, sizeof xf) == 0) {
      check(lrr_icp_errors_collect(dir_, &s2, &swr, &sw), "error metrics");  // queued by prefetch_errors()
    } else {
      check(lrr_icp_errors(dir_, &xf, &s2, &swr, &sw), "error metrics");
    }
    err_queued_version_ = 0;
    err_version_ = state_version_;
    err_xf_ = xf;
    err_sums_[0] = s2; err_sums_[1] = swr; err_sums_[2] = sw;
  }
  // queues the error pass for the current transform and state without waiting for it: errors() collects it if nothing
  // has changed by then (the symmetric estimate queues it before it inverts the new forward transform on the host)
  void prefetch_errors() {
    lrr_icp_xform xf = xform_of(trans_);
    if (err_version_ == state_version_ && std::memcmp(&xf, &err_xf_, sizeof xf) == 0) return;
    check(lrr_icp_errors_launch(dir_, &xf), "error metrics");
    err_queued_version_ = state_version_;
    err_queued_xf_ = xf;
  }
  void touch() { ++state_version_; matches_dirty_ = true; }
  void accumulate(trans_sptr_type const &jac_trans, int flags, double *AtA, double *Atb) {
    double c[3] = {jac_trans->center_moving_[0], jac_trans->center_moving_[1], jac_trans->center_moving_[2]};
    check(lrr_icp_accumulate(dir_, (int)dof, c, &norm_, flags, AtA, Atb), "normal equations");
  }
  void solve_and_set(trans_sptr_type const &T, const double *AtA, const double *Atb, bool estimate_parameters) {
    for (unsigned i = 0; i < dof * dof; ++i) last_AtA_[i] = AtA[i];
    for (unsigned i = 0; i < dof; ++i) last_Atb_[i] = Atb[i];
    solve_normal_equations<dof>(T, AtA, Atb, estimate_parameters);
  }
  // rigid increment bracket (cdcl .txx:455-484 and :490-496)
  trans_sptr_type begin_rigid_increment() {
    trans_sptr_type global_transform = trans_;
    trans_ = trans_->create_new();
    trans_->set_covariance(global_transform->get_covariance());
    lrr_icp_xform G = xform_of(global_transform);
    check(lrr_icp_premap(dir_, &G), "rigid pre-map");
    if (premapped_ && premap_G_) {
      // copies that are pre-mapped again (estimate_LS_backward after an iteration): their covariance carries both
      // rotations; only the rotation part of premap_G_ is used later (the angle diagnostics)
      trans_sptr_type both = global_transform->clone();
      both->get_A() = global_transform->get_A() * premap_G_->get_A();
      premap_G_ = both;
    } else {
      premap_G_ = global_transform->clone();
    }
    premapped_ = true;
    touch();
    trans_->set_parameterization(vnl_vector<double>(6, 0.0));
    return global_transform;
  }
  void end_rigid_increment(trans_sptr_type const &global_transform) {
    global_transform->recompose_increment(trans_);
    global_transform->set_covariance(trans_->get_covariance());
    trans_ = global_transform;
  }
  // sheet_angles / tube_angles.  cdcl_feature_ICP: mapped directions_ (rrl .txx:281-375).  cdcl_feature_with_shape:
  // eigenvectors of the covariances, the from-feature's rotated when it is a pre-mapped copy (cdcl .txx:273-382, 473-479).
  double angles(bool tube) {
    double sum_wa = 0.0, sum_w = 0.0;
    if (feature_traits<FeatureT>::covariance_angles) {
      double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      if (premapped_ && premap_G_) {
        vnl_matrix_fixed<double, 3, 3> &A = premap_G_->get_A();
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = A(r, c);
      }
      check(lrr_icp_angles_unmapped(dir_, R, tube ? 1 : 0, &sum_wa, &sum_w), "angle diagnostics");
    } else {
      lrr_icp_xform xf = xform_of(trans_);
      check(lrr_icp_angles(dir_, &xf, tube ? 1 : 0, &sum_wa, &sum_w), "angle diagnostics");
    }
    return sum_wa / sum_w;
  }

  vcl_vector<vcl_vector<feature_sptr_type> > moving_, fixed_;
  vcl_vector<match_sptr_type> matches_;
  trans_sptr_type trans_;
  unsigned int res_;
  unsigned int number_points_;
  bool normalize_matches_, weight_by_strength_, scale_estimated_;

  device_cloud<FeatureT> moving_cloud_, fixed_cloud_;
  lrr_icp_dir *dir_;
  lrr_icp_norm norm_;
  uint64_t rng_;
  double injected_scale_, last_scale_;
  int nn_method_;
  bool matches_dirty_, premapped_;
  trans_sptr_type premap_G_;
  std::vector<int64_t> explicit_order_;
  unsigned iterations_run_;
  double last_AtA_[dof * dof], last_Atb_[dof];
  uint64_t state_version_, err_version_, err_queued_version_;
  lrr_icp_xform err_xf_, err_queued_xf_;
  double err_sums_[3];
};

// =============================================================================================
// symmetric estimator
// =============================================================================================
template <unsigned dof, class FeatureT, bool RRL>
class icp_symmetric : public cdcl_estimation_abs<3, dof, FeatureT> {
 public:
  static const unsigned dim = 3;
  typedef icp_direction<dof, FeatureT, RRL> estimation_type;
  typedef cdcl_estimation_abs<3, dof, FeatureT> superclass_type;
  typedef FeatureT feature_type;
  typedef typename superclass_type::feature_sptr_type feature_sptr_type;
  typedef typename superclass_type::match_sptr_type match_sptr_type;
  typedef typename superclass_type::trans_sptr_type trans_sptr_type;
  typedef typename superclass_type::display_callback_type display_callback_type;
  typedef vnl_vector_fixed<double, 3> location_type;

  // (rrl sym .txx:198-220; cdcl sym .txx:144-164)
  icp_symmetric(vcl_vector<feature_sptr_type> const &moving, vcl_vector<feature_sptr_type> const &fixed,
                trans_sptr_type const &trans, trans_sptr_type const &trans_inverse,
                vcl_vector<feature_sptr_type> const &moving_inside, vcl_vector<feature_sptr_type> const &fixed_inside)
      : normalize_matches_(true) {
    forward_ = new estimation_type(moving_inside, fixed, trans);
    backward_ = new estimation_type(fixed_inside, moving, trans_inverse);
    forward_->weight_by_strength(true);
    forward_->normalize_matches();
    backward_->weight_by_strength(true);
    backward_->normalize_matches();
    set_subset_seed(1);
  }
  // region form (rrl sym .txx:135-192): ROI copies cut out of the full sets
  icp_symmetric(vcl_vector<feature_sptr_type> const &moving, vcl_vector<feature_sptr_type> const &fixed,
                trans_sptr_type const &trans, trans_sptr_type const &trans_inverse, location_type const &moving_x0,
                location_type const &moving_x1, location_type const &fixed_x0, location_type const &fixed_x1)
      : normalize_matches_(true) {
    vcl_vector<feature_sptr_type> fixed_inside = inside(fixed, fixed_x0, fixed_x1);
    vcl_vector<feature_sptr_type> moving_inside = inside(moving, moving_x0, moving_x1);
    forward_ = new estimation_type(moving_inside, fixed, trans);
    backward_ = new estimation_type(fixed_inside, moving, trans_inverse);
    forward_->weight_by_strength(true);
    forward_->normalize_matches();
    backward_->weight_by_strength(true);
    backward_->normalize_matches();
    set_subset_seed(1);
  }

  // (rrl sym .txx:224-264)
  bool set_regions(vcl_vector<feature_sptr_type> const &moving, vcl_vector<feature_sptr_type> const &fixed,
                   location_type const &moving_x0, location_type const &moving_x1, location_type const &fixed_x0,
                   location_type const &fixed_x1) {
    vcl_vector<feature_sptr_type> fixed_inside = inside(fixed, fixed_x0, fixed_x1);
    vcl_vector<feature_sptr_type> moving_inside = inside(moving, moving_x0, moving_x1);
    this->set_features(moving_inside, fixed_inside);
    return !(moving_inside.size() < 1000 || fixed_inside.size() < 1000);
  }
  // (rrl sym .txx:268-289)
  void set_features(vcl_vector<feature_sptr_type> const &moving_inside, vcl_vector<feature_sptr_type> const &fixed_inside) {
    vcl_vector<vcl_vector<feature_sptr_type> > mm(1, moving_inside), ff(1, fixed_inside);
    forward_->set_moving_data(mm);
    backward_->set_moving_data(ff);
    reset_schedule();
  }
  // (rrl sym .txx:293-299) The reference hands its two precomputed 1 mm Voronoi index volumes over here.  The images
  // themselves are not read: the call switches both directions to the quantised lookup SEMANTICS of the volumes
  // (LRR_ICP_NN_VORONOI, include/lrr_icp.h: the device rebuilds the equivalent of the volume from the feature set), so the
  // match sets are the production binary's.  voronoi_lookup::enable(false) / LRR_ICP_VORONOI=0 keeps the exact search.
  template <class VoronoiPointer> void set_voronoi_maps(VoronoiPointer const &, VoronoiPointer const &) {
    if (voronoi_lookup::enabled()) set_nn_method(LRR_ICP_NN_VORONOI);
    else voronoi_lookup::note_exact();
  }
  // (rrl sym .txx:303-315)
  void set_transform(trans_sptr_type const &trans) {
    forward_->set_transform(trans);
    backward_->set_transform(trans->inverse());
    reset_schedule();
  }
  virtual void initialize() { forward_->initialize(); backward_->initialize(); }
  virtual bool run(void * = 0, display_callback_type = 0) { return false; }  // stub in the reference (rrl sym .h)
  void find_closest_euclidean() { estimation_type::find_closest_euclidean_pair(*forward_, *backward_); }

  // (rrl sym .txx:341-438; cdcl sym .txx:268-363)
  virtual bool one_iteration(void *caller, display_callback_type display_points, unsigned int iteration) {
    double weighted_error_before = forward_->weighted_error();
    this->find_closest_euclidean();
    this->estimate_LS(true);
    if (caller && display_points)
      display_points(caller, forward_->moving_[forward_->res_], forward_->fixed_[forward_->res_], forward_->matches(), forward_->trans_, iteration);
    double weighted_error_after = forward_->weighted_error();
    return sched_.step(weighted_error_before, weighted_error_after, iteration, RRL, [this](unsigned n) {
      bool forward_at_max = forward_->set_number_points(n);
      bool backward_at_max = backward_->set_number_points(n);
      return forward_at_max && backward_at_max;
    });
  }

  // (rrl sym .txx:699-918; cdcl sym .txx:398-701).  The per-match work of both lists is ONE fused,
  // sync-free device submission (lrr_icp_estimate_symmetric); the host applies the returned
  // statistics to the transform objects, solves the dof x dof system and un-normalises.
  void estimate_LS(bool estimate_parameters) {
    estimation_type &F = *forward_, &B = *backward_;
    trans_sptr_type forward_global, backward_global;
    const bool rigid = !RRL && F.is_rigid();
    if (rigid) {
      forward_global = F.begin_rigid_increment();
      if (B.is_rigid()) backward_global = B.begin_rigid_increment();
    }
    lrr_icp_sym_opts o;
    o.dof = (int)dof;
    o.rrl = RRL ? 1 : 0;
    o.normalize = normalize_matches_ ? 1 : 0;
    o.strength_f = F.weight_by_strength_ ? 1 : 0;
    o.strength_b = B.weight_by_strength_ ? 1 : 0;
    o.rigid_increment = rigid ? 1 : 0;
    o.scale_mode_f = scale_mode(F, o.injected_f);
    o.scale_mode_b = scale_mode(B, o.injected_b);
    // centre of the forward transform as the accumulation sees it: the affine normalize() zeroes
    // it (cdcl_trans_affine.txx:252), the rigid one keeps it (cdcl_trans_rigid3d.cxx:260-280)
    for (int d = 0; d < 3; ++d) o.jac_centre[d] = (normalize_matches_ && dof == 12) ? 0.0 : F.trans_->center_moving_[d];
    lrr_icp_xform xf = estimation_type::xform_of(F.trans_), xb = estimation_type::xform_of(B.trans_);
    lrr_icp_sym_out r;
    check(lrr_icp_estimate_symmetric(F.dir_, B.dir_, &xf, &xb, &o, &r), "fused symmetric estimate");
    F.scale_estimated_ = B.scale_estimated_ = true;
    F.last_scale_ = r.sigma_f;
    B.last_scale_ = r.sigma_b;
    F.touch();
    B.touch();
    for (unsigned i = 0; i < dof * dof; ++i) F.last_AtA_[i] = r.AtA[i];
    for (unsigned i = 0; i < dof; ++i) F.last_Atb_[i] = r.Atb[i];
    symmetric_apply_solution<dof>(F.trans_, B.trans_, r, normalize_matches_, estimate_parameters, estimate_parameters);
    if (rigid) F.end_rigid_increment(forward_global);
    // the iteration asks for the forward list's weighted error next: its pass runs while the host inverts the transform
    if (estimate_parameters && !rigid) F.prefetch_errors();
    if (estimate_parameters) B.trans_ = F.trans_->inverse();
  }
  // rrl: swap forward_/backward_, estimate_LS, swap back (rrl sym .txx:921-934).
  // cdcl: the class's own hand-written variant (cdcl sym .txx:704-863), step by step through the granular C-ABI calls
  // (a diagnostic outside the iteration loop): only the backward list is pre-mapped (rigid); both lists get their scale
  // and Beaton-Tukey weights; the forward matches join the backward list as role-reversed clones BEFORE the
  // normalisation statistics are taken (every clone counts among the distinct to_); strength weighting of the merged
  // list, no spatial weighting; normal equations under the backward transform's Jacobian; covariance (and, when asked,
  // parameters) of the backward transform; forward = inverse(backward) — of the INCREMENT in the rigid case, before the
  // increment is recomposed, as the reference orders it (:843-858).
  void estimate_LS_backward(bool estimate_parameters) {
    if (RRL) {
      vbl_smart_ptr<estimation_type> f = forward_, b = backward_;
      forward_ = b;
      backward_ = f;
      this->estimate_LS(estimate_parameters);
      forward_ = f;
      backward_ = b;
      return;
    }
    estimation_type &F = *forward_, &B = *backward_;
    trans_sptr_type backward_global;
    const bool rigid = B.is_rigid();
    if (rigid) backward_global = B.begin_rigid_increment();
    F.estimate_scale_and_assign_weight();
    B.estimate_scale_and_assign_weight();
    check(lrr_icp_clone_weights(F.dir_), "add_constraints");
    lrr_icp_norm nm = estimation_type::identity_norm();
    location_type cm(0.0), cf(0.0);
    if (normalize_matches_) {
      check(lrr_icp_normalize_stats_merged(B.dir_, F.dir_, &nm), "merged normalize statistics");
      for (int d = 0; d < 3; ++d) { cm[d] = nm.cm[d]; cf[d] = nm.cf[d]; }
      B.trans_->normalize(nm.rm, nm.rf, cm, cf);
    }
    if (B.weight_by_strength_) {
      double gs = 0;
      check(lrr_icp_weight_by_strength(B.dir_, F.dir_, &gs), "weight_by_strength");
    }
    // the clones: from_ = the forward match's to_ (normalised like the merged from-side), to_ = its from_
    lrr_icp_norm flipped;
    for (int d = 0; d < 3; ++d) { flipped.cm[d] = nm.cf[d]; flipped.cf[d] = nm.cm[d]; }
    flipped.rm = nm.rf;
    flipped.rf = nm.rm;
    double c[3] = {B.trans_->center_moving_[0], B.trans_->center_moving_[1], B.trans_->center_moving_[2]};
    double AtA[dof * dof], Atb[dof], AtA2[dof * dof], Atb2[dof];
    const int inc = rigid ? LRR_ICP_ACC_RIGID_INCREMENT : 0;
    check(lrr_icp_accumulate(B.dir_, (int)dof, c, &nm, inc, AtA, Atb), "normal equations");
    check(lrr_icp_accumulate(F.dir_, (int)dof, c, &flipped, inc | LRR_ICP_ACC_REVERSED | LRR_ICP_ACC_CLONE_W, AtA2, Atb2), "normal equations");
    for (unsigned i = 0; i < dof * dof; ++i) AtA[i] += AtA2[i];
    for (unsigned i = 0; i < dof; ++i) Atb[i] += Atb2[i];
    B.solve_and_set(B.trans_, AtA, Atb, estimate_parameters);
    if (normalize_matches_) B.trans_->unnormalize(nm.rm, nm.rf, cm, cf);
    F.touch();
    B.touch();
    if (estimate_parameters) F.trans_ = B.trans_->inverse();
    if (rigid) B.end_rigid_increment(backward_global);
  }

  // stubs in the reference too ("to be implemented": cdcl sym .h:90-93, rrl sym .h:104-107)
  virtual bool estimate_parameters() { return false; }
  virtual bool estimate_covariance() { return false; }
  trans_sptr_type get_transform() { return forward_->get_transform(); }
  trans_sptr_type get_transform_backward() { return backward_->get_transform(); }
  void weight_by_strength(bool flag) { forward_->weight_by_strength(flag); backward_->weight_by_strength(flag); }
  virtual void normalize_matches() { forward_->normalize_matches(); backward_->normalize_matches(); normalize_matches_ = true; }
  double RMS_error() { return forward_->RMS_error(); }
  bool oscillated() { return sched_.oscillated(); }
  double weighted_error() { return forward_->weighted_error(); }
  void compute_transfer_error_covariance() { forward_->compute_transfer_error_covariance(); backward_->compute_transfer_error_covariance(); }
  double sheet_angles() { return forward_->sheet_angles(); }
  double tube_angles() { return forward_->tube_angles(); }
  double weighted_error_backward() { return backward_->weighted_error(); }
  double sheet_angles_backward() { return backward_->sheet_angles(); }
  double tube_angles_backward() { return backward_->tube_angles(); }

  // ---- hooks that do not exist in the reference ----------------------------------------------
  void set_subset_seed(uint64_t seed) { forward_->set_subset_seed(seed); backward_->set_subset_seed(seed ^ 0xD1B54A32D192ED03ULL); }
  void inject_initial_scale(double fwd, double bwd) { forward_->inject_initial_scale(fwd); backward_->inject_initial_scale(bwd); }
  void set_number_points(unsigned n) {
    bool a = forward_->set_number_points(n), b = backward_->set_number_points(n);
    sched_.number_matches_ = n;
    sched_.finest_level_ = a && b;
  }
  void set_nn_method(int m) { forward_->set_nn_method(m); backward_->set_nn_method(m); }
  bool finest_level() const { return sched_.finest_level_; }
  estimation_type &forward() { return *forward_; }
  estimation_type &backward() { return *backward_; }

 protected:
  static vcl_vector<feature_sptr_type> inside(vcl_vector<feature_sptr_type> const &all, location_type const &x0, location_type const &x1) {
    vcl_vector<feature_sptr_type> out;
    for (size_t i = 0; i < all.size(); ++i) {
      location_type const &p = all[i]->location_;
      bool in = true;
      for (unsigned d = 0; d < 3; ++d) if (x0[d] > p[d] || p[d] > x1[d]) in = false;
      if (in) out.push_back(new feature_type(*(all[i])));
    }
    return out;
  }
  // first estimate after set_transform: MUSE (or the injected test value); afterwards the RMS of
  // the residuals under the previous weights (rrl .txx:620-641)
  static int scale_mode(estimation_type &D, double &injected) {
    injected = 0.0;
    if (D.scale_estimated_) return LRR_ICP_SCALE_RMS;
    if (D.injected_scale_ > 0.0) { injected = D.injected_scale_; return LRR_ICP_SCALE_INJECTED; }
    return LRR_ICP_SCALE_MUSE;
  }
  void reset_schedule() { sched_.reset(); }

  vbl_smart_ptr<estimation_type> forward_, backward_;
  bool normalize_matches_;
  symmetric_schedule sched_;
};

}  // namespace lrr_amd
#endif
