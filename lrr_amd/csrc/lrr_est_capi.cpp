// lrr_est_capi.cpp — flat C view of the C++ drop-in estimator classes (include/lrr_amd/*.h).
//
// The product interface is the C++ class family; this file only exists so that Python
// (tests/, bench.py, the multi-GPU driver) can drive those classes through ctypes: it builds
// cdcl_feature_* objects from flat arrays, instantiates the four classes at <3,6> and <3,12>
// and forwards the reference's method names one to one.  No arithmetic of the hot path lives
// here.
#include <chrono>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "lrr_amd/cdcl_estimation_ICP_matching_all.h"
#include "lrr_amd/cdcl_estimation_symmetric_ICP_matching_all.h"
#include "lrr_amd/rrl_estimation_ICP_matching_all.h"
#include "lrr_amd/rrl_estimation_symmetric_ICP_matching_all.h"
#include "lrr_amd/location_batch.h"
#include "lrr_amd/cdcl_estimation.h"
#include "lrr_amd/vtp_features.h"

extern "C" {
typedef struct {
  int64_t n;
  const double *loc;       // n x 3
  const double *proj;      // n x 9
  const double *strength;  // n
  const uint8_t *shape;    // n (may be NULL)
  const double *dirs;      // n x 6 (may be NULL)
  const double *cov;       // n x 9 row-major, covariance_ of cdcl_feature_with_shape (may be NULL)
} lrr_est_features;
}

namespace {

typedef vnl_vector_fixed<double, 3> vec3;
typedef vnl_matrix_fixed<double, 3, 3> mat3;

thread_local std::string g_error;

template <class F>
std::vector<typename F::sptr> make_features(const lrr_est_features *f);

template <class F>
void fill_common(F *p, const lrr_est_features *f, int64_t i) {
  for (int d = 0; d < 3; ++d) p->location_[d] = f->loc[3 * i + d];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) p->error_projector_(r, c) = f->proj[9 * i + 3 * r + c];
  p->shape_ = (typename F::shape_type)(f->shape ? f->shape[i] : 0);
  if (f->dirs) {
    p->directions_.push_back(vec3(f->dirs[6 * i], f->dirs[6 * i + 1], f->dirs[6 * i + 2]));
    p->directions_.push_back(vec3(f->dirs[6 * i + 3], f->dirs[6 * i + 4], f->dirs[6 * i + 5]));
  }
}

template <>
std::vector<cdcl_feature_with_shape<3>::sptr> make_features<cdcl_feature_with_shape<3> >(const lrr_est_features *f) {
  std::vector<cdcl_feature_with_shape<3>::sptr> out;
  if (!f) return out;
  out.reserve((size_t)f->n);
  for (int64_t i = 0; i < f->n; ++i) {
    cdcl_feature_with_shape<3> *p = new cdcl_feature_with_shape<3>;
    fill_common(p, f, i);
    p->strength_ = f->strength ? f->strength[i] : 0.0;
    if (f->cov)
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) p->covariance_(r, c) = f->cov[9 * i + 3 * r + c];
    out.push_back(p);
  }
  return out;
}

template <>
std::vector<cdcl_feature_ICP<3>::sptr> make_features<cdcl_feature_ICP<3> >(const lrr_est_features *f) {
  std::vector<cdcl_feature_ICP<3>::sptr> out;
  if (!f) return out;
  out.reserve((size_t)f->n);
  for (int64_t i = 0; i < f->n; ++i) {
    cdcl_feature_ICP<3> *p = new cdcl_feature_ICP<3>;
    fill_common(p, f, i);
    p->strength_ = (float)(f->strength ? f->strength[i] : 0.0);
    out.push_back(p);
  }
  return out;
}

template <unsigned dof> struct trans_factory;
template <> struct trans_factory<6> {
  static cdcl_trans<3, 6>::sptr make(const double *A, const double *t, const double *c) {
    mat3 R; vec3 tt, cc;
    for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) R(r, k) = A[3 * r + k]; tt[r] = t[r]; cc[r] = c[r]; }
    return new cdcl_trans_rigid3d(R, tt, cc);
  }
};
template <> struct trans_factory<12> {
  static cdcl_trans<3, 12>::sptr make(const double *A, const double *t, const double *c) {
    mat3 M; vec3 tt, cc;
    for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) M(r, k) = A[3 * r + k]; tt[r] = t[r]; cc[r] = c[r]; }
    return new cdcl_trans_affine<3, 12>(M, tt, cc);
  }
};

template <unsigned dof>
void read_trans(typename cdcl_trans<3, dof>::sptr const &T, double *A, double *t, double *c, double *covar) {
  if (A) for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) A[3 * r + k] = T->get_A()(r, k);
  if (t) for (int r = 0; r < 3; ++r) t[r] = T->get_translation()[r];
  if (c) for (int r = 0; r < 3; ++r) c[r] = T->center_moving_[r];
  if (covar) for (unsigned r = 0; r < dof; ++r) for (unsigned k = 0; k < dof; ++k) covar[r * dof + k] = T->get_covariance()(r, k);
}

// validTransform (rrl/itkLocationRegistration.cxx:2914-2926): singular values of A within [0.1, 10]
template <unsigned dof>
bool singular_values_in_range(typename cdcl_trans<3, dof>::sptr const &T) {
  double U[9], sv[3], V[9];
  lrr_amd::linalg::svd3(T->get_A().data_block(), U, sv, V);
  for (int d = 0; d < 3; ++d) if (sv[d] > 10.0 || sv[d] < 0.1) return false;
  return true;
}

// set_voronoi_maps(moving_map, fixed_map) of the symmetric classes / set_fixed_voronoi_map(map) of the rrl forward class, called
// the way the reference's caller does (rrl/itkLocationRegistration.cxx hands the two index images over); classes without
// the method (the cdcl forward class) have nothing to call
template <class Est, class = void> struct voronoi_call {
  static void apply(Est &) { throw std::runtime_error("this class has no Voronoi-map setter"); }
};
template <class Est> struct voronoi_call<Est, decltype(std::declval<Est &>().set_voronoi_maps((const void *)0, (const void *)0))> {
  static void apply(Est &e) { e.set_voronoi_maps((const void *)0, (const void *)0); }
};
template <unsigned dim, unsigned dof> struct voronoi_call<rrl_estimation_ICP_matching_all<dim, dof>, void> {
  static void apply(rrl_estimation_ICP_matching_all<dim, dof> &e) { e.set_fixed_voronoi_map((const void *)0); }
};

struct est_base {
  int dof = 0;
  virtual ~est_base() {}
  virtual void set_transform(const double *A, const double *t, const double *c) = 0;
  virtual void set_features(const lrr_est_features *mi, const lrr_est_features *fi) = 0;
  virtual void weight_by_strength(bool) = 0;
  virtual void normalize_matches() = 0;
  virtual void set_number_points(int64_t n) = 0;
  virtual void inject_initial_scale(double f, double b) = 0;
  virtual void set_subset_order(int dir, const int64_t *order, int64_t n) = 0;
  virtual void set_nn_method(int m) = 0;
  virtual void set_voronoi_maps() = 0;  // the class's own call, with null images (the drop-in never reads them)
  virtual void set_subset_seed(uint64_t seed) = 0;
  virtual void find_closest_euclidean() = 0;
  virtual bool one_iteration(unsigned it) = 0;
  virtual bool run(unsigned max_it, unsigned *done, int64_t *pairs, bool check_valid = false, int *valid = nullptr) = 0;
  virtual void estimate_LS(bool) = 0;
  virtual void estimate_LS_backward(bool) = 0;
  virtual void get_transform(int dir, double *A, double *t, double *c, double *covar) = 0;
  virtual double RMS_error() = 0;
  virtual double median_error() = 0;
  virtual double weighted_error(int dir) = 0;
  virtual double sheet_angles(int dir) = 0;
  virtual double tube_angles(int dir) = 0;
  virtual void transfer_error_covariance(double *cf, double *cb) = 0;
  virtual bool oscillated() = 0;
  virtual bool finest_level() = 0;
  virtual int64_t num_matches(int dir) = 0;
  virtual void get_matches(int dir, int64_t *f, int64_t *t, double *w) = 0;
  virtual int64_t number_points(int dir) = 0;
  virtual double last_scale(int dir) = 0;
  virtual void normal_equations(double *AtA, double *Atb) = 0;
  virtual lrr_icp_dir *device_dir(int dir) = 0;
};

template <class Dir>
void dir_matches(Dir &d, int64_t *f, int64_t *t, double *w) {
  if (lrr_icp_get_matches(d.device_dir(), f, t, w) != LRR_ICP_OK) throw std::runtime_error("get_matches failed");
}

template <unsigned DOF, class F, class Est>
struct fwd_impl : est_base {
  vbl_smart_ptr<Est> e;
  fwd_impl(const lrr_est_features *m, const lrr_est_features *f, const double *A, const double *t, const double *c, uint64_t seed, int nn) {
    this->dof = DOF;
    e = new Est(make_features<F>(m), make_features<F>(f), trans_factory<DOF>::make(A, t, c));
    e->set_subset_seed(seed);
    e->set_nn_method(nn);
  }
  void set_transform(const double *A, const double *t, const double *c) override { e->set_transform(trans_factory<DOF>::make(A, t, c)); }
  void set_features(const lrr_est_features *mi, const lrr_est_features *) override {
    std::vector<std::vector<typename F::sptr> > mm(1, make_features<F>(mi));
    e->set_moving_data(mm);
  }
  void weight_by_strength(bool b) override { e->weight_by_strength(b); }
  void normalize_matches() override { e->normalize_matches(); }
  void set_number_points(int64_t n) override { e->set_number_points((unsigned)n); }
  void inject_initial_scale(double f, double) override { e->inject_initial_scale(f); }
  void set_subset_order(int, const int64_t *order, int64_t n) override { e->set_subset_order(std::vector<int64_t>(order, order + n)); }
  void set_nn_method(int m) override { e->set_nn_method(m); }
  void set_voronoi_maps() override { voronoi_call<Est>::apply(*e); }
  void set_subset_seed(uint64_t seed) override { e->set_subset_seed(seed); }
  void find_closest_euclidean() override { e->find_closest_euclidean(); }
  bool one_iteration(unsigned it) override { return e->one_iteration(0, 0, it); }
  bool run(unsigned max_it, unsigned *done, int64_t *pairs, bool check_valid, int *valid) override {
    // run() of the reference hard-codes 30 iterations (cdcl .txx:419)
    bool conv = false;
    unsigned it = 0;
    if (valid) *valid = 1;
    e->initialize();
    while (!conv && it < max_it) {
      conv = e->one_iteration(0, 0, it);
      ++it;
      if (pairs) *pairs += e->number_points();
      if (check_valid && !singular_values_in_range<DOF>(e->get_transform())) { if (valid) *valid = 0; break; }
    }
    if (done) *done = it;
    return conv;
  }
  void estimate_LS(bool b) override { e->estimate_LS(b); }
  void estimate_LS_backward(bool) override {}
  void get_transform(int, double *A, double *t, double *c, double *covar) override { read_trans<DOF>(e->get_transform(), A, t, c, covar); }
  double RMS_error() override { return e->RMS_error(); }
  double median_error() override { return e->median_error(); }
  double weighted_error(int) override { return e->weighted_error(); }
  double sheet_angles(int) override { return e->sheet_angles(); }
  double tube_angles(int) override { return e->tube_angles(); }
  void transfer_error_covariance(double *cf, double *) override {
    e->compute_transfer_error_covariance();
    if (cf) for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) cf[3 * r + k] = e->get_transform()->get_covarianceJ()(r, k);
  }
  bool oscillated() override { return false; }
  bool finest_level() override { return false; }
  int64_t num_matches(int) override { return lrr_icp_dir_num_matches(e->device_dir()); }
  void get_matches(int, int64_t *f, int64_t *t, double *w) override { dir_matches(*e, f, t, w); }
  int64_t number_points(int) override { return e->number_points(); }
  double last_scale(int) override { return e->last_scale(); }
  void normal_equations(double *AtA, double *Atb) override { e->last_normal_equations(AtA, Atb); }
  lrr_icp_dir *device_dir(int) override { return e->device_dir(); }
};

template <unsigned DOF, class F, class Est>
struct sym_impl : est_base {
  vbl_smart_ptr<Est> e;
  sym_impl(const lrr_est_features *m, const lrr_est_features *f, const lrr_est_features *mi, const lrr_est_features *fi,
           const double *A, const double *t, const double *c, uint64_t seed, int nn) {
    this->dof = DOF;
    typename cdcl_trans<3, DOF>::sptr T = trans_factory<DOF>::make(A, t, c);
    e = new Est(make_features<F>(m), make_features<F>(f), T, T->inverse(), make_features<F>(mi), make_features<F>(fi));
    e->set_subset_seed(seed);
    e->set_nn_method(nn);
  }
  typename Est::estimation_type &d(int dir) { return dir ? e->backward() : e->forward(); }
  void set_transform(const double *A, const double *t, const double *c) override { e->set_transform(trans_factory<DOF>::make(A, t, c)); }
  void set_features(const lrr_est_features *mi, const lrr_est_features *fi) override { e->set_features(make_features<F>(mi), make_features<F>(fi)); }
  void weight_by_strength(bool b) override { e->weight_by_strength(b); }
  void normalize_matches() override { e->normalize_matches(); }
  void set_number_points(int64_t n) override { e->set_number_points((unsigned)n); }
  void inject_initial_scale(double f, double b) override { e->inject_initial_scale(f, b); }
  void set_subset_order(int dir, const int64_t *order, int64_t n) override { d(dir).set_subset_order(std::vector<int64_t>(order, order + n)); }
  void set_nn_method(int m) override { e->set_nn_method(m); }
  void set_voronoi_maps() override { voronoi_call<Est>::apply(*e); }
  void set_subset_seed(uint64_t seed) override { e->set_subset_seed(seed); }
  void find_closest_euclidean() override { e->find_closest_euclidean(); }
  bool one_iteration(unsigned it) override { return e->one_iteration(0, 0, it); }
  bool run(unsigned max_it, unsigned *done, int64_t *pairs, bool check_valid, int *valid) override {
    // the caller's loop (rrl/itkLocationRegistration.cxx:2418-2482): the class's own run() is a stub
    bool conv = false;
    unsigned it = 0;
    if (valid) *valid = 1;
    while (!conv && it < max_it) {
      if (pairs) *pairs += e->forward().number_points() + e->backward().number_points();
      conv = e->one_iteration(0, 0, it);
      ++it;
      // `if( !validTransform( estimation->get_transform() ) ) break;` (:2482)
      if (check_valid && !singular_values_in_range<DOF>(e->get_transform())) { if (valid) *valid = 0; break; }
    }
    if (done) *done = it;
    return conv;
  }
  void estimate_LS(bool b) override { e->estimate_LS(b); }
  void estimate_LS_backward(bool b) override { e->estimate_LS_backward(b); }
  void get_transform(int dir, double *A, double *t, double *c, double *covar) override {
    read_trans<DOF>(dir ? e->get_transform_backward() : e->get_transform(), A, t, c, covar);
  }
  double RMS_error() override { return e->RMS_error(); }
  double median_error() override { return e->forward().median_error(); }  // (the symmetric classes have none: their forward list's)
  double weighted_error(int dir) override { return dir ? e->weighted_error_backward() : e->weighted_error(); }
  double sheet_angles(int dir) override { return dir ? e->sheet_angles_backward() : e->sheet_angles(); }
  double tube_angles(int dir) override { return dir ? e->tube_angles_backward() : e->tube_angles(); }
  void transfer_error_covariance(double *cf, double *cb) override {
    e->compute_transfer_error_covariance();
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) {
      if (cf) cf[3 * r + k] = e->get_transform()->get_covarianceJ()(r, k);
      if (cb) cb[3 * r + k] = e->get_transform_backward()->get_covarianceJ()(r, k);
    }
  }
  bool oscillated() override { return e->oscillated(); }
  bool finest_level() override { return e->finest_level(); }
  int64_t num_matches(int dir) override { return lrr_icp_dir_num_matches(d(dir).device_dir()); }
  void get_matches(int dir, int64_t *f, int64_t *t, double *w) override { dir_matches(d(dir), f, t, w); }
  int64_t number_points(int dir) override { return d(dir).number_points(); }
  double last_scale(int dir) override { return d(dir).last_scale(); }
  void normal_equations(double *AtA, double *Atb) override { e->forward().last_normal_equations(AtA, Atb); }
  lrr_icp_dir *device_dir(int dir) override { return d(dir).device_dir(); }
};

template <class Fn>
int guard(Fn fn) {
  try {
    fn();
    return 0;
  } catch (const std::exception &ex) {
    g_error = ex.what();
    return -1;
  } catch (...) {
    g_error = "unknown exception";
    return -1;
  }
}

}  // namespace

// info[8]: objective of the parameter estimate at the start / end, of the covariance estimate at the start / end, the two
// optimiser codes, the two evaluation counts (zeros where a step did not run); state[3] (may be NULL): resolution after the
// iteration, matches_.size() of its last compute_weights, 1 when that fell back to the closest points, its avg_largest_eval
template <class EM>
static void cdc_em_iterate(EM *em, unsigned iteration, int *converged, double *info, double *state) {
  bool cv = em->one_iteration(iteration);
  if (converged) *converged = cv ? 1 : 0;
  if (info) {
    const typename EM::step_info &s = em->last_step();
    const bool par = iteration > 5;
    info[0] = par ? s.par_start : 0.0; info[1] = par ? s.par_end : 0.0; info[2] = s.cov_start; info[3] = s.cov_end;
    info[4] = par ? s.par_code : 0; info[5] = s.cov_code; info[6] = par ? s.par_evals : 0; info[7] = s.cov_evals;
  }
  if (state) { state[0] = em->resolution(); state[1] = (double)em->matches_size(); state[2] = em->used_closest_euclidean() ? 1.0 : 0.0; state[3] = em->avg_largest_eval(); state[4] = em->switches_down(); state[5] = em->switches_up(); }
}

extern "C" {

typedef struct lrr_est lrr_est;

const char *lrr_est_last_error() { return g_error.c_str(); }

int lrr_est_set_device(int device) {
  return guard([&] { lrr_amd::device::set_device(device); (void)lrr_amd::device::ctx(); });
}

// context of the calling thread (for the timing helpers of lrr_icp.h)
lrr_icp_ctx *lrr_est_context() {
  lrr_icp_ctx *c = nullptr;
  guard([&] { c = lrr_amd::device::ctx(); });
  return c;
}

// kind: 0 cdcl forward, 1 rrl forward, 2 cdcl symmetric, 3 rrl symmetric; dof 6 or 12
lrr_est *lrr_est_create(int kind, int dof, const lrr_est_features *moving, const lrr_est_features *fixed,
                        const lrr_est_features *moving_inside, const lrr_est_features *fixed_inside, const double *A,
                        const double *t, const double *c, uint64_t seed, int nn_method) {
  est_base *out = nullptr;
  typedef cdcl_feature_with_shape<3> FS;
  typedef cdcl_feature_ICP<3> FI;
  int rc = guard([&] {
    if (!moving || !fixed || !A || !t || !c) throw std::runtime_error("lrr_est_create: null argument");
    if (kind >= 2 && (!moving_inside || !fixed_inside)) throw std::runtime_error("lrr_est_create: symmetric kinds need the inside sets");
    if (dof == 6) {
      if (kind == 0) out = new fwd_impl<6, FS, cdcl_estimation_ICP_matching_all<3, 6> >(moving, fixed, A, t, c, seed, nn_method);
      else if (kind == 1) out = new fwd_impl<6, FI, rrl_estimation_ICP_matching_all<3, 6> >(moving, fixed, A, t, c, seed, nn_method);
      else if (kind == 2) out = new sym_impl<6, FS, cdcl_estimation_symmetric_ICP_matching_all<3, 6> >(moving, fixed, moving_inside, fixed_inside, A, t, c, seed, nn_method);
      else if (kind == 3) out = new sym_impl<6, FI, rrl_estimation_symmetric_ICP_matching_all<3, 6> >(moving, fixed, moving_inside, fixed_inside, A, t, c, seed, nn_method);
    } else if (dof == 12) {
      if (kind == 0) out = new fwd_impl<12, FS, cdcl_estimation_ICP_matching_all<3, 12> >(moving, fixed, A, t, c, seed, nn_method);
      else if (kind == 1) out = new fwd_impl<12, FI, rrl_estimation_ICP_matching_all<3, 12> >(moving, fixed, A, t, c, seed, nn_method);
      else if (kind == 2) out = new sym_impl<12, FS, cdcl_estimation_symmetric_ICP_matching_all<3, 12> >(moving, fixed, moving_inside, fixed_inside, A, t, c, seed, nn_method);
      else if (kind == 3) out = new sym_impl<12, FI, rrl_estimation_symmetric_ICP_matching_all<3, 12> >(moving, fixed, moving_inside, fixed_inside, A, t, c, seed, nn_method);
    }
    if (!out) throw std::runtime_error("lrr_est_create: unsupported kind/dof");
  });
  if (rc) return nullptr;
  return reinterpret_cast<lrr_est *>(out);
}

#define E(e) (reinterpret_cast<est_base *>(e))

void lrr_est_destroy(lrr_est *e) { delete E(e); }
int lrr_est_set_transform(lrr_est *e, const double *A, const double *t, const double *c) { return guard([&] { E(e)->set_transform(A, t, c); }); }
int lrr_est_set_features(lrr_est *e, const lrr_est_features *mi, const lrr_est_features *fi) { return guard([&] { E(e)->set_features(mi, fi); }); }
int lrr_est_weight_by_strength(lrr_est *e, int flag) { return guard([&] { E(e)->weight_by_strength(flag != 0); }); }
int lrr_est_normalize_matches(lrr_est *e) { return guard([&] { E(e)->normalize_matches(); }); }
int lrr_est_set_number_points(lrr_est *e, int64_t n) { return guard([&] { E(e)->set_number_points(n); }); }
int lrr_est_inject_initial_scale(lrr_est *e, double f, double b) { return guard([&] { E(e)->inject_initial_scale(f, b); }); }
int lrr_est_set_subset_order(lrr_est *e, int dir, const int64_t *order, int64_t n) { return guard([&] { E(e)->set_subset_order(dir, order, n); }); }
int lrr_est_set_nn_method(lrr_est *e, int m) { return guard([&] { E(e)->set_nn_method(m); }); }
// the caller's set_voronoi_maps() / set_fixed_voronoi_map(); lookup: what lrr_amd::voronoi_lookup::enable() is set to first (-1: leave)
int lrr_est_set_voronoi_maps(lrr_est *e, int lookup) {
  return guard([&] { if (lookup >= 0) lrr_amd::voronoi_lookup::enable(lookup != 0); E(e)->set_voronoi_maps(); });
}
int lrr_est_find_closest_euclidean(lrr_est *e) { return guard([&] { E(e)->find_closest_euclidean(); }); }
int lrr_est_one_iteration(lrr_est *e, unsigned it, int *converged) { return guard([&] { bool c = E(e)->one_iteration(it); if (converged) *converged = c ? 1 : 0; }); }
// `count` iterations numbered from `first`, whatever they report (a benchmark's timed steps: no interpreter between them)
int lrr_est_iterate(lrr_est *e, unsigned first, unsigned count, int *converged) {
  return guard([&] { bool c = false; for (unsigned k = 0; k < count; ++k) c = E(e)->one_iteration(first + k); if (converged) *converged = c ? 1 : 0; });
}
int lrr_est_run(lrr_est *e, unsigned max_it, unsigned *done, int *converged) { return guard([&] { bool c = E(e)->run(max_it, done, nullptr); if (converged) *converged = c ? 1 : 0; }); }
// same loop, also counting the matched pairs formed (sum over iterations of the active subset sizes)
int lrr_est_run_counted(lrr_est *e, unsigned max_it, unsigned *done, int *converged, int64_t *pairs) { return guard([&] { if (pairs) *pairs = 0; bool c = E(e)->run(max_it, done, pairs); if (converged) *converged = c ? 1 : 0; }); }
// the caller's loop with its validTransform break (rrl/itkLocationRegistration.cxx:2418-2482): *valid = 0 when it ended the loop
int lrr_est_run_caller(lrr_est *e, unsigned max_it, int check_valid, unsigned *done, int *converged, int64_t *pairs, int *valid) {
  return guard([&] { if (pairs) *pairs = 0; bool c = E(e)->run(max_it, done, pairs, check_valid != 0, valid); if (converged) *converged = c ? 1 : 0; });
}
int lrr_est_set_subset_seed(lrr_est *e, uint64_t seed) { return guard([&] { E(e)->set_subset_seed(seed); }); }
int lrr_est_estimate_LS(lrr_est *e, int p) { return guard([&] { E(e)->estimate_LS(p != 0); }); }
int lrr_est_estimate_LS_backward(lrr_est *e, int p) { return guard([&] { E(e)->estimate_LS_backward(p != 0); }); }
int lrr_est_get_transform(lrr_est *e, int dir, double *A, double *t, double *c, double *covar) { return guard([&] { E(e)->get_transform(dir, A, t, c, covar); }); }
int lrr_est_RMS_error(lrr_est *e, double *out) { return guard([&] { *out = E(e)->RMS_error(); }); }
int lrr_est_median_error(lrr_est *e, double *out) { return guard([&] { *out = E(e)->median_error(); }); }
int lrr_est_weighted_error(lrr_est *e, int dir, double *out) { return guard([&] { *out = E(e)->weighted_error(dir); }); }
int lrr_est_sheet_angles(lrr_est *e, int dir, double *out) { return guard([&] { *out = E(e)->sheet_angles(dir); }); }
int lrr_est_tube_angles(lrr_est *e, int dir, double *out) { return guard([&] { *out = E(e)->tube_angles(dir); }); }
int lrr_est_compute_transfer_error_covariance(lrr_est *e, double *cf, double *cb) { return guard([&] { E(e)->transfer_error_covariance(cf, cb); }); }
int lrr_est_oscillated(lrr_est *e) { return E(e)->oscillated() ? 1 : 0; }
int lrr_est_finest_level(lrr_est *e) { return E(e)->finest_level() ? 1 : 0; }
int64_t lrr_est_num_matches(lrr_est *e, int dir) { int64_t n = 0; guard([&] { n = E(e)->num_matches(dir); }); return n; }
int lrr_est_get_matches(lrr_est *e, int dir, int64_t *f, int64_t *t, double *w) { return guard([&] { E(e)->get_matches(dir, f, t, w); }); }
int64_t lrr_est_get_number_points(lrr_est *e, int dir) { int64_t n = 0; guard([&] { n = E(e)->number_points(dir); }); return n; }
int lrr_est_last_scale(lrr_est *e, int dir, double *out) { return guard([&] { *out = E(e)->last_scale(dir); }); }
// the device state of one direction (C-ABI handle: tests reach the lrr_icp_* diagnostics with it)
lrr_icp_dir *lrr_est_device_dir(lrr_est *e, int dir) { return E(e)->device_dir(dir); }
int lrr_est_last_normal_equations(lrr_est *e, double *AtA, double *Atb) { return guard([&] { E(e)->normal_equations(AtA, Atb); }); }


// ---- many query locations per submission (include/lrr_amd/location_batch.h) ---------------------------
// one row of doubles per location:
//   0 converged, 1 iterations, 2..10 A, 11..13 t, 14..16 centre, 17 weighted_error, 18 RMS_error, 19 (caller's index),
//   20 enough_features, 21 oscillated, 22 valid_transform, 23 n_moving_inside, 24 n_fixed_inside, 25 matched_pairs,
//   26 initial_RMS, 27 weighted_error_backward, 28 sheet_angles, 29 tube_angles, 30 sheet_angles_backward,
//   31 tube_angles_backward, 32..39 the 8 decision measurements, 40..54 backward A, t, centre,
//   55..198 forward covariance (12 x 12), 199..342 backward covariance
enum { LRR_EST_BATCH_ROW = 343 };
typedef struct lrr_est_batch lrr_est_batch;
typedef lrr_amd::rrl_location_batch<cdcl_feature_ICP<3> > batch_type;
#define BT(b) (reinterpret_cast<batch_type *>(b))

int lrr_est_batch_row_width() { return LRR_EST_BATCH_ROW; }

lrr_est_batch *lrr_est_batch_create(const lrr_est_features *moving, const lrr_est_features *fixed) {
  batch_type *out = nullptr;
  int rc = guard([&] {
    if (!moving || !fixed) throw std::runtime_error("lrr_est_batch_create: null argument");
    out = new batch_type(make_features<cdcl_feature_ICP<3> >(moving), make_features<cdcl_feature_ICP<3> >(fixed));
  });
  return rc ? nullptr : reinterpret_cast<lrr_est_batch *>(out);
}
void lrr_est_batch_destroy(lrr_est_batch *b) { delete BT(b); }
// the batch's matcher: method >= 0: set_nn_method(method); method < 0: the caller's set_voronoi_maps() (null images), after
// lrr_amd::voronoi_lookup::enable(lookup != 0) when lookup >= 0
int lrr_est_batch_set_matcher(lrr_est_batch *b, int method, int lookup) {
  return guard([&] {
    if (method >= 0) { BT(b)->set_nn_method(method); return; }
    if (lookup >= 0) lrr_amd::voronoi_lookup::enable(lookup != 0);
    BT(b)->set_voronoi_maps((const void *)0, (const void *)0);
  });
}
lrr_icp_batch *lrr_est_batch_handle(lrr_est_batch *b) { return BT(b)->handle(); }

int lrr_est_batch_run(lrr_est_batch *bt, int64_t B, const double *A0, const double *t0, const double *c0, const double *moving_box,
                      const double *fixed_box, const uint64_t *seeds, unsigned max_iterations, int diagnostics, int check_valid,
                      double *rows) {
  return guard([&] {
    const bool timing = std::getenv("LRR_BATCH_TIMING") != nullptr;
    const auto w0 = std::chrono::steady_clock::now();
    std::vector<lrr_amd::location_query> q((size_t)B);
    for (int64_t b = 0; b < B; ++b) {
      q[(size_t)b].transform = trans_factory<12>::make(A0 + 9 * b, t0 + 3 * b, c0 + 3 * b);
      for (int d = 0; d < 3; ++d) {
        q[(size_t)b].moving_x0[d] = moving_box[6 * b + d]; q[(size_t)b].moving_x1[d] = moving_box[6 * b + 3 + d];
        q[(size_t)b].fixed_x0[d] = fixed_box[6 * b + d]; q[(size_t)b].fixed_x1[d] = fixed_box[6 * b + 3 + d];
      }
      q[(size_t)b].seed = seeds[b];
    }
    BT(bt)->set_max_iterations(max_iterations);
    BT(bt)->set_diagnostics(diagnostics != 0);
    BT(bt)->set_check_valid_transform(check_valid != 0);
    std::vector<lrr_amd::location_result> res;
    const auto w1 = std::chrono::steady_clock::now();
    BT(bt)->run(q, res);
    const auto w2 = std::chrono::steady_clock::now();
    for (int64_t b = 0; b < B; ++b) {
      double *r = rows + (size_t)b * LRR_EST_BATCH_ROW;
      const lrr_amd::location_result &R = res[(size_t)b];
      for (int k = 0; k < LRR_EST_BATCH_ROW; ++k) r[k] = 0.0;
      r[0] = R.converged; r[1] = R.iterations; r[17] = R.weighted_error; r[18] = R.RMS_error; r[19] = (double)b;
      r[20] = R.enough_features; r[21] = R.oscillated; r[22] = R.valid_transform;
      r[23] = (double)R.n_moving_inside; r[24] = (double)R.n_fixed_inside; r[25] = (double)R.matched_pairs;
      r[26] = R.initial_RMS; r[27] = R.weighted_error_backward;
      r[28] = R.sheet_angles; r[29] = R.tube_angles; r[30] = R.sheet_angles_backward; r[31] = R.tube_angles_backward;
      for (int k = 0; k < 8; ++k) r[32 + k] = R.measurements[k];
      if (R.forward) read_trans<12>(R.forward, r + 2, r + 11, r + 14, r + 55);
      if (R.backward) read_trans<12>(R.backward, r + 40, r + 49, r + 52, r + 199);
    }
    if (timing) {
      const auto w3 = std::chrono::steady_clock::now();
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      std::fprintf(stderr, "[lrr batch wrapper] queries %.2f ms run %.2f ms rows %.2f ms\n", ms(w0, w1), ms(w1, w2), ms(w2, w3));
    }
  });
}


// phases of the last lrr_est_batch_run as "name=ms;name=ms;..." (wall clock inside rrl_location_batch::run); returns the
// length written (0: nothing ran yet)
int lrr_est_batch_last_timing(lrr_est_batch *bt, char *out, int cap) {
  if (!bt || !out || cap <= 0) return 0;
  std::string s;
  for (auto const &p : BT(bt)->last_phases()) { char b[96]; std::snprintf(b, sizeof b, "%s=%.6f;", p.first.c_str(), p.second); s += b; }
  if ((int)s.size() >= cap) s.resize((size_t)cap - 1);
  std::memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// ---- feature files (include/lrr_amd/vtp_features.h) -----------------------------------------------------
typedef struct lrr_vtp lrr_vtp;
#define VT(v) (reinterpret_cast<lrr_amd::vtp_arrays *>(v))

lrr_vtp *lrr_vtp_open(const char *path) {
  lrr_amd::vtp_arrays *a = new lrr_amd::vtp_arrays;
  int rc = guard([&] { lrr_amd::read_vtp(path, *a); });
  if (rc) { delete a; return nullptr; }
  return reinterpret_cast<lrr_vtp *>(a);
}
void lrr_vtp_close(lrr_vtp *v) { delete VT(v); }
int64_t lrr_vtp_count(lrr_vtp *v) { return (int64_t)VT(v)->n; }
// the arrays as stored (widened to double); present[k] = 0 when array k is absent from the file
int lrr_vtp_arrays(lrr_vtp *v, double *points, double *proj, double *strengths, double *shapes, double *normals, double *binormals,
                   int *present) {
  return guard([&] {
    const lrr_amd::vtp_arrays &a = *VT(v);
    const std::vector<double> *src[6] = {&a.points, &a.errorProjectors, &a.strengths, &a.shapes, &a.normals, &a.binormals};
    double *dst[6] = {points, proj, strengths, shapes, normals, binormals};
    for (int k = 0; k < 6; ++k) {
      if (present) present[k] = src[k]->empty() ? 0 : 1;
      if (dst[k] && !src[k]->empty()) std::memcpy(dst[k], src[k]->data(), sizeof(double) * src[k]->size());
    }
  });
}
// the features vtkPolyDataToFeaturesICPFilter builds from the file: dirs9 = up to three directions_ per feature
int lrr_vtp_features(lrr_vtp *v, double *loc, double *proj, double *strength, uint8_t *shape, uint8_t *ndirs, double *dirs9) {
  return guard([&] {
    std::vector<cdcl_feature_ICP<3>::sptr> f = lrr_amd::features_from_vtp<cdcl_feature_ICP<3> >(*VT(v));
    for (size_t i = 0; i < f.size(); ++i) {
      for (int d = 0; d < 3; ++d) loc[3 * i + d] = f[i]->location_[d];
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) proj[9 * i + 3 * r + c] = f[i]->error_projector_(r, c);
      strength[i] = (double)f[i]->strength_;
      shape[i] = (uint8_t)f[i]->shape_;
      ndirs[i] = (uint8_t)f[i]->directions_.size();
      for (int k = 0; k < 9; ++k) dirs9[9 * i + k] = 0.0;
      for (size_t k = 0; k < f[i]->directions_.size() && k < 3; ++k)
        for (int d = 0; d < 3; ++d) dirs9[9 * i + 3 * k + d] = f[i]->directions_[k][d];
    }
  });
}

// ---- covariance-driven EM estimator (cdcl_estimation, include/lrr_amd/cdcl_estimation.h) ------------------------------
struct lrr_cdc_em {
  int dof = 12;
  std::vector<lrr_icp_cloud *> moving, fixed;   // one pair per resolution, [0] the finest
  cdcl_trans<3, 12>::sptr trans12;
  cdcl_trans<3, 6>::sptr trans6;
  lrr_amd::cdcl_estimation_em *em12 = nullptr;
  lrr_amd::cdcl_estimation_em_rigid *em6 = nullptr;
  ~lrr_cdc_em() {
    delete em12;
    delete em6;
    for (lrr_icp_cloud *c : moving) if (c) lrr_icp_cloud_destroy(c);
    for (lrr_icp_cloud *c : fixed) if (c) lrr_icp_cloud_destroy(c);
  }
};
static lrr_icp_cloud *cdc_cloud(const lrr_est_features *f) {
  if (!f || !f->cov) throw std::runtime_error("cdc_em: the features need covariances");
  // (cdcl_feature: location, covariance, strength — f->proj is not read; the projector slot takes the covariance's normal projector)
  return lrr_amd::cdcl_feature_cloud(f->n, f->loc, f->cov, f->strength);
}
// dof 12: affine; dof 6: rigid3d (small-angle increment).  levels pairs of feature sets, [0] the finest, with the point spacing
// of every fixed set (levels == 1: spacing may be NULL); covar: dof x dof
int lrr_cdc_em_create_pyramid(int dof, int levels, const lrr_est_features *const *moving, const lrr_est_features *const *fixed, const double *spacing,
                              const double *A, const double *t, const double *c, const double *covar, uint64_t seed, int max_matches, lrr_cdc_em **out) {
  return guard([&] {
    if ((dof != 6 && dof != 12) || levels < 1 || !moving || !fixed) throw std::runtime_error("cdc_em: bad argument");
    std::unique_ptr<lrr_cdc_em> h(new lrr_cdc_em);
    h->dof = dof;
    for (int r = 0; r < levels; ++r) { h->moving.push_back(cdc_cloud(moving[r])); h->fixed.push_back(cdc_cloud(fixed[r])); }
    std::vector<double> sp(spacing ? spacing : nullptr, spacing ? spacing + levels : nullptr);
    if (dof == 12) {
      h->trans12 = trans_factory<12>::make(A, t, c);
      vnl_matrix_fixed<double, 12, 12> C;
      for (unsigned r = 0; r < 12; ++r) for (unsigned k = 0; k < 12; ++k) C(r, k) = covar[r * 12 + k];
      h->trans12->set_covariance(C);
      if (levels == 1) h->em12 = new lrr_amd::cdcl_estimation_em(h->moving[0], h->fixed[0], h->trans12, seed, max_matches);
      else h->em12 = new lrr_amd::cdcl_estimation_em(h->moving, h->fixed, h->trans12, sp, seed, max_matches);
    } else {
      h->trans6 = trans_factory<6>::make(A, t, c);
      vnl_matrix_fixed<double, 6, 6> C;
      for (unsigned r = 0; r < 6; ++r) for (unsigned k = 0; k < 6; ++k) C(r, k) = covar[r * 6 + k];
      h->trans6->set_covariance(C);
      if (levels == 1) h->em6 = new lrr_amd::cdcl_estimation_em_rigid(h->moving[0], h->fixed[0], h->trans6, seed, max_matches);
      else h->em6 = new lrr_amd::cdcl_estimation_em_rigid(h->moving, h->fixed, h->trans6, sp, seed, max_matches);
    }
    *out = h.release();
  });
}
int lrr_cdc_em_create(const lrr_est_features *moving, const lrr_est_features *fixed, const double *A, const double *t, const double *c,
                      const double *covar144, uint64_t seed, int max_matches, lrr_cdc_em **out) {
  return lrr_cdc_em_create_pyramid(12, 1, &moving, &fixed, nullptr, A, t, c, covar144, seed, max_matches, out);
}
void lrr_cdc_em_destroy(lrr_cdc_em *h) { delete h; }
int lrr_cdc_em_one_iteration(lrr_cdc_em *h, unsigned iteration, int *converged, double *info) {
  return guard([&] { if (h->em12) cdc_em_iterate(h->em12, iteration, converged, info, (double *)nullptr); else cdc_em_iterate(h->em6, iteration, converged, info, (double *)nullptr); });
}
int lrr_cdc_em_one_iteration_state(lrr_cdc_em *h, unsigned iteration, int *converged, double *info, double *state) {
  return guard([&] { if (h->em12) cdc_em_iterate(h->em12, iteration, converged, info, state); else cdc_em_iterate(h->em6, iteration, converged, info, state); });
}
// normalize_matches() of the base class (cdcl/cdcl_estimation_ICP.h:76); on by default here, `on` = 0 switches it off
int lrr_cdc_em_normalize_matches(lrr_cdc_em *h, int on) {
  return guard([&] { if (h->em12) h->em12->normalize_matches(on != 0); else h->em6->normalize_matches(on != 0); });
}
// initialize() (cdcl/cdcl_estimation.txx:70-84); scale: the scale of its robust weights (may be NULL)
int lrr_cdc_em_initialize(lrr_cdc_em *h, double *scale) {
  return guard([&] {
    if (h->em12) { h->em12->initialize(); if (scale) *scale = h->em12->initial_scale(); }
    else { h->em6->initialize(); if (scale) *scale = h->em6->initial_scale(); }
  });
}
// run() (cdcl/cdcl_estimation.txx:190-204): initialize + one_iteration until converged, 85 at most
int lrr_cdc_em_run(lrr_cdc_em *h, int *converged, unsigned *iterations) {
  return guard([&] {
    bool c = h->em12 ? h->em12->run(iterations) : h->em6->run(iterations);
    if (converged) *converged = c ? 1 : 0;
  });
}
// the resolution the estimator is at (0 = the finest)
int lrr_cdc_em_resolution(lrr_cdc_em *h, int *res) {
  return guard([&] { *res = (int)(h->em12 ? h->em12->resolution() : h->em6->resolution()); });
}
int lrr_cdc_em_get_transform(lrr_cdc_em *h, double *A, double *t, double *c, double *covar) {
  return guard([&] { if (h->em12) read_trans<12>(h->em12->get_transform(), A, t, c, covar); else read_trans<6>(h->em6->get_transform(), A, t, c, covar); });
}
// rows of the CURRENT resolution (row_count: one entry per moving feature of that resolution)
int lrr_cdc_em_rows(lrr_cdc_em *h, int64_t *row_count, int64_t *to_idx, double *w) {
  return guard([&] { lrr_amd::check(lrr_icp_cdc_set_get(h->em12 ? h->em12->rows().handle() : h->em6->rows().handle(), row_count, to_idx, w), "cdc rows"); });
}

}  // extern "C"
