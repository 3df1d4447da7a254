// cdcl_types.h — the value types of the estimator interface: features, matches, transforms.
//
// These are NOT what this project replaces: inside the reference tree define
// LRR_AMD_USE_REFERENCE_TYPES and the reference's own headers are used unchanged
// (cdcl/cdcl_feature.h, cdcl_feature_with_shape.h, cdcl_feature_ICP.h, cdcl_match.h,
// cdcl_trans.h, cdcl_trans_rigid3d.h, cdcl_trans_affine.h).  Stand-alone (no VXL in this image)
// the look-alikes below keep the same names, public fields and method meaning so that the
// estimator classes, their tests and the flat test API compile against either.
#ifndef LRR_AMD_CDCL_TYPES_H_
#define LRR_AMD_CDCL_TYPES_H_

#ifdef LRR_AMD_USE_REFERENCE_TYPES
#include "cdcl_feature.h"
#include "cdcl_feature_ICP.h"
#include "cdcl_feature_with_shape.h"
#include "cdcl_match.h"
#include "cdcl_trans.h"
#include "cdcl_trans_affine.h"
#include "cdcl_trans_rigid3d.h"
#else

#include <cassert>
#include <cmath>

#include "lrr_amd/linalg.h"
#include "lrr_amd/vxl_compat.h"

// ---- features (cdcl/cdcl_feature.h:17-64, cdcl_feature_with_shape.h:22-89, cdcl_feature_ICP.h:22-96)
template <unsigned int dim>
class cdcl_feature : public vbl_ref_count {
 public:
  typedef vbl_smart_ptr<cdcl_feature<dim> > sptr;
  typedef double coord_type;
  cdcl_feature() : location_(0.0), strength_(0.0), covariance_(0.0) {}
  cdcl_feature(vnl_vector_fixed<coord_type, dim> const &location, double const &strength,
               vnl_matrix_fixed<coord_type, dim, dim> const &covariance)
      : location_(location), strength_(strength), covariance_(covariance) {}
  virtual typename cdcl_feature<dim>::sptr clone() { return new cdcl_feature<dim>(*this); }

  vnl_vector_fixed<coord_type, dim> location_;
  double strength_;
  vnl_matrix_fixed<coord_type, dim, dim> covariance_;
};

template <unsigned int dim>
class cdcl_feature_with_shape : public cdcl_feature<dim> {
 public:
  typedef vbl_smart_ptr<cdcl_feature_with_shape<dim> > sptr;
  typedef typename cdcl_feature<dim>::coord_type coord_type;
  enum shape_type { CORNER, TUBE, SHEET };
  cdcl_feature_with_shape() : cdcl_feature<dim>(), shape_(CORNER), error_projector_(0.0) {}
  cdcl_feature_with_shape(vnl_vector_fixed<coord_type, dim> const &location, double const &strength,
                          vnl_matrix_fixed<coord_type, dim, dim> const &covariance,
                          vnl_matrix_fixed<coord_type, dim, dim> const &error_projector, shape_type shape,
                          vcl_vector<vnl_vector_fixed<coord_type, dim> > const &directions)
      : cdcl_feature<dim>(location, strength, covariance), shape_(shape), error_projector_(error_projector), directions_(directions) {}
  virtual typename cdcl_feature<dim>::sptr clone() { return new cdcl_feature_with_shape<dim>(*this); }

  shape_type shape_;
  vnl_matrix_fixed<coord_type, dim, dim> error_projector_;
  vcl_vector<vnl_vector_fixed<coord_type, dim> > directions_;
};

template <unsigned int dim>
class cdcl_feature_ICP : public vbl_ref_count {
 public:
  typedef vbl_smart_ptr<cdcl_feature_ICP<dim> > sptr;
  typedef double coord_type;
  typedef vnl_vector_fixed<coord_type, dim> location_type;
  enum shape_type { CORNER, TUBE, SHEET };
  cdcl_feature_ICP() : location_(0.0), strength_(0.0f), shape_(CORNER), error_projector_(0.0) {}
  cdcl_feature_ICP(location_type const &location, double const &strength,
                   vnl_matrix_fixed<coord_type, dim, dim> const &error_projector, shape_type shape,
                   vcl_vector<location_type> const &directions)
      : location_(location), strength_((float)strength), shape_(shape), error_projector_(error_projector), directions_(directions) {}
  virtual typename cdcl_feature_ICP<dim>::sptr clone() { return new cdcl_feature_ICP<dim>(*this); }

  location_type location_;
  float strength_;  // float in the reference (cdcl_feature_ICP.h:83): strength products are float
  shape_type shape_;
  vnl_matrix_fixed<coord_type, dim, dim> error_projector_;
  vcl_vector<location_type> directions_;
};

// ---- match (cdcl/cdcl_match.h:17-44): one from_, many to_ with weights; always one on this path
template <unsigned int dim, class feature_typeT = cdcl_feature<dim> >
class cdcl_match : public vbl_ref_count {
 public:
  typedef vbl_smart_ptr<cdcl_match> sptr;
  typedef feature_typeT feature_type;
  typedef typename feature_type::sptr feature_sptr_type;
  cdcl_match() : from_(0) {}
  cdcl_match(feature_sptr_type const &from, vcl_vector<feature_sptr_type> const &to, vcl_vector<double> const &w)
      : from_(from), to_(to), w_(w) {}
  feature_sptr_type from_;
  vcl_vector<feature_sptr_type> to_;
  vcl_vector<double> w_;
};

// ---- transforms (cdcl/cdcl_trans.h:21-148)
#define cdcl_type_macro(classname)                                              \
  typedef classname self;                                                       \
  static const vcl_type_info &type_id() { return typeid(self); }                \
  virtual bool is_type(const vcl_type_info &type) const { return (typeid(self) == type) != 0; }

template <unsigned int dim, unsigned int dof>
class cdcl_trans : public vbl_ref_count {
 public:
  typedef vbl_smart_ptr<cdcl_trans<dim, dof> > sptr;
  typedef double coord_type;
  typedef vnl_matrix_fixed<coord_type, dof, dof> covariance_type;
  cdcl_trans() : covar_(0.0), covarJ_(0.0), center_moving_(0.0) {}
  virtual sptr create_new() = 0;
  virtual sptr clone() = 0;
  virtual sptr inverse() = 0;
  virtual vnl_vector<coord_type> get_parameterization() = 0;
  virtual void set_parameterization(vnl_vector<coord_type> const &param) = 0;
  virtual vnl_vector_fixed<coord_type, dim> map_loc(vnl_vector_fixed<coord_type, dim> const &from) const = 0;
  virtual vnl_matrix_fixed<coord_type, dim, dim> jacobian_wrt_loc() const = 0;
  virtual void jacobian_wrt_par_thread(vnl_vector_fixed<coord_type, dim> const &location,
                                       vnl_matrix_fixed<coord_type, dim, dof> &jacobian_wrt_par) = 0;
  virtual vnl_matrix_fixed<coord_type, dim, dof> const &jacobian_wrt_par(vnl_vector_fixed<coord_type, dim> const &location) {
    this->jacobian_wrt_par_thread(location, jacobian_wrt_par_);
    return jacobian_wrt_par_;
  }
  virtual void normalize(coord_type const &avg_rad_moving, coord_type const &avg_rad_fixed,
                         vnl_vector_fixed<coord_type, dim> const &center_moving,
                         vnl_vector_fixed<coord_type, dim> const &center_fixed) = 0;
  virtual void unnormalize(coord_type const &avg_rad_moving, coord_type const &avg_rad_fixed,
                           vnl_vector_fixed<coord_type, dim> const &center_moving,
                           vnl_vector_fixed<coord_type, dim> const &center_fixed) = 0;
  virtual covariance_type const &get_covariance() { return covar_; }
  virtual vnl_matrix_fixed<coord_type, dim, dim> const &get_covarianceJ() { return covarJ_; }
  virtual void set_covariance(covariance_type const &covar) { covar_ = covar; }
  virtual void set_covarianceJ(vnl_matrix_fixed<coord_type, dim, dim> const &covarJ) { covarJ_ = covarJ; }
  virtual void recompose_increment(typename cdcl_trans<dim, dof>::sptr const &) {}
  virtual vnl_matrix_fixed<coord_type, dim, dim> &get_A() = 0;
  virtual vnl_vector_fixed<coord_type, dim> &get_translation() = 0;
  virtual void print(vcl_ostream &) {}
  cdcl_type_macro(cdcl_trans);

 protected:
  vnl_matrix_fixed<coord_type, dim, dof> jacobian_wrt_par_;
  covariance_type covar_;
  vnl_matrix_fixed<coord_type, dim, dim> covarJ_;

 public:
  vnl_vector_fixed<coord_type, dim> center_moving_;
};

namespace lrr_amd {
namespace detail {
typedef vnl_vector_fixed<double, 3> vec3;
typedef vnl_matrix_fixed<double, 3, 3> mat3;
inline mat3 inverse_by_svd(const mat3 &A) {
  mat3 out;
  linalg::inverse3(A.data_block(), out.data_block());
  return out;
}
}  // namespace detail
}  // namespace lrr_amd

// ---- 3-D rigid (cdcl/cdcl_trans_rigid3d.{h,cxx}); parameters (alpha, beta, gamma, tx, ty, tz)
class cdcl_trans_rigid3d : public cdcl_trans<3, 6> {
 public:
  static const unsigned int dim = 3, dof = 6;
  typedef vbl_smart_ptr<cdcl_trans_rigid3d> sptr;
  typedef lrr_amd::detail::vec3 vec3;
  typedef lrr_amd::detail::mat3 mat3;
  cdcl_trans_rigid3d() : alpha_(0.0), beta_(0.0), gamma_(0.0), t_(0.0) { R_.set_identity(); }
  // R = Rx(alpha) Ry(beta) Rz(gamma)  (cdcl_trans_rigid3d.cxx:31-64)
  cdcl_trans_rigid3d(coord_type const &alpha, coord_type const &beta, coord_type const &gamma, vec3 const &t)
      : alpha_(alpha), beta_(beta), gamma_(gamma), t_(t) {
    mat3 Rx(0.0), Ry(0.0), Rz(0.0);
    double sa = std::sin(alpha), ca = std::cos(alpha), sb = std::sin(beta), cb = std::cos(beta), sg = std::sin(gamma), cg = std::cos(gamma);
    Rx(0, 0) = 1; Rx(1, 1) = ca; Rx(1, 2) = -sa; Rx(2, 1) = sa; Rx(2, 2) = ca;
    Ry(0, 0) = cb; Ry(0, 2) = sb; Ry(1, 1) = 1; Ry(2, 0) = -sb; Ry(2, 2) = cb;
    Rz(0, 0) = cg; Rz(0, 1) = -sg; Rz(1, 0) = sg; Rz(1, 1) = cg; Rz(2, 2) = 1;
    R_ = Rx * Ry * Rz;
  }
  // from a rotation matrix: Euler angles recovered as in cdcl_trans_rigid3d.cxx:83-108
  cdcl_trans_rigid3d(mat3 const &R, vec3 const &t, vec3 const &center_moving) : R_(R), t_(t) {
    beta_ = std::asin(R_(0, 2));
    double C = std::cos(beta_);
    if (std::fabs(C) > 1e-6) {
      alpha_ = std::atan2(-R(1, 2) / C, R(2, 2) / C);
      gamma_ = std::atan2(-R(0, 1) / C, R(0, 0) / C);
    } else {
      alpha_ = 0;
      gamma_ = std::atan2(R(1, 0), R(1, 1));
    }
    this->center_moving_ = center_moving;
  }
  virtual cdcl_trans<3, 6>::sptr create_new() { return new cdcl_trans_rigid3d; }
  virtual cdcl_trans<3, 6>::sptr clone() { return new cdcl_trans_rigid3d(*this); }
  virtual cdcl_trans<3, 6>::sptr inverse() {  // (R^-1, -R^-1 t + c, 0): cdcl_trans_rigid3d.cxx:113-118
    mat3 Ri = lrr_amd::detail::inverse_by_svd(R_);
    return new cdcl_trans_rigid3d(Ri, -(Ri * t_) + this->center_moving_, vec3(0.0));
  }
  virtual vnl_vector<coord_type> get_parameterization() {
    vnl_vector<coord_type> p(6, 0.0);
    p[0] = alpha_; p[1] = beta_; p[2] = gamma_; p[3] = t_[0]; p[4] = t_[1]; p[5] = t_[2];
    return p;
  }
  // small-angle R = I + [w]x (cdcl_trans_rigid3d.cxx:140-166)
  virtual void set_parameterization(vnl_vector<coord_type> const &p) {
    assert(p.size() == 6);
    alpha_ = p[0]; beta_ = p[1]; gamma_ = p[2];
    R_(0, 0) = 1.0;   R_(0, 1) = -p[2]; R_(0, 2) = p[1];
    R_(1, 0) = p[2];  R_(1, 1) = 1.0;   R_(1, 2) = -p[0];
    R_(2, 0) = -p[1]; R_(2, 1) = p[0];  R_(2, 2) = 1.0;
    t_[0] = p[3]; t_[1] = p[4]; t_[2] = p[5];
  }
  virtual vec3 map_loc(vec3 const &from) const { return R_ * (from - this->center_moving_) + t_; }
  virtual mat3 jacobian_wrt_loc() const { return R_; }
  virtual void jacobian_wrt_par_thread(vec3 const &location, vnl_matrix_fixed<coord_type, 3, 6> &J) {
    vec3 l = location - this->center_moving_;
    J.fill(0.0);
    J(1, 0) = -l[2]; J(2, 0) = l[1];
    J(0, 1) = l[2];  J(2, 1) = -l[0];
    J(0, 2) = -l[1]; J(1, 2) = l[0];
    J(0, 3) = J(1, 4) = J(2, 5) = 1.0;
  }
  // covariance and centre untouched (cdcl_trans_rigid3d.cxx:260-280)
  virtual void normalize(coord_type const &rm, coord_type const &rf, vec3 const &cm, vec3 const &cf) {
    R_ = (rm / rf) * R_;
    t_ = t_ / rf + (R_ * (cm - this->center_moving_)) / rm - cf / rf;
  }
  virtual void unnormalize(coord_type const &rm, coord_type const &rf, vec3 const &cm, vec3 const &cf) {
    R_ = (rf / rm) * R_;
    t_ = -(R_ * cm) + rf * t_ + cf;
    this->center_moving_ = cm;
  }
  // dR <- U V^T; R <- R dR; t <- dR t + dt (cdcl_trans_rigid3d.cxx:309-325)
  virtual void recompose_increment(cdcl_trans<3, 6>::sptr const &increment) {
    mat3 dR = increment->get_A();
    vec3 dt = increment->get_translation();
    double U[9], s[3], V[9];
    lrr_amd::linalg::svd3(dR.data_block(), U, s, V);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double a = 0.0;
        for (int k = 0; k < 3; ++k) a += U[i * 3 + k] * V[j * 3 + k];
        dR(i, j) = a;
      }
    R_ = R_ * dR;
    t_ = dR * t_ + dt;
  }
  virtual mat3 &get_A() { return R_; }
  virtual vec3 &get_translation() { return t_; }
  cdcl_type_macro(cdcl_trans_rigid3d);

 private:
  coord_type alpha_, beta_, gamma_;
  mat3 R_;
  vec3 t_;
};

// ---- affine (cdcl/cdcl_trans_affine.{h,txx}); parameters: A row-major, then t
template <unsigned int dm, unsigned int df>
class cdcl_trans_affine : public cdcl_trans<dm, df> {
 public:
  typedef vbl_smart_ptr<cdcl_trans_affine<dm, df> > sptr;
  typedef typename cdcl_trans<dm, df>::coord_type coord_type;
  typedef vnl_vector_fixed<coord_type, dm> vec;
  typedef vnl_matrix_fixed<coord_type, dm, dm> mat;
  cdcl_trans_affine() : A_(0.0), t_(0.0) { A_.set_identity(); }
  cdcl_trans_affine(mat const &A, vec const &t, vec const &center_moving) : A_(A), t_(t) { this->center_moving_ = center_moving; }
  virtual typename cdcl_trans<dm, df>::sptr create_new() { return new cdcl_trans_affine<dm, df>; }
  virtual typename cdcl_trans<dm, df>::sptr clone() { return new cdcl_trans_affine<dm, df>(*this); }
  virtual typename cdcl_trans<dm, df>::sptr inverse() {  // cdcl_trans_affine.txx:54-59
    static_assert(dm == 3, "3-D only");
    mat Ai = lrr_amd::detail::inverse_by_svd(A_);
    return new cdcl_trans_affine<dm, df>(Ai, -(Ai * t_) + this->center_moving_, vec(0.0));
  }
  virtual vnl_vector<coord_type> get_parameterization() {
    vnl_vector<coord_type> p(df, 0.0);
    unsigned k = 0;
    for (unsigned r = 0; r < dm; ++r) for (unsigned c = 0; c < dm; ++c) p[k++] = A_(r, c);
    for (unsigned d = 0; d < dm; ++d) p[k++] = t_[d];
    return p;
  }
  virtual void set_parameterization(vnl_vector<coord_type> const &p) {
    assert(p.size() == df);
    unsigned k = 0;
    for (unsigned r = 0; r < dm; ++r) for (unsigned c = 0; c < dm; ++c) A_(r, c) = p[k++];
    for (unsigned d = 0; d < dm; ++d) t_[d] = p[k++];
  }
  virtual vec map_loc(vec const &from) const { return A_ * (from - this->center_moving_) + t_; }
  virtual mat jacobian_wrt_loc() const { return A_; }
  virtual void jacobian_wrt_par_thread(vec const &location, vnl_matrix_fixed<coord_type, dm, df> &J) {
    vec l = location - this->center_moving_;
    J.fill(0.0);
    for (unsigned r = 0; r < dm; ++r) {
      for (unsigned c = 0; c < dm; ++c) J(r, r * dm + c) = l[c];
      J(r, dm * dm + r) = 1.0;
    }
  }
  // also maps the covariance by B C B^T and zeroes the centre (cdcl_trans_affine.txx:220-253)
  virtual void normalize(coord_type const &rm, coord_type const &rf, vec const &cm, vec const &cf) {
    A_ = (rm / rf) * A_;
    t_ = t_ / rf + (A_ * (cm - this->center_moving_)) / rm - cf / rf;
    vnl_matrix_fixed<coord_type, df, df> B(0.0);
    const unsigned dm2 = dm * dm;
    for (unsigned d = 0; d < dm2; ++d) B(d, d) = rm / rf;
    const vec factor = (cm - this->center_moving_) / rf;
    for (unsigned r = 0; r < dm; ++r) for (unsigned c = 0; c < dm; ++c) B(dm2 + r, r * dm + c) = factor[c];
    for (unsigned d = dm2; d < df; ++d) B(d, d) = 1.0 / rf;
    this->covar_ = B * this->covar_ * B.transpose();
    this->center_moving_.fill(0.0);
  }
  virtual void unnormalize(coord_type const &rm, coord_type const &rf, vec const &cm, vec const &cf) {
    A_ = (rf / rm) * A_;
    t_ = rf * t_ + cf;
    vnl_matrix_fixed<coord_type, df, df> B(0.0);
    for (unsigned d = 0; d < dm * dm; ++d) B(d, d) = rf / rm;
    for (unsigned d = dm * dm; d < df; ++d) B(d, d) = rf;
    this->covar_ = B * this->covar_ * B;
    this->center_moving_ = cm;
  }
  virtual mat &get_A() { return A_; }
  virtual vec &get_translation() { return t_; }
  cdcl_type_macro(cdcl_trans_affine);

 private:
  mat A_;
  vec t_;
};

#endif  // LRR_AMD_USE_REFERENCE_TYPES
#endif
