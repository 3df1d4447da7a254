// vxl_compat.h — the handful of VXL value types the ICP estimator interface is written in.
//
// The reference's public interface (cdcl/cdcl_estimation_abs.h:20-65, cdcl/cdcl_feature*.h,
// cdcl/cdcl_match.h, cdcl/cdcl_trans.h) is expressed with vnl_vector_fixed, vnl_matrix_fixed,
// vnl_vector, vbl_ref_count and vbl_smart_ptr.  Inside the reference tree define
// LRR_AMD_USE_VXL and the real headers are used; stand-alone (this repo: VXL is not installed)
// the minimal look-alikes below provide exactly the members the drop-in classes touch.
#ifndef LRR_AMD_VXL_COMPAT_H_
#define LRR_AMD_VXL_COMPAT_H_

#ifdef LRR_AMD_USE_VXL
#include <vbl/vbl_ref_count.h>
#include <vbl/vbl_smart_ptr.h>
#include <vcl_iostream.h>
#include <vcl_vector.h>
#include <vnl/vnl_matrix_fixed.h>
#include <vnl/vnl_vector.h>
#include <vnl/vnl_vector_fixed.h>
#else

#include <cmath>
#include <cstddef>
#include <iostream>
#include <typeinfo>
#include <vector>

#define vcl_vector std::vector
#define vcl_ostream std::ostream
#define vcl_cout std::cout
#define vcl_endl std::endl
#define vcl_type_info std::type_info

template <class T, unsigned n>
class vnl_vector_fixed {
 public:
  vnl_vector_fixed() {}
  explicit vnl_vector_fixed(T v) { fill(v); }
  vnl_vector_fixed(T a, T b, T c) { static_assert(n == 3, "3-vector"); d_[0] = a; d_[1] = b; d_[2] = c; }
  T &operator[](unsigned i) { return d_[i]; }
  const T &operator[](unsigned i) const { return d_[i]; }
  T &operator()(unsigned i) { return d_[i]; }
  const T &operator()(unsigned i) const { return d_[i]; }
  void fill(T v) { for (unsigned i = 0; i < n; ++i) d_[i] = v; }
  unsigned size() const { return n; }
  T *data_block() { return d_; }
  const T *data_block() const { return d_; }
  T squared_magnitude() const { T s = 0; for (unsigned i = 0; i < n; ++i) s += d_[i] * d_[i]; return s; }
  T magnitude() const { return std::sqrt(squared_magnitude()); }
  vnl_vector_fixed &operator+=(const vnl_vector_fixed &o) { for (unsigned i = 0; i < n; ++i) d_[i] += o.d_[i]; return *this; }
  vnl_vector_fixed &operator-=(const vnl_vector_fixed &o) { for (unsigned i = 0; i < n; ++i) d_[i] -= o.d_[i]; return *this; }
  vnl_vector_fixed &operator*=(T s) { for (unsigned i = 0; i < n; ++i) d_[i] *= s; return *this; }
  vnl_vector_fixed &operator/=(T s) { for (unsigned i = 0; i < n; ++i) d_[i] /= s; return *this; }
  vnl_vector_fixed operator+(const vnl_vector_fixed &o) const { vnl_vector_fixed r(*this); r += o; return r; }
  vnl_vector_fixed operator-(const vnl_vector_fixed &o) const { vnl_vector_fixed r(*this); r -= o; return r; }
  vnl_vector_fixed operator-() const { vnl_vector_fixed r; for (unsigned i = 0; i < n; ++i) r.d_[i] = -d_[i]; return r; }
  vnl_vector_fixed operator*(T s) const { vnl_vector_fixed r(*this); r *= s; return r; }
  vnl_vector_fixed operator/(T s) const { vnl_vector_fixed r(*this); r /= s; return r; }

 private:
  T d_[n];
};
template <class T, unsigned n>
inline vnl_vector_fixed<T, n> operator*(T s, const vnl_vector_fixed<T, n> &v) { return v * s; }
template <class T, unsigned n>
inline T dot_product(const vnl_vector_fixed<T, n> &a, const vnl_vector_fixed<T, n> &b) {
  T s = 0;
  for (unsigned i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

template <class T, unsigned R, unsigned C>
class vnl_matrix_fixed {
 public:
  vnl_matrix_fixed() {}
  explicit vnl_matrix_fixed(T v) { fill(v); }
  T &operator()(unsigned r, unsigned c) { return d_[r][c]; }
  const T &operator()(unsigned r, unsigned c) const { return d_[r][c]; }
  T *operator[](unsigned r) { return d_[r]; }
  const T *operator[](unsigned r) const { return d_[r]; }
  void fill(T v) { for (unsigned r = 0; r < R; ++r) for (unsigned c = 0; c < C; ++c) d_[r][c] = v; }
  void set_identity() { fill(0); for (unsigned i = 0; i < (R < C ? R : C); ++i) d_[i][i] = 1; }
  unsigned rows() const { return R; }
  unsigned cols() const { return C; }
  T *data_block() { return &d_[0][0]; }
  const T *data_block() const { return &d_[0][0]; }
  vnl_matrix_fixed<T, C, R> transpose() const {
    vnl_matrix_fixed<T, C, R> t;
    for (unsigned r = 0; r < R; ++r) for (unsigned c = 0; c < C; ++c) t(c, r) = d_[r][c];
    return t;
  }
  vnl_vector_fixed<T, R> get_column(unsigned c) const { vnl_vector_fixed<T, R> v; for (unsigned r = 0; r < R; ++r) v[r] = d_[r][c]; return v; }
  vnl_matrix_fixed &operator*=(T s) { for (unsigned r = 0; r < R; ++r) for (unsigned c = 0; c < C; ++c) d_[r][c] *= s; return *this; }
  vnl_matrix_fixed &operator/=(T s) { for (unsigned r = 0; r < R; ++r) for (unsigned c = 0; c < C; ++c) d_[r][c] /= s; return *this; }
  vnl_matrix_fixed &operator+=(const vnl_matrix_fixed &o) { for (unsigned r = 0; r < R; ++r) for (unsigned c = 0; c < C; ++c) d_[r][c] += o.d_[r][c]; return *this; }
  vnl_matrix_fixed operator*(T s) const { vnl_matrix_fixed m(*this); m *= s; return m; }
  vnl_vector_fixed<T, R> operator*(const vnl_vector_fixed<T, C> &v) const {
    vnl_vector_fixed<T, R> o;
    for (unsigned r = 0; r < R; ++r) { T s = 0; for (unsigned c = 0; c < C; ++c) s += d_[r][c] * v[c]; o[r] = s; }
    return o;
  }
  template <unsigned K>
  vnl_matrix_fixed<T, R, K> operator*(const vnl_matrix_fixed<T, C, K> &b) const {
    vnl_matrix_fixed<T, R, K> o;
    for (unsigned r = 0; r < R; ++r) for (unsigned k = 0; k < K; ++k) { T s = 0; for (unsigned c = 0; c < C; ++c) s += d_[r][c] * b(c, k); o(r, k) = s; }
    return o;
  }

 private:
  T d_[R][C];
};
template <class T, unsigned R, unsigned C>
inline vnl_matrix_fixed<T, R, C> operator*(T s, const vnl_matrix_fixed<T, R, C> &m) { return m * s; }

template <class T>
class vnl_vector {
 public:
  vnl_vector() {}
  explicit vnl_vector(unsigned n) : d_(n) {}
  vnl_vector(unsigned n, T v) : d_(n, v) {}
  unsigned size() const { return (unsigned)d_.size(); }
  T &operator[](unsigned i) { return d_[i]; }
  const T &operator[](unsigned i) const { return d_[i]; }
  T *data_block() { return d_.data(); }
  const T *data_block() const { return d_.data(); }

 private:
  std::vector<T> d_;
};

class vbl_ref_count {
 public:
  vbl_ref_count() : ref_count_(0) {}
  vbl_ref_count(const vbl_ref_count &) : ref_count_(0) {}
  vbl_ref_count &operator=(const vbl_ref_count &) { return *this; }
  virtual ~vbl_ref_count() {}
  void ref() { ++ref_count_; }
  void unref() { if (--ref_count_ <= 0) delete this; }
  int get_references() const { return ref_count_; }

 private:
  int ref_count_;
};

template <class T>
class vbl_smart_ptr {
 public:
  vbl_smart_ptr() : p_(nullptr) {}
  vbl_smart_ptr(T *p) : p_(p) { if (p_) p_->ref(); }
  vbl_smart_ptr(const vbl_smart_ptr &o) : p_(o.p_) { if (p_) p_->ref(); }
  ~vbl_smart_ptr() { if (p_) p_->unref(); }
  vbl_smart_ptr &operator=(const vbl_smart_ptr &o) { return *this = o.p_; }
  vbl_smart_ptr &operator=(T *p) {
    if (p) p->ref();
    T *old = p_;
    p_ = p;
    if (old) old->unref();
    return *this;
  }
  T *operator->() const { return p_; }
  T &operator*() const { return *p_; }
  T *as_pointer() const { return p_; }
  T *ptr() const { return p_; }
  explicit operator bool() const { return p_ != nullptr; }
  bool operator!() const { return p_ == nullptr; }
  bool operator==(const vbl_smart_ptr &o) const { return p_ == o.p_; }
  bool operator!=(const vbl_smart_ptr &o) const { return p_ != o.p_; }
  bool operator<(const vbl_smart_ptr &o) const { return p_ < o.p_; }

 private:
  T *p_;
};

#endif  // LRR_AMD_USE_VXL
#endif
