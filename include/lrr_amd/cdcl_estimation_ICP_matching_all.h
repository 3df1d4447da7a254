// cdcl_estimation_ICP_matching_all.h — MI355X drop-in for the reference class of the same name (cdcl/cdcl_estimation_ICP_matching_all.h:30-211):
// same template parameters, constructor and public methods; the per-match loops run on the GPU
// through the C-ABI of include/lrr_icp.h (see lrr_amd/icp_engine.h).
#ifndef LRR_AMD_cdcl_estimation_ICP_matching_all_H_
#define LRR_AMD_cdcl_estimation_ICP_matching_all_H_

#include "lrr_amd/icp_engine.h"

template <unsigned int dim, unsigned int dof>
class cdcl_estimation_ICP_matching_all : public lrr_amd::icp_direction<dof, cdcl_feature_with_shape<dim>, false> {
  static_assert(dim == 3, "the MI355X path implements the 3-D instantiations (<3,6> and <3,12>)");

 public:
  typedef vbl_smart_ptr<cdcl_estimation_ICP_matching_all> sptr;
  typedef lrr_amd::icp_direction<dof, cdcl_feature_with_shape<dim>, false> engine_type;
  typedef typename engine_type::feature_sptr_type feature_sptr_type;
  typedef typename engine_type::trans_sptr_type trans_sptr_type;
  cdcl_estimation_ICP_matching_all(vcl_vector<feature_sptr_type> const &moving, vcl_vector<feature_sptr_type> const &fixed,
     trans_sptr_type const &trans)
      : engine_type(moving, fixed, trans) {}

  // host utilities of the reference class that act on a caller's feature set, not on the matches
  // (cdcl/cdcl_estimation_ICP_matching_all.txx:1096-1155): zero-mean locations of average radius 1
  void cdcl_normalize_data(vcl_vector<feature_sptr_type> &feature_set, vnl_vector_fixed<double, dim> &center, double &avg_radius) {
    center_of(feature_set, center);
    avg_radius = 0.0;
    for (size_t i = 0; i < feature_set.size(); ++i) {
      feature_set[i]->location_ -= center;
      avg_radius += feature_set[i]->location_.magnitude();
    }
    avg_radius /= feature_set.size();
    rescale(feature_set, avg_radius);
  }
  void cdcl_normalize_data_known_radius(vcl_vector<feature_sptr_type> &feature_set, vnl_vector_fixed<double, dim> &center,
                                        double const &avg_radius) {
    center_of(feature_set, center);
    for (size_t i = 0; i < feature_set.size(); ++i) feature_set[i]->location_ -= center;
    rescale(feature_set, avg_radius);
  }

 private:
  static void center_of(vcl_vector<feature_sptr_type> const &feature_set, vnl_vector_fixed<double, dim> &center) {
    center.fill(0.0);
    for (size_t i = 0; i < feature_set.size(); ++i) center += feature_set[i]->location_;
    center /= feature_set.size();
  }
  static void rescale(vcl_vector<feature_sptr_type> &feature_set, double avg_radius) {
    for (size_t i = 0; i < feature_set.size(); ++i) {
      feature_set[i]->location_ /= avg_radius;
      feature_set[i]->covariance_ /= (avg_radius * avg_radius);
    }
  }
};
#endif
