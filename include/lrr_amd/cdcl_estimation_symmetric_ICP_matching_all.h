// cdcl_estimation_symmetric_ICP_matching_all.h — MI355X drop-in for the reference class of the same name (cdcl/cdcl_estimation_symmetric_ICP_matching_all.h):
// same template parameters, constructors and public methods (the class location_registration
// drives: rrl/itkLocationRegistration.cxx:1635-1643, 2303, 2422-2446, 1828-1845).
#ifndef LRR_AMD_cdcl_estimation_symmetric_ICP_matching_all_H_
#define LRR_AMD_cdcl_estimation_symmetric_ICP_matching_all_H_

#include "lrr_amd/icp_engine.h"

template <unsigned int dim, unsigned int dof>
class cdcl_estimation_symmetric_ICP_matching_all : public lrr_amd::icp_symmetric<dof, cdcl_feature_with_shape<dim>, false> {
  static_assert(dim == 3, "the MI355X path implements the 3-D instantiations (<3,6> and <3,12>)");

 public:
  typedef vbl_smart_ptr<cdcl_estimation_symmetric_ICP_matching_all> sptr;
  typedef lrr_amd::icp_symmetric<dof, cdcl_feature_with_shape<dim>, false> engine_type;
  typedef typename engine_type::feature_sptr_type feature_sptr_type;
  typedef typename engine_type::trans_sptr_type trans_sptr_type;
  typedef typename engine_type::location_type location_type;
  cdcl_estimation_symmetric_ICP_matching_all(vcl_vector<feature_sptr_type> const &moving, vcl_vector<feature_sptr_type> const &fixed,
     trans_sptr_type const &trans, trans_sptr_type const &trans_inverse,
     vcl_vector<feature_sptr_type> const &moving_inside, vcl_vector<feature_sptr_type> const &fixed_inside)
      : engine_type(moving, fixed, trans, trans_inverse, moving_inside, fixed_inside) {}
  cdcl_estimation_symmetric_ICP_matching_all(vcl_vector<feature_sptr_type> const &moving, vcl_vector<feature_sptr_type> const &fixed,
     trans_sptr_type const &trans, trans_sptr_type const &trans_inverse, location_type const &moving_x0,
     location_type const &moving_x1, location_type const &fixed_x0, location_type const &fixed_x1)
      : engine_type(moving, fixed, trans, trans_inverse, moving_x0, moving_x1, fixed_x0, fixed_x1) {}

  // The reference merges the backward list into the forward one as role-reversed clones before its
  // normal equations (cdcl sym .txx:366-394, called at :542).  On the device the merged list is never
  // built: the backward matches enter the sums as clones with their own weights (LRR_ICP_ACC_CLONE_W).
  // This host version, on caller-owned match vectors, is kept for code that calls it directly.
  typedef typename engine_type::estimation_type::match_type match_type;
  typedef typename match_type::sptr match_sptr_type;
  typedef typename match_type::feature_type feature_type;
  void add_constraints(vcl_vector<match_sptr_type> const &from_matches, vcl_vector<match_sptr_type> &to_matches) {
    for (size_t i = 0; i < from_matches.size(); ++i) {
      vcl_vector<feature_sptr_type> p_vec(1, feature_sptr_type(new feature_type(*from_matches[i]->from_)));
      for (size_t j = 0; j < from_matches[i]->to_.size(); ++j) {
        vcl_vector<double> w_vec(1, from_matches[i]->w_[j]);
        to_matches.push_back(new match_type(feature_sptr_type(new feature_type(*from_matches[i]->to_[j])), p_vec, w_vec));
      }
    }
  }
};
#endif
