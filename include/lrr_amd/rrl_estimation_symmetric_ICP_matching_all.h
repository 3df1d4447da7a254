// rrl_estimation_symmetric_ICP_matching_all.h — MI355X drop-in for the reference class of the same name (rrl/rrl_estimation_symmetric_ICP_matching_all.h:40-246):
// same template parameters, constructors and public methods (the class location_registration
// drives: rrl/itkLocationRegistration.cxx:1635-1643, 2303, 2422-2446, 1828-1845).
#ifndef LRR_AMD_rrl_estimation_symmetric_ICP_matching_all_H_
#define LRR_AMD_rrl_estimation_symmetric_ICP_matching_all_H_

#include "lrr_amd/icp_engine.h"

template <unsigned int dim, unsigned int dof>
class rrl_estimation_symmetric_ICP_matching_all : public lrr_amd::icp_symmetric<dof, cdcl_feature_ICP<dim>, true> {
  static_assert(dim == 3, "the MI355X path implements the 3-D instantiations (<3,6> and <3,12>)");

 public:
  typedef vbl_smart_ptr<rrl_estimation_symmetric_ICP_matching_all> sptr;
  typedef lrr_amd::icp_symmetric<dof, cdcl_feature_ICP<dim>, true> engine_type;
  typedef typename engine_type::feature_sptr_type feature_sptr_type;
  typedef typename engine_type::trans_sptr_type trans_sptr_type;
  typedef typename engine_type::location_type location_type;
  rrl_estimation_symmetric_ICP_matching_all(vcl_vector<feature_sptr_type> const &moving, vcl_vector<feature_sptr_type> const &fixed,
     trans_sptr_type const &trans, trans_sptr_type const &trans_inverse,
     vcl_vector<feature_sptr_type> const &moving_inside, vcl_vector<feature_sptr_type> const &fixed_inside)
      : engine_type(moving, fixed, trans, trans_inverse, moving_inside, fixed_inside) {}
  rrl_estimation_symmetric_ICP_matching_all(vcl_vector<feature_sptr_type> const &moving, vcl_vector<feature_sptr_type> const &fixed,
     trans_sptr_type const &trans, trans_sptr_type const &trans_inverse, location_type const &moving_x0,
     location_type const &moving_x1, location_type const &fixed_x0, location_type const &fixed_x1)
      : engine_type(moving, fixed, trans, trans_inverse, moving_x0, moving_x1, fixed_x0, fixed_x1) {}
};
#endif
