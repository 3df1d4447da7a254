// rrl_estimation_ICP_matching_all.h — MI355X drop-in for the reference class of the same name (rrl/rrl_estimation_ICP_matching_all.h):
// same template parameters, constructor and public methods; the per-match loops run on the GPU
// through the C-ABI of include/lrr_icp.h (see lrr_amd/icp_engine.h).
#ifndef LRR_AMD_rrl_estimation_ICP_matching_all_H_
#define LRR_AMD_rrl_estimation_ICP_matching_all_H_

#include "lrr_amd/icp_engine.h"

template <unsigned int dim, unsigned int dof>
class rrl_estimation_ICP_matching_all : public lrr_amd::icp_direction<dof, cdcl_feature_ICP<dim>, true> {
  static_assert(dim == 3, "the MI355X path implements the 3-D instantiations (<3,6> and <3,12>)");

 public:
  typedef vbl_smart_ptr<rrl_estimation_ICP_matching_all> sptr;
  typedef lrr_amd::icp_direction<dof, cdcl_feature_ICP<dim>, true> engine_type;
  typedef typename engine_type::feature_sptr_type feature_sptr_type;
  typedef typename engine_type::trans_sptr_type trans_sptr_type;
  rrl_estimation_ICP_matching_all(vcl_vector<feature_sptr_type> const &moving, vcl_vector<feature_sptr_type> const &fixed,
     trans_sptr_type const &trans)
      : engine_type(moving, fixed, trans) {}
  // (rrl/rrl_estimation_ICP_matching_all.h:101) the image is not read: the call selects the map's quantised lookup semantics
  // (LRR_ICP_NN_VORONOI); lrr_amd::voronoi_lookup::enable(false) keeps the exact search (see icp_symmetric::set_voronoi_maps)
  template <class VoronoiPointer> void set_fixed_voronoi_map(VoronoiPointer const &) {
    if (lrr_amd::voronoi_lookup::enabled()) this->set_nn_method(LRR_ICP_NN_VORONOI);
    else lrr_amd::voronoi_lookup::note_exact();
  }
};
#endif
