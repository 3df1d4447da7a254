// headers_check.cpp — every public header of include/lrr_amd in one translation unit, with the templates the
// callers use instantiated: compiled by tests/test_reference_headers.py stand-alone and (syntax only) against the
// reference's own cdcl_* type headers.
#include "lrr_amd/cdcl_estimation.h"
#include "lrr_amd/cdcl_estimation_ICP_matching_all.h"
#include "lrr_amd/cdcl_estimation_symmetric_ICP_matching_all.h"
#include "lrr_amd/location_batch.h"
#include "lrr_amd/rrl_estimation_ICP_matching_all.h"
#include "lrr_amd/rrl_estimation_symmetric_ICP_matching_all.h"
#include "lrr_amd/vtp_features.h"

template class lrr_amd::rrl_location_batch<cdcl_feature_ICP<3> >;
template std::vector<cdcl_feature_ICP<3>::sptr> lrr_amd::features_from_vtp<cdcl_feature_ICP<3> >(const lrr_amd::vtp_arrays &);

template class lrr_amd::cdcl_obj_fun_par<3, 12>;
template class lrr_amd::cdcl_obj_fun_cov<3, 12>;
template class lrr_amd::cdcl_lbfgs<lrr_amd::cdcl_obj_fun_par<3, 12> >;

int main() { return 0; }
