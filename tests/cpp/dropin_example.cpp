// dropin_example.cpp — the C++ drop-in classes used exactly as location_registration uses the
// reference classes (rrl/itkLocationRegistration.cxx:1635-1643, 2418-2446): construct
// rrl_estimation_symmetric_ICP_matching_all<3,12> from feature vectors, call one_iteration() until
// it converges, read the transforms.  Reads a small binary problem file written by the test and
// prints the result as text.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lrr_amd/cdcl_estimation_ICP_matching_all.h"
#include "lrr_amd/cdcl_estimation_symmetric_ICP_matching_all.h"
#include "lrr_amd/rrl_estimation_ICP_matching_all.h"
#include "lrr_amd/rrl_estimation_symmetric_ICP_matching_all.h"

// every member of every drop-in class compiles (the instantiations the reference ships:
// cdcl/cdcl_estimation_ICP_matching_all+3+6-.cxx etc.)
template class cdcl_estimation_ICP_matching_all<3, 6>;
template class cdcl_estimation_ICP_matching_all<3, 12>;
template class rrl_estimation_ICP_matching_all<3, 12>;
template class cdcl_estimation_symmetric_ICP_matching_all<3, 6>;
template class cdcl_estimation_symmetric_ICP_matching_all<3, 12>;
template class rrl_estimation_symmetric_ICP_matching_all<3, 12>;

typedef cdcl_feature_ICP<3> feature_type;
typedef feature_type::sptr feature_sptr_type;
typedef rrl_estimation_symmetric_ICP_matching_all<3, 12> estimation_symmetric_type;

static std::vector<feature_sptr_type> read_features(FILE *f) {
  long long n = 0;
  if (fread(&n, sizeof n, 1, f) != 1) std::exit(2);
  std::vector<feature_sptr_type> out;
  for (long long i = 0; i < n; ++i) {
    double rec[3 + 9 + 1 + 1 + 6];
    if (fread(rec, sizeof(double), 20, f) != 20) std::exit(2);
    vnl_vector_fixed<double, 3> loc(rec[0], rec[1], rec[2]);
    vnl_matrix_fixed<double, 3, 3> P;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) P(r, c) = rec[3 + 3 * r + c];
    std::vector<vnl_vector_fixed<double, 3> > dirs;
    dirs.push_back(vnl_vector_fixed<double, 3>(rec[14], rec[15], rec[16]));
    dirs.push_back(vnl_vector_fixed<double, 3>(rec[17], rec[18], rec[19]));
    out.push_back(new feature_type(loc, rec[12], P, (feature_type::shape_type)(int)rec[13], dirs));
  }
  return out;
}

static int n_callbacks = 0;
static void display_points(void *, std::vector<feature_sptr_type> const &, std::vector<feature_sptr_type> const &,
                           std::vector<estimation_symmetric_type::match_sptr_type> const &matches,
                           estimation_symmetric_type::trans_sptr_type const &, unsigned int) {
  if (!matches.empty() && matches[0]->to_.size() == 1 && matches[0]->w_.size() == 1) ++n_callbacks;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  double T[15];
  if (fread(T, sizeof(double), 15, f) != 15) return 2;
  std::vector<feature_sptr_type> moving = read_features(f), fixed = read_features(f);
  fclose(f);
  vnl_matrix_fixed<double, 3, 3> A;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A(r, c) = T[3 * r + c];
  vnl_vector_fixed<double, 3> t(T[9], T[10], T[11]), c(T[12], T[13], T[14]);
  cdcl_trans<3, 12>::sptr transform = new cdcl_trans_affine<3, 12>(A, t, c);

  // as SetupEstimation does (the whole volumes double as the ROI sets here)
  estimation_symmetric_type::sptr estimation =
      new estimation_symmetric_type(moving, fixed, transform, transform->inverse(), moving, fixed);
  estimation->weight_by_strength(true);
  estimation->normalize_matches();
  estimation->initialize();

  unsigned int iteration = 0;
  bool converged = false;
  int caller_token = 0;
  while (!converged && iteration < 70) {
    converged = estimation->one_iteration(iteration == 0 ? (void *)&caller_token : 0, &display_points, iteration);
    ++iteration;
  }
  cdcl_trans<3, 12>::sptr fwd = estimation->get_transform(), bwd = estimation->get_transform_backward();
  printf("converged %d iterations %u callbacks %d\n", converged ? 1 : 0, iteration, n_callbacks);
  printf("A");
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) printf(" %.17g", fwd->get_A()(r, k));
  printf("\nt %.17g %.17g %.17g\n", fwd->get_translation()[0], fwd->get_translation()[1], fwd->get_translation()[2]);
  printf("c %.17g %.17g %.17g\n", fwd->center_moving_[0], fwd->center_moving_[1], fwd->center_moving_[2]);
  printf("weighted_error %.17g backward %.17g rms %.17g oscillated %d\n", estimation->weighted_error(),
         estimation->weighted_error_backward(), estimation->RMS_error(), estimation->oscillated() ? 1 : 0);
  printf("tb %.17g %.17g %.17g\n", bwd->get_translation()[0], bwd->get_translation()[1], bwd->get_translation()[2]);
  return 0;
}
