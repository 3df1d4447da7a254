// Reads matrices from a file (int32 count; then per matrix int32 n and n*n doubles), writes
// sym_inverse of each to stdout as text.  Driven by tests/test_linalg_cpu.py.
#include <cstdio>
#include <vector>
#include "lrr_amd/linalg.h"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int count = 0;
  if (std::fread(&count, sizeof count, 1, f) != 1) return 2;
  for (int m = 0; m < count; ++m) {
    int n = 0;
    if (std::fread(&n, sizeof n, 1, f) != 1 || n < 1 || n > 12) return 2;
    std::vector<double> A((size_t)n * n), out((size_t)n * n, 0.0);
    if (std::fread(A.data(), sizeof(double), A.size(), f) != A.size()) return 2;
    bool chol = lrr_amd::linalg::chol_inverse(n, A.data(), out.data());
    lrr_amd::linalg::sym_inverse(n, A.data(), out.data());
    std::printf("%d %d", n, chol ? 1 : 0);
    for (double v : out) std::printf(" %.17g", v);
    std::printf("\n");
  }
  std::fclose(f);
  return 0;
}
