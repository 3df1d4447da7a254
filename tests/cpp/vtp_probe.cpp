// vtp_probe.cpp — reads one feature file with include/lrr_amd/vtp_features.h and prints "ok <n>" or "error <message>".
// tests/test_vtp_cpu.py builds it with -fsanitize=address,undefined (host code only) and feeds it truncated and
// corrupted files: a damaged file must end in the reader's runtime_error, never in an out-of-bounds access.
#include <cstdio>
#include <exception>

#include "lrr_amd/vtp_features.h"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  try {
    lrr_amd::vtp_arrays a;
    lrr_amd::read_vtp(argv[1], a);
    std::printf("ok %zu\n", a.n);
  } catch (const std::exception &e) {
    std::printf("error %s\n", e.what());
  }
  return 0;
}
