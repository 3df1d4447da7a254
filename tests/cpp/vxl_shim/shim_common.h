// test-only: the VXL names the reference's cdcl_* type headers use, mapped onto include/lrr_amd/vxl_compat.h and the
// standard library, so that the drop-in classes can be SYNTAX-checked (g++ -fsyntax-only) against the reference's
// own headers (tests/test_reference_headers.py).  Nothing is built or run from it.
#ifndef LRR_TEST_VXL_SHIM_COMMON_H_
#define LRR_TEST_VXL_SHIM_COMMON_H_
#include <cstdlib>
#include <iostream>
#include <istream>

#include "lrr_amd/vxl_compat.h"
#define vcl_istream std::istream
#define vcl_cerr std::cerr
#define vcl_abort std::abort
#endif
