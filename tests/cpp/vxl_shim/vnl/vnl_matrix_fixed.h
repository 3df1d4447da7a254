// test-only forwarding header (see shim_common.h)
#include "shim_common.h"
