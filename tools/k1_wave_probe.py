"""Wave-level trip counts of the thread-per-query ball search in the moving-transform regime (a build with -DLRR_PROFILE_K1):
row groups, candidate batches and exact comparisons a WAVEFRONT executes (set by its slowest lane), beside the per-lane averages."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from lrr_amd import estimation as E, synth  # noqa: E402

lib = E.hip_lib()
E.set_device(0)
n = 200000
pair = synth.make_pair(n, n, 20244, kind="affine")
T0 = synth.perturbed(pair.planted, 3)
est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c, moving_inside=pair.moving, fixed_inside=pair.fixed, seed=7)
out = (C.c_ulonglong * 32)()
lib.lrr_icp_debug_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for r in range(2):
    Tp = synth.perturbed(pair.planted, 500 + r)
    est.set_transform(Tp.A, Tp.t, Tp.c)
    est.set_number_points(10**9)
    est.one_iteration(0)          # first search (wide balls) not counted
    lib.lrr_icp_debug_prof(out, 1)
    its = 0
    for it in range(1, 6):
        est.one_iteration(it)
        its += 1
    lib.lrr_icp_debug_prof(out, 0)
    v = np.array(list(out), dtype=np.float64)
    waves = max(v[24], 1)
    print(f"run {r}: {its} iterations, searching wavefronts {int(v[24])}: per wavefront row groups {v[25] / waves:.2f}, candidate batches {v[26] / waves:.2f}, "
          f"exact comparisons executed {v[27] / waves:.2f} with {v[28] / max(v[27], 1):.1f} lanes each; per lane rows {v[7] / max(v[24] * 64, 1):.2f} candidates {v[8] / max(v[24] * 64, 1):.1f}")
