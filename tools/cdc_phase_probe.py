"""Phase cycle counts of the CDC-EM E-step kernel (a build with -DLRR_PROFILE_K1: tools/cdc_phase_probe.sh)."""
import ctypes as C
import dataclasses
import sys

import numpy as np

sys.path.insert(0, ".")
from lrr_amd import estimation as E, synth  # noqa: E402

lib = E.hip_lib()
E.set_device(0)
n = 200000
pair = synth.make_pair(n, n, 20247, kind="affine")
mv = dataclasses.replace(pair.moving, cov=synth.covariances(pair.moving))
fx = dataclasses.replace(pair.fixed, cov=synth.covariances(pair.fixed))
T = synth.perturbed(pair.planted, 5, rot_deg=0.6, trans_mm=0.8)
B = np.random.default_rng(4).normal(size=(12, 12)) * 2e-3
B[9:] *= 50.0
Ct = B @ B.T
cm, cf = E.DeviceCloud(mv), E.DeviceCloud(fx)
cm.set_covariances(mv.cov)
cf.set_covariances(fx.cov)
S = E.CdcSet(cm, cf, max_matches=100)
S.compute_weights(12, T.A, T.t, T.c, Ct, seed=1)
out = (C.c_ulonglong * 32)()
lib.lrr_icp_debug_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.lrr_icp_debug_prof(out, 1)
S.compute_weights(12, T.A, T.t, T.c, Ct, seed=2)
lib.lrr_icp_debug_prof(out, 0)
v = np.array(list(out), dtype=np.float64)
tot = v[11] + v[12] + v[13] + v[14]
print("rows %d; per row: setup %.0f  ball rows + candidates %.0f  ranks / threshold %.0f  kept pairs %.0f cycles (%.0f%% %.0f%% %.0f%% %.0f%%); "
      "candidates in radius %.1f, cell rows %.1f" % (n, v[11] / n, v[12] / n, v[13] / n, v[14] / n, 100 * v[11] / tot, 100 * v[12] / tot,
                                                       100 * v[13] / tot, 100 * v[14] / tot, v[15] / n, v[3] / n))
