"""Phase cycle counts of the thread-per-query K1 of the batch (a build with -DLRR_PROFILE_K1: tools/k1_phase_probe.sh)."""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from lrr_amd import estimation as E, locations, synth  # noqa: E402

lib = E.hip_lib()
E.set_device(0)
pair = synth.make_pair(200000, 200000, 20245, kind="affine")
rois = synth.query_rois(pair, 256, 20245, half=25.0)
inits = [synth.perturbed(pair.planted, 100 + i) for i in range(len(rois))]
batch = E.LocationBatch(pair.moving, pair.fixed)
locations.register_locations(pair, rois[:4], inits, 0, 1, batch=batch)
out = (C.c_ulonglong * 32)()
lib.lrr_icp_debug_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.lrr_icp_debug_prof(out, 1)
locations.register_locations(pair, rois, inits, 0, 1, batch=batch)
lib.lrr_icp_debug_prof(out, 0)
v = np.array(list(out), dtype=np.float64)
names = ["inactive: loads+perm", "active: loads+perm", "active: search", "-", "search: bound/reseed/ranges", "search: row bounds", "search: candidates",
         "rows (sum)", "candidates (sum)", "inactive lanes", "active lanes"]
print(json.dumps({n: v[i] for i, n in enumerate(names)}, indent=1))
print("per active lane: loads+perm %.0f  search %.0f (bound %.0f rows %.0f cand %.0f) cycles; rows %.1f candidates %.1f" % (
    v[1] / v[10], v[2] / v[10], v[4] / v[10], v[5] / v[10], v[6] / v[10], v[7] / v[10], v[8] / v[10]))
print("per inactive lane: %.0f cycles" % (v[0] / max(v[9], 1)))
print("residual pass, per wavefront: loop %.0f  block sum %.0f cycles (%d wavefronts)" % (v[21] / max(v[23], 1), v[22] / max(v[23], 1), int(v[23])))
print("ROI fill, per workgroup: zero %.0f  mark %.0f  prefix %.0f  emit %.0f cycles (%d workgroups)" % (v[16] / max(v[20], 1), v[17] / max(v[20], 1), v[18] / max(v[20], 1), v[19] / max(v[20], 1), int(v[20])))
