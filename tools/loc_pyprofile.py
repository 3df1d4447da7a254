"""Where the host time of register_locations goes (cProfile over 20 batched runs)."""
import cProfile
import pstats
import sys

sys.path.insert(0, ".")
from lrr_amd import estimation as E, locations, synth  # noqa: E402

E.hip_lib()
E.set_device(0)
pair = synth.make_pair(200000, 200000, 20245, kind="affine")
rois = synth.query_rois(pair, 256, 20245, half=25.0)
inits = [synth.perturbed(pair.planted, 100 + i) for i in range(len(rois))]
batch = E.LocationBatch(pair.moving, pair.fixed)
for _ in range(2):
    locations.register_locations(pair, rois, inits, 0, 1, batch=batch)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    locations.register_locations(pair, rois, inits, 0, 1, batch=batch)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
