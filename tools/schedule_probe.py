"""The faithful 1000 -> N schedule of the BASELINE workload, run `reps` times as ONE run() call each (what bench.py reports
as schedule.run_call_ms); under `rocprofv3 --kernel-trace` the per-dispatch series (tools/trace_series.py) shows what
every iteration of every level costs kernel by kernel.  Prints wall times and the per-iteration points of one stepped run."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lrr_amd import estimation as E  # noqa: E402
from lrr_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    E.set_device(0)
    pair = synth.make_pair(n, n, 20243, kind="affine")
    T0 = synth.perturbed(pair.planted, 3)
    timer = E.DeviceTimer()
    est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c, moving_inside=pair.moving, fixed_inside=pair.fixed, seed=7)
    out = {"n": n, "runs_ms": [], "iterations": [], "pairs": []}
    for r in range(reps + 1):
        est.set_transform(T0.A, T0.t, T0.c)
        est.set_subset_seed(7)
        timer.synchronize()
        t0 = time.perf_counter()
        conv, its, pairs = est.run_counted(70)
        timer.synchronize()
        dt = time.perf_counter() - t0
        if r:
            out["runs_ms"].append(1e3 * dt); out["iterations"].append(its); out["pairs"].append(pairs)
    # one stepped run: points per iteration
    est.set_transform(T0.A, T0.t, T0.c)
    est.set_subset_seed(7)
    pts, conv, it = [], False, 0
    while not conv and it < 70:
        pts.append(int(est.number_points(0)))
        conv = est.one_iteration(it)
        it += 1
    out["n_points"] = pts
    out["k1_paths"] = timer.k1_path_counts()
    out["median_ms"] = float(np.median(out["runs_ms"]))
    print(json.dumps(out))
    est.close()


if __name__ == "__main__":
    main()
