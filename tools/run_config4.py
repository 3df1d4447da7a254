#!/usr/bin/env python3
"""Config 4: the location_registration estimation stage on a feature-file pair, e.g. the reference's example
lung-CT pair (README.md:22; not in this image and not fetchable: supply it).

    python tools/run_config4.py DATADIR [--fixed 4893/4978whole] [--moving 4717/4722whole]
                                [--locations FILE] [--init FILE] [--roi 50] [--max-locations 100] [--out rows.npy]

Reads `<DATADIR>/<fixed>_00.vtk` and `<DATADIR>/<moving>_00.vtk` (VTK XML PolyData written by ExtractFeatures;
rrl/itkLocationRegistration.cxx:803-841) with the VTK-free reader, selects the ROI features of every query
location on the device (:1324-1360, >= 1000 each :1632) and runs rrl_estimation_symmetric_ICP_matching_all<3,12>
to convergence for all locations in one batch — matched with the lookup semantics of the Voronoi maps the production
binary uses (`--matcher exact`: the exact search) —, then the caller's decision measurements (:1816-1871).

Query locations: FILE with one `x y z` (physical coordinates of the fixed volume) per line, default
`<DATADIR>/<fixed>nodules.txt` (README.md:64) when it exists, else features sampled from the fixed volume.
Initial transforms: the reference derives them from keypoint-descriptor matches of its indexing stage
(:2179-2230), which is outside this path; here `--init FILE` gives 12 numbers per location (A row-major, t;
centre 0), default identity.  Prints one JSON line per location and a summary; "not run: data absent" when
the feature files are missing.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("datadir")
    ap.add_argument("--fixed", default="4893/4978whole")
    ap.add_argument("--moving", default="4717/4722whole")
    ap.add_argument("--locations")
    ap.add_argument("--init")
    ap.add_argument("--roi", type=float, default=50.0, help="ROI edge in mm (m_ROISize, 50)")
    ap.add_argument("--max-locations", type=int, default=100)  # :2134
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--out")
    ap.add_argument("--matcher", choices=["voronoi", "exact"], default="voronoi",
                    help="voronoi: the lookup semantics of the 1 mm Voronoi maps the production binary hands to every estimator "
                         "(set_voronoi_maps, rrl/itkLocationRegistration.cxx:1641-1643); exact: the exact nearest neighbour")
    a = ap.parse_args()
    ff, mf = os.path.join(a.datadir, a.fixed + "_00.vtk"), os.path.join(a.datadir, a.moving + "_00.vtk")
    if not (os.path.exists(ff) and os.path.exists(mf)):
        print(f"not run: data absent ({ff} / {mf})")
        return 0
    from lrr_amd import estimation as E
    from lrr_amd import locations, vtp
    fixed, moving = vtp.read_features(ff), vtp.read_features(mf)
    print(f"read {fixed.n} fixed and {moving.n} moving features", file=sys.stderr)
    lf = a.locations or os.path.join(a.datadir, a.fixed + "nodules.txt")
    if os.path.exists(lf):
        centres = np.loadtxt(lf, ndmin=2)[:, :3]
    else:
        centres = fixed.loc[np.random.default_rng(a.seed).choice(fixed.n, size=min(a.max_locations, fixed.n), replace=False)]
    centres = centres[: a.max_locations]
    B = centres.shape[0]
    if a.init:
        p = np.loadtxt(a.init, ndmin=2)
        if p.shape[0] == 1:
            p = np.repeat(p, B, axis=0)
        A0, t0 = p[:B, :9].reshape(B, 3, 3), p[:B, 9:12]
    else:
        A0, t0 = np.tile(np.eye(3), (B, 1, 1)), np.zeros((B, 3))
    c0 = np.zeros((B, 3))
    half = a.roi / 2.0
    # the moving ROI is centred on the query point mapped back by the initial transform (:1531-1549)
    cm = np.einsum("bij,bj->bi", np.linalg.inv(A0), centres - t0)
    fbox = np.concatenate([centres - half, centres + half], axis=1)
    mbox = np.concatenate([cm - half, cm + half], axis=1)
    seeds = np.array([locations.location_seed(a.seed, i) for i in range(B)], dtype=np.uint64)
    E.set_device(int(os.environ.get("LRR_ICP_DEVICE", "0")))
    bt = E.LocationBatch(moving, fixed)
    if a.matcher == "voronoi":
        bt.set_voronoi_maps(True)
    rows = bt.run(A0, t0, c0, mbox, fbox, seeds, max_iterations=locations.MAX_ITERATIONS, diagnostics=True, check_valid=True)
    C = bt.COL
    for b in range(B):
        r = rows[b]
        print(json.dumps({"location": b, "centre": centres[b].tolist(), "enough_features": bool(r[C["enough_features"]]),
                          "n_moving_inside": int(r[C["n_moving_inside"]]), "n_fixed_inside": int(r[C["n_fixed_inside"]]),
                          "converged": bool(r[C["converged"]]), "oscillated": bool(r[C["oscillated"]]),
                          "valid_transform": bool(r[C["valid"]]), "iterations": int(r[C["iterations"]]),
                          "A": r[C["A"]].tolist(), "t": r[C["t"]].tolist(), "c": r[C["c"]].tolist(),
                          "weighted_error": r[C["weighted_error"]], "measurements": r[C["measurements"]].tolist()}))
    ran = rows[:, C["enough_features"]] > 0
    print(f"{int(ran.sum())} of {B} locations had >= 1000 features in both ROIs; {int(rows[ran, 0].sum())} converged",
          file=sys.stderr)
    if a.out:
        np.save(a.out, rows)
    bt.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
