"""Does a second batch in flight fill the tail of the lockstep loop?  256 query locations (config 5) as ONE batch, and as
S interleaved part-batches on S host threads (each its own device context = its own stream and its own copy of the two
volumes) that run concurrently on the same GPU.  Prints seconds per call (median of 5) and locations/s.
usage: python tools/locations_streams_probe.py [n=200000] [locations=256]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lrr_amd import estimation as E, locations, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    nl = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    E.hip_lib()
    E.set_device(0)
    pair = synth.make_pair(n, n, 20245, kind="affine")
    rois = synth.query_rois(pair, nl, 20245, half=25.0)
    inits = [synth.perturbed(pair.planted, 1000 + i) for i in range(len(rois))]
    ref = None
    for S in (1, 2, 3, 4):
        batches = [None] * S

        def work(k):
            E.set_device(0)
            if batches[k] is None:
                batches[k] = E.LocationBatch(pair.moving, pair.fixed)
            # part k of S (contiguous blocks, as the ranks of a multi-GPU run take them): a location keeps its index and with it
            # the seed of its subset permutations, so the rows can be compared with the one-batch rows
            idx = list(locations.multi_gpu.shard(len(rois), k, S))
            mb, fb = locations.roi_boxes(rois)
            A0 = np.stack([inits[i].A for i in idx]); t0 = np.stack([inits[i].t for i in idx]); c0 = np.stack([inits[i].c for i in idx])
            seeds = np.array([locations.location_seed(7, i) for i in idx], dtype=np.uint64)
            rows = batches[k].run(A0, t0, c0, mb[idx], fb[idx], seeds, max_iterations=locations.MAX_ITERATIONS)
            rows[:, 19] = idx
            return rows, int(rows[:, E.LocationBatch.COL["matched_pairs"]].sum())

        def call():
            futs = [locations._worker_thread(k).submit(work, k) for k in range(S)]
            parts = [f.result() for f in futs]
            rows = np.concatenate([p[0] for p in parts])
            return rows[np.argsort(rows[:, 19])], sum(p[1] for p in parts)

        call()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            rows, pairs = call()
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        if ref is None:
            ref = rows
        same = bool(np.array_equal(np.delete(ref, 19, axis=1), np.delete(rows, 19, axis=1), equal_nan=True))
        print(f"part-batches {S}: {1e3 * t:.2f} ms per {len(rois)} locations = {len(rois) / t:.0f} locations/s, {pairs / t:.3e} pairs/s, rows equal the one-batch rows: {same}", flush=True)


if __name__ == "__main__":
    main()
