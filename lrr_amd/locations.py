"""Config 5: many independent query locations against one volume pair, sharded over GPUs.

Mirrors the outer loop of ``rrl/itkLocationRegistration.cxx:2134-2138`` for the estimation part
only: for every query location the features inside the 50 mm ROIs become the active sets of a
``rrl_estimation_symmetric_ICP_matching_all<3,12>`` estimate (``set_features`` + ``set_transform``,
``itkLocationRegistration.cxx:1635-1637``), iterated until it reports convergence or 70 iterations
(``:2418``), followed by the caller's decision measurements (``:1816-1871``).  Ranks take static
blocks of the locations (``multi_gpu.shard``) and one all_gather returns every location's result
row to every rank.

``register_locations`` runs a rank's block through the batched engine
(``include/lrr_amd/location_batch.h``: both volumes resident, ROI selection on the device, every
device phase of an iteration one submission for all locations of the block);
``register_locations_sequential`` is the one-estimator-object-at-a-time form the reference's caller
has, kept as the comparison the tests and the bench quote.
"""
from __future__ import annotations

import numpy as np

from . import estimation as E
from . import multi_gpu, synth

MAX_ITERATIONS = 70  # rrl/itkLocationRegistration.cxx:2413-2418


def location_seed(seed: int, index: int) -> int:
    return (int(seed) * 1000003 + int(index) * 7919 + 1) & 0xFFFFFFFFFFFFFFFF


def roi_boxes(rois, half=25.0):
    """(B, 6) boxes [x0, x1] of the moving and the fixed ROI of every query (synth.query_rois): the cube of
    half-width `half` around the query centre (rrl/itkLocationRegistration.cxx:1531-1549)"""
    cf = np.array([r[0] for r in rois], dtype=np.float64).reshape(-1, 3)
    cm = np.array([r[1] for r in rois], dtype=np.float64).reshape(-1, 3)
    return np.concatenate([cm - half, cm + half], axis=1), np.concatenate([cf - half, cf + half], axis=1)


def register_locations(pair: synth.VolumePair, rois, inits, rank: int = 0, world: int = 1, seed: int = 7,
                       gather_device=None, half: float = 25.0, batch=None, diagnostics=True, max_batch: int = 1024,
                       check_valid: bool = True):
    """Batched form.  rois: synth.query_rois output (only the centres are used: the ROI lists are selected
    on the device); inits: the initial transform of every location; batch: an estimation.LocationBatch
    that already holds the two volumes (kept by the caller across calls), else one is built and closed.
    Returns (rows of LocationBatch.width doubles for ALL locations, ordered by location index, matched
    pairs processed by this rank)."""
    mine = list(multi_gpu.shard(len(rois), rank, world))
    own = batch is None
    if own:
        batch = E.LocationBatch(pair.moving, pair.fixed)
    width = batch.width
    mbox, fbox = roi_boxes(rois, half)
    rows = []
    for k0 in range(0, len(mine), max_batch):
        idx = np.array(mine[k0:k0 + max_batch], dtype=np.int64)
        A0 = np.stack([inits[i].A for i in idx]); t0 = np.stack([inits[i].t for i in idx]); c0 = np.stack([inits[i].c for i in idx])
        # the subset permutations of a location depend only on (seed, location): results do not depend on
        # which rank or batch handles it
        seeds = np.array([location_seed(seed, int(i)) for i in idx], dtype=np.uint64)
        # check_valid: the caller's `if (!validTransform(...)) break;` (rrl/itkLocationRegistration.cxx:2482)
        r = batch.run(A0, t0, c0, mbox[idx], fbox[idx], seeds, max_iterations=MAX_ITERATIONS, diagnostics=diagnostics,
                      check_valid=check_valid)
        r[:, 19] = idx
        rows.append(r)
    if own:
        batch.close()
    local = np.concatenate(rows) if rows else np.zeros((0, width))
    pairs = int(local[:, E.LocationBatch.COL["matched_pairs"]].sum()) if local.shape[0] else 0
    if world == 1:  # also a single-rank call made inside a multi-rank job (bench warm-up): no collective
        return local, pairs
    return multi_gpu.gather_results(local, len(rois), device=gather_device), pairs


def _run_block(pair, rois, inits, indices, seed, est=None, diagnostics=True, check_valid=True):
    """One estimator object re-used over `indices` (a host thread = one device context).  `est`: an
    estimator that already holds the two volumes (uploaded by an earlier block); returns it for the
    next one."""
    rows, pairs = [], 0
    for i in indices:
        _, _, mi, fi = rois[i]
        m_in, f_in = pair.moving.subset(mi), pair.fixed.subset(fi)
        T0 = inits[i]
        if est is None:
            est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c,
                              moving_inside=m_in, fixed_inside=f_in, seed=seed)
        else:
            est.set_features(m_in, f_in)
            est.set_transform(T0.A, T0.t, T0.c)
        # the subset permutations of a location depend only on (seed, location): results do not
        # depend on which rank / thread / estimator object handles it
        est.set_subset_seed(location_seed(seed, i))
        converged, it, p = est.run_counted(MAX_ITERATIONS, check_valid=check_valid)  # the caller's loop, in C++ (no GIL held)
        pairs += p
        A, t, c, _ = est.get_transform()
        row = multi_gpu.pack_result(i, converged, it, A, t, c, est.weighted_error(), est.RMS_error())
        if diagnostics:
            # what the caller asks of the estimator after convergence for its verifier
            # (rrl/itkLocationRegistration.cxx:1828-1845): angles and errors of both directions, the
            # backward estimate on the final matches (which re-centres the forward transform: the
            # result row above is taken before it, like the caller's own copy of the transform)
            est.sheet_angles(0); est.tube_angles(0)
            est.estimate_LS_backward(False)
            est.sheet_angles(1); est.tube_angles(1)
            est.weighted_error(1)
        rows.append(row)
    return rows, pairs, est


_WORKER_THREADS = []


def _worker_thread(k: int):
    """The k-th persistent single-thread executor of the sequential driver."""
    from concurrent.futures import ThreadPoolExecutor
    while len(_WORKER_THREADS) <= k:
        _WORKER_THREADS.append(ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"lrr-loc-{len(_WORKER_THREADS)}"))
    return _WORKER_THREADS[k]


def register_locations_sequential(pair: synth.VolumePair, rois, inits, rank: int = 0, world: int = 1,
                                  seed: int = 7, gather_device=None, workers: int = 1, device_index=None,
                                  estimators=None, diagnostics=True):
    """One estimator object re-used over the locations, as the reference's caller does.
    rois: list of (centre_fixed, centre_moving, moving_idx, fixed_idx) (synth.query_rois);
    inits: list of synth.Transform, the initial transform of every location.
    workers > 1: the rank's block is dealt round-robin to that many host threads, each with its own
    device context and estimator.  Every location is estimated by exactly one estimator started from
    its own initial transform: results do not depend on `workers`.
    diagnostics: also run the post-convergence calls the reference's caller makes per location.
    estimators: a list (one slot per worker, None = not built yet) that keeps the estimator objects —
    and with them the two uploaded volumes — alive between calls; without it they are built and
    closed inside the call.
    Returns (rows of multi_gpu.RESULT_WIDTH for ALL locations, matched pairs processed by this rank)."""
    mine = list(multi_gpu.shard(len(rois), rank, world))
    keep = estimators is not None
    if estimators is None:
        estimators = [None] * max(1, workers)
    while len(estimators) < max(1, workers):
        estimators.append(None)
    if workers <= 1 or len(mine) <= 1:
        rows, pairs, estimators[0] = _run_block(pair, rois, inits, mine, seed, estimators[0], diagnostics)
    else:
        def work(k):
            if device_index is not None:
                E.set_device(device_index)  # contexts are per host thread
            return _run_block(pair, rois, inits, mine[k::workers], seed, estimators[k], diagnostics)

        # estimator k lives on worker thread k for as long as the process runs: a device context belongs to the host
        # thread that made it, so the thread must outlive the call that keeps its estimator (`estimators`)
        futures = [_worker_thread(k).submit(work, k) for k in range(workers)]
        parts = [f.result() for f in futures]
        for k, part in enumerate(parts):
            estimators[k] = part[2]
        rows = [r for p in parts for r in p[0]]
        rows.sort(key=lambda r: r[19])
        pairs = sum(p[1] for p in parts)
    if not keep:
        for e in estimators:
            if e is not None:
                e.close()
    local = np.stack(rows) if rows else np.zeros((0, multi_gpu.RESULT_WIDTH))
    if world == 1:  # also a single-rank call made inside a multi-rank job (bench warm-up): no collective
        return local, pairs
    return multi_gpu.gather_results(local, len(rois), device=gather_device), pairs
