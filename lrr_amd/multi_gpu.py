"""Independent query locations sharded over the GPUs of one node.

The path shards at the granularity of query locations (the outer loop of
``rrl/itkLocationRegistration.cxx:2134-2138``): every location is a self-contained symmetric
ICP estimate against the same two volumes.  Each rank (one process per GPU) takes a static
block of the locations, runs them with no inter-GPU traffic, and ONE collective at the end
gathers the results (one row per location: converged flag, iteration count, 12 parameters + centre,
12 x 12 covariances of both directions, final errors, angles and the verifier's 8 decision
measurements: 2.7 KB per location, 0.7 MB for 256 locations — latency-bound on xGMI).  ``torch.distributed`` is plumbing: backend
"nccl" is RCCL on ROCm, "gloo" on CPU for the tests.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np

RESULT_WIDTH = 20  # converged, iterations, A(9), t(3), c(3), weighted_error, rms, location index: the leading columns of every row layout


def shard(n_items: int, rank: int, world: int) -> range:
    """Static block partition: ranks get contiguous, near-equal blocks (32 per GPU for 256/8)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def pack_result(index, converged, iterations, A, t, c, weighted_error, rms) -> np.ndarray:
    r = np.empty(RESULT_WIDTH, dtype=np.float64)
    r[0], r[1] = float(converged), float(iterations)
    r[2:11], r[11:14], r[14:17] = np.asarray(A).ravel(), t, c
    r[17], r[18], r[19] = weighted_error, rms, float(index)
    return r


def run_locations(estimate_one: Callable[[int], np.ndarray], n_locations: int, rank: int, world: int) -> np.ndarray:
    """Runs this rank's block; estimate_one(i) returns a packed result row."""
    mine = shard(n_locations, rank, world)
    rows = [estimate_one(i) for i in mine]
    return np.stack(rows) if rows else np.zeros((0, RESULT_WIDTH))


def gather_results(local: np.ndarray, n_locations: int, device=None) -> np.ndarray:
    """The single end-of-job collective: all ranks end up with every location's row, ordered by
    location index.  Uses all_gather on a padded block (RCCL has no variable-size gather)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    cap = (n_locations + world - 1) // world
    width = local.shape[1]  # every rank packs the same row layout; column 19 is the location index
    buf = torch.full((cap, width), float("nan"), dtype=torch.float64, device=device)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.from_numpy(local).to(buf.device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    rows = torch.cat(out).cpu().numpy()
    rows = rows[~np.isnan(rows[:, 19])]
    return rows[np.argsort(rows[:, 19], kind="stable")]
