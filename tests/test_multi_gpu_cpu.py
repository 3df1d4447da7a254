"""The N > 1 path on CPU: two gloo ranks shard the query locations, run their blocks with no
data-path collective, and ONE all_gather delivers every location's result to every rank
(lrr_amd/multi_gpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lrr_amd import multi_gpu


def test_shard_is_a_partition():
    for n, world in ((256, 8), (10, 3), (2, 4), (0, 2), (7, 1)):
        seen = []
        for r in range(world):
            seen += list(multi_gpu.shard(n, r, world))
        assert seen == list(range(n))
        sizes = [len(multi_gpu.shard(n, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    assert len(multi_gpu.shard(256, 3, 8)) == 32  # config 5: 32 locations per GPU


WIDE = 343  # row width of the batched engine (estimation.LocationBatch): the gather takes any width


def _fake_estimate(i, width=multi_gpu.RESULT_WIDTH):
    # deterministic stand-in for one location's symmetric ICP estimate
    rng = np.random.default_rng(i)
    row = multi_gpu.pack_result(i, i % 3 != 0, 10 + i, rng.normal(size=(3, 3)), rng.normal(size=3),
                                rng.normal(size=3), 0.5 + i, 1.5 + i)
    if width > row.shape[0]:
        row = np.concatenate([row, rng.normal(size=width - row.shape[0])])
    return row


def _worker(rank, world, port, n_locations, out_dir, width=multi_gpu.RESULT_WIDTH):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = multi_gpu.run_locations(lambda i: _fake_estimate(i, width), n_locations, rank, world)
        assert local.shape[0] == len(multi_gpu.shard(n_locations, rank, world))
        rows = multi_gpu.gather_results(local, n_locations)
        np.save(os.path.join(out_dir, f"rows_{rank}.npy"), rows)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_locations", [7, 8])
def test_two_rank_gloo_gather(tmp_path, n_locations):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_locations, str(tmp_path)), nprocs=world, join=True)
    want = np.stack([_fake_estimate(i) for i in range(n_locations)])
    for r in range(world):
        rows = np.load(tmp_path / f"rows_{r}.npy")
        assert rows.shape == want.shape
        assert np.array_equal(rows, want)  # every rank holds every location, ordered by index


def test_two_rank_gloo_gather_of_wide_rows(tmp_path):
    """the result row of the batched engine (covariances, angles, decision measurements: 343 doubles)"""
    world, n_locations = 2, 9
    mp.spawn(_worker, args=(world, _free_port(), n_locations, str(tmp_path), WIDE), nprocs=world, join=True)
    want = np.stack([_fake_estimate(i, WIDE) for i in range(n_locations)])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"rows_{r}.npy"), want)


def test_single_process_gather_is_identity():
    local = np.stack([_fake_estimate(i) for i in range(3)])
    assert multi_gpu.gather_results(local, 3) is local
