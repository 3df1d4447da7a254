#!/usr/bin/env python3
"""Benchmark of the ICP hot path on MI355X — BASELINE.json's metric:
matched-feature-pairs/s (+ ms/iteration) on 200k x 200k synthetic 3-D features.

A "step" is ONE iteration of rrl_estimation_symmetric_ICP_matching_all<3,12>::one_iteration at
the finest level (every point of both ROIs matched: 200k forward + 200k backward pairs), i.e.
config[2] of BASELINE.json, which fits one GPU.  Inputs are resident in HBM before the timed
region (uploaded by the estimator constructor).  N > 1 (one process per GPU, launched by
torch.distributed.run): every rank runs an independent problem of the same size (weak scaling:
the path shards over independent query problems), then ONE all_gather of the
final transforms over RCCL.

Prints one JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FEATURES = 200_000
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
# algorithmic bytes per matched pair, SURVEY.md §8(d): K1 = moving xyz 24 B + index write 4 B +
# one stream of the fixed cloud 24*Nf/n B; whole fused iteration = 180 B/pair
K1_BYTES_PER_PAIR_FIXED = 28.0
ITER_BYTES_PER_PAIR = 180.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=N_FEATURES, help="features per volume (default: BASELINE size)")
    ap.add_argument("--nn", choices=["grid", "brute"], default="grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=60)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on fewer GPUs)")
    ap.add_argument("--workload", choices=["c3", "locations"], default="c3",
                    help="c3: the BASELINE metric (default); locations: config 5, independent query locations sharded over the ranks")
    ap.add_argument("--locations", type=int, default=256)
    ap.add_argument("--workers", type=int, default=4, help="host threads (device contexts) per rank in the sequential locations workload")
    ap.add_argument("--locations-mode", choices=["batch", "sequential"], default="batch",
                    help="batch: all locations of a rank advance in lockstep, one submission per phase; sequential: one estimator object at a time (the reference caller's form)")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed region; the median is reported")
    return ap.parse_args()


def cpu_baseline(pair, T0, n, iters):
    """The CPU oracle (reference restatement; kd-tree NN like the reference's rsdl_kd_tree path),
    1 thread as the reference is single-threaded (rrl sym .txx:760), on a bounded sample of the
    same workload: `iters` finest-level symmetric iterations."""
    from oracle import binding as ob
    e = ob.OracleEstimator(ob.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c,
                           moving_inside=pair.moving, fixed_inside=pair.fixed, seed=7, nn_backend=ob.NN_KDTREE)
    e.set_number_points(10**9)  # finest level directly
    e.one_iteration(0)          # warm-up (includes the MUSE scale)
    t0 = time.perf_counter()
    for it in range(iters):
        e.one_iteration(it + 1)
    dt = time.perf_counter() - t0
    pairs = iters * 2 * n
    return {"value": pairs / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "ms_per_iteration": 1e3 * dt / iters,
            "sample": f"{iters} finest-level symmetric iterations of the same {n}x{n} workload "
                      f"(C oracle, kd-tree NN, {dt:.1f} s)"}


def pmc_traffic(args):
    """HBM bytes per K1 launch from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled per the
    gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as is).  Counters cannot be collected from inside
    this process; the value is the one measured on this code at the BASELINE size with the grid
    matcher (profiles/pmc_traffic.json), null for any other configuration."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if args.nn != "grid" or args.n != N_FEATURES or not os.path.exists(path):
        return None
    try:
        return float(json.load(open(path))["traffic_bytes_per_launch"])
    except Exception:
        return None


def bench_locations(args, rank, local_rank, world):
    """config 5: `--locations` query locations x one 200k volume pair, static blocks per rank
    (32 per GPU at 256/8), one all_gather of the result rows.  Strong scaling: total work fixed.
    A step = one query location registered to convergence plus the caller's decision measurements."""
    import torch
    import torch.distributed as dist

    from lrr_amd import estimation as E
    from lrr_amd import locations, synth
    pair = synth.make_pair(args.n, args.n, 20245, kind="affine")  # same volumes on every rank
    rois = synth.query_rois(pair, args.locations, 20245, half=25.0)
    inits = [synth.perturbed(pair.planted, 1000 + i) for i in range(len(rois))]
    dev = (torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")) if world > 1 else None
    sequential = args.locations_mode == "sequential"
    reps = max(1, args.reps)
    times, pairs, rows = [], 0, None
    if sequential:
        ests = []  # the estimators (and the two uploaded volumes) live across the calls: inputs resident before timing
        locations.register_locations_sequential(pair, rois[: max(1, args.workers)], inits, 0, 1, workers=args.workers,
                                                device_index=local_rank, estimators=ests)  # warm-up
        run = lambda: locations.register_locations_sequential(pair, rois, inits, rank, world, gather_device=dev, workers=args.workers,
                                                              device_index=local_rank, estimators=ests)
    else:
        batch = E.LocationBatch(pair.moving, pair.fixed)  # the two volumes resident before timing
        locations.register_locations(pair, rois[:4], inits, 0, 1, batch=batch)  # warm-up: libraries, kernels, buffers
        run = lambda: locations.register_locations(pair, rois, inits, rank, world, gather_device=dev, batch=batch)
    for _ in range(reps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rows, pairs = run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        times.append(time.perf_counter() - t0)
    if sequential:
        for e in ests:
            if e is not None:
                e.close()
    else:
        batch.close()
    dt = float(np.median(times))
    tt = torch.tensor([dt, float(pairs)], dtype=torch.float64, device=dev if dev is not None else "cuda")
    if world > 1:
        dist.all_reduce(tt[0:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(tt[1:2], op=dist.ReduceOp.SUM)
    if rank == 0:
        dt, pairs = float(tt[0]), float(tt[1])
        print(json.dumps({
            "metric": "ICP matched-feature-pairs/sec", "value": pairs / dt, "unit": "pairs/s", "n_gpus": world,
            "steps": len(rois), "warmup": 1, "ms_per_step": 1e3 * dt / max(len(rois), 1), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{len(rois)} query locations (50 mm ROIs) x {args.n}-feature volume pair, "
                                   "symmetric ICP to convergence per location (<= 70 iterations) + decision measurements",
                       "mode": args.locations_mode,
                       "parallelism": (f"{world} rank(s), static blocks of locations, one batch per rank, one all_gather" if not sequential else
                                       f"{world} rank(s) x {args.workers} host thread(s), static blocks of locations, one all_gather")},
            "locations_per_s": len(rois) / dt, "converged": int(rows[:, 0].sum()),
            "mean_iterations": float(rows[:, 1].mean()), "max_iterations": int(rows[:, 1].max()),
            "repetitions": reps, "seconds_all": times, "row_bytes": int(rows.shape[1] * 8),
            "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ICP hot path has no CPU fallback")
    local_rank = local_rank % torch.cuda.device_count()  # rehearsals run several ranks on one GPU
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from lrr_amd import estimation as E
    from lrr_amd import multi_gpu, synth

    E.set_device(local_rank)
    if args.workload == "locations":
        return bench_locations(args, rank, local_rank, world)
    n = args.n
    # independent problem per rank: same surfaces family, different seed
    pair = synth.make_pair(n, n, 20243 + rank, kind="affine")
    T0 = synth.perturbed(pair.planted, 3 + rank)
    nn = E.NN_GRID if args.nn == "grid" else E.NN_BRUTE
    t_up = time.perf_counter()
    est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c,
                      moving_inside=pair.moving, fixed_inside=pair.fixed, seed=7, nn_method=nn)
    upload_s = time.perf_counter() - t_up
    timer = E.DeviceTimer()

    # reach the finest level with the faithful coarse-to-fine schedule, then warm up
    it = 0
    while not est.finest_level() and it < 70:
        est.one_iteration(it)
        it += 1
    schedule_iters = it
    for _ in range(args.warmup):
        est.one_iteration(it)
        it += 1
    n_pairs_step = est.number_points(0) + est.number_points(1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timer.synchronize()

    timer.match_profile(enable=True, reset=True)
    timer.cert_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        est.one_iteration(it)
        it += 1
    barrier()
    dt = time.perf_counter() - t0
    match_ms, match_launches = timer.match_profile(enable=False, reset=False)
    cert_seen, cert_searched = timer.cert_stats()

    A, t, c, _ = est.get_transform()
    row = multi_gpu.pack_result(rank, True, it, A, t, c, est.weighted_error(), est.RMS_error())
    if world > 1:
        cdev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        rows = multi_gpu.gather_results(row[None], world, device=cdev)
    else:
        rows = row[None]

    if rank == 0:
        total_pairs = args.steps * n_pairs_step * world
        value = total_pairs / dt
        k1_ms = match_ms / max(match_launches, 1)
        n_dir = n_pairs_step // 2
        k1_bytes = n_dir * K1_BYTES_PER_PAIR_FIXED + 24.0 * n  # per launch: n_dir queries vs one n-point cloud
        achieved = k1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        traffic = pmc_traffic(args)  # measured HBM bytes per launch (PMC), for the bandwidth actually drawn
        out = {
            "metric": "ICP matched-feature-pairs/sec",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{n}x{n} synthetic 3-D features, symmetric ICP "
                                   "(rrl_estimation_symmetric_ICP_matching_all<3,12>), finest-level iteration, "
                                   f"{n_pairs_step} matched pairs per step per GPU",
                       "nn": args.nn, "schedule_iterations_before_timing": schedule_iters,
                       "parallelism": f"{world} independent problem(s), one per GPU; one all_gather of the final transforms"},
            "roofline": {"bound": "hbm",
                         "kernel": ("K1 find_closest_euclidean = k_match_ball2 + k_match_hard2: every phase "
                                    "handles the forward and the backward search in one launch (lrr_icp_match_pair); "
                                    "kernel_ms = HIP-event time of the pair / 2 = one direction"
                                    if args.nn == "grid" else "K1 find_closest_euclidean = k_match_brute + k_brute_merge"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_GBs": (traffic / (k1_ms * 1e-3) / 1e9) if (traffic and k1_ms > 0) else None,
                         "traffic_frac": (traffic / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and k1_ms > 0) else None,
                         "kernel_ms": k1_ms, "launches": match_launches,
                         "certificate": {"queries": cert_seen, "searched": cert_searched,
                                         "skipped_frac": (1.0 - cert_searched / cert_seen) if cert_seen else None},
                         "algorithmic_bytes_per_launch": k1_bytes,
                         "iteration_achieved_GBs": value / world * ITER_BYTES_PER_PAIR / 1e9,
                         "iteration_frac": value / world * ITER_BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBS},
            "upload_s": upload_s,
            "final_weighted_error": float(rows[0, 17]),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pair, T0, n, args.cpu_iters)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            out["speedup_vs_cpu_1thread"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    est.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
