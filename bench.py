#!/usr/bin/env python3
"""Benchmark of the ICP hot path on MI355X — BASELINE.json's metric:
matched-feature-pairs/s (+ ms/iteration) on 200k x 200k synthetic 3-D features.

A "step" is ONE iteration of rrl_estimation_symmetric_ICP_matching_all<3,12>::one_iteration at
the finest level (every point of both ROIs matched: 200k forward + 200k backward pairs), i.e.
config[2] of BASELINE.json, which fits one GPU.  Inputs are resident in HBM before the timed
region (uploaded by the estimator constructor).  N > 1: one process per GPU — started by
torch.distributed.run, or by this script itself when it is called as `python bench.py --gpus N`
without a rendezvous in the environment (the parent makes no GPU call and only starts the ranks):
every rank runs an independent problem of the same size (weak scaling: the path shards over
independent query problems), then ONE all_gather of the final transforms over RCCL.

The same line carries, as sub-records: the three regimes of the 200k x 200k workload (`value` = the
converged steady state; `moving_transform` = set_transform(perturbed) at the finest level -> iterate to
convergence; `schedule` = the faithful 1000 -> N run), config 5 (`locations`: 256 query locations sharded
over the ranks, with its own roofline and CPU baseline), the production matcher's lookup semantics on
the GPU (`voronoi`), and the CPU oracle beside them (`cpu_baseline`).

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
    ap.add_argument("--cpu-iters", type=int, default=24, help="finest-level oracle iterations of the single-thread CPU baseline")
    ap.add_argument("--no-schedule", action="store_true", help="skip the faithful 1000 -> N schedule run")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on fewer GPUs)")
    ap.add_argument("--workload", choices=["c3", "locations", "cdc"], default="c3",
                    help="c3: the BASELINE metric (default); locations: config 5, independent query locations sharded over the ranks; "
                         "cdc: the covariance-driven EM step (SURVEY 8(f)4): E-step + the M-step's objective evaluations, one GPU")
    ap.add_argument("--locations", type=int, default=256)
    ap.add_argument("--workers", type=int, default=4, help="host threads (device contexts) per rank in the sequential locations workload")
    ap.add_argument("--locations-mode", choices=["batch", "sequential"], default="batch",
                    help="batch: all locations of a rank advance in lockstep, one submission per phase; sequential: one estimator object at a time (the reference caller's form)")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed region; the median is reported")
    ap.add_argument("--no-locations", action="store_true", help="skip the config-5 sub-record of the default workload")
    ap.add_argument("--no-moving", action="store_true", help="skip the moving-transform regime")
    ap.add_argument("--no-voronoi", action="store_true", help="skip the GPU run with the Voronoi-lookup semantics")
    ap.add_argument("--cpu-locations", type=int, default=64, help="query locations the CPU oracle registers for the locations baseline")
    ap.add_argument("--code-stamp", action="store_true", help="print the stamp of the kernel sources and exit (tools/*.sh record it)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a rendezvous in the environment: this process has made no GPU call (torch is not
    even imported yet) and starts the N ranks as children through torch.distributed.run, one per GPU, rendezvous on
    127.0.0.1; rank 0's JSON line is passed through to stdout (anything else the children write there goes to stderr).  Never
    an exec: the parent waits and returns the children's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    # the contract is ONE JSON line on stdout: whatever else the children (or the collective library under them) write to
    # stdout goes to stderr, rank 0's line is printed last
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks printed no result line\n")
        rc = 1
    return rc


def _oracle_iterations(pair, T0, n, iters, nn_backend, out, k):
    from oracle import binding as ob
    e = ob.OracleEstimator(ob.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c,
                           moving_inside=pair.moving, fixed_inside=pair.fixed, seed=7 + k, nn_backend=nn_backend)
    e.set_number_points(10**9)  # finest level directly
    e.one_iteration(0)          # warm-up (includes the MUSE scale)
    t0 = time.perf_counter()
    for it in range(iters):
        e.one_iteration(it + 1)
    out[k] = (time.perf_counter() - t0, t0)
    e.close()


def cpu_baseline(pair, T0, n, iters):
    """The CPU oracle (reference restatement) on bounded samples of the same workload, on the GPU box's host cores:
    * value: finest-level symmetric iterations with the kd-tree matcher (the cdcl classes' rsdl_kd_tree path),
      1 thread — the reference is single-threaded (rrl sym .txx:760);
    * all_cores: the same iterations run concurrently on independent copies of the problem, one per core (ctypes
      releases the GIL): the all-core throughput of independent estimates;
    * brute: the all-vs-all matcher alone on a sample of the queries, extrapolated to one iteration's matching.
    The production rrl class uses neither: it looks matches up in a precomputed 1 mm Voronoi volume (O(1) per
    query; not restated)."""
    import threading

    from oracle import binding as ob
    res = {}
    _oracle_iterations(pair, T0, n, iters, ob.NN_KDTREE, res, 0)
    dt = res[0][0]
    pairs = iters * 2 * n
    out = {"value": pairs / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
           "matcher": "kd-tree (exact, as rsdl_kd_tree of the cdcl classes); the rrl classes' Voronoi-volume lookup is timed beside it: voronoi_lookup",
           "ms_per_iteration": 1e3 * dt / iters,
           "sample": f"{iters} finest-level symmetric iterations of the same {n}x{n} workload "
                     f"(C oracle, kd-tree NN, {dt:.1f} s)"}
    # the production matcher of the rrl class: O(1) lookups in the 1 mm Voronoi index volume (filled lazily by the
    # warm-up iteration and the first timed ones; the reference precomputes it offline with compute_distance_map)
    res = {}
    _oracle_iterations(pair, T0, n, iters, ob.NN_VORONOI, res, 0)
    out["voronoi_lookup"] = {"value": pairs / res[0][0], "unit": "pairs/s", "cores": 1, "ms_per_iteration": 1e3 * res[0][0] / iters,
                             "sample": f"{iters} finest-level symmetric iterations, matches looked up in the 1 mm Voronoi index volume "
                                       f"(rrl/rrl_estimation_ICP_matching_all.txx:530-543; {res[0][0]:.1f} s); different match sets than the "
                                       "exact matchers: the GPU line above runs the exact search"}
    # all cores: independent problems side by side
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # the GPU box gives one GPU's share of the host: 16 cores
    it2 = max(2, iters // 4)
    res, th = {}, []
    t0 = time.perf_counter()
    for k in range(cores):
        th.append(threading.Thread(target=_oracle_iterations, args=(pair, T0, n, it2, ob.NN_KDTREE, res, k)))
        th[-1].start()
    for t in th:
        t.join()
    span = max(r[1] + r[0] for r in res.values()) - min(r[1] for r in res.values())  # timed regions only
    out["all_cores"] = {"value": cores * it2 * 2 * n / span, "unit": "pairs/s", "cores": cores,
                        "sample": f"{cores} independent copies of the workload, {it2} finest-level iterations each, one thread per copy ({span:.1f} s)"}
    # brute force: the matcher alone, sampled
    nq = 1500
    q = pair.planted.map(pair.moving.loc[:nq])
    t0 = time.perf_counter()
    ob.nn(pair.fixed.loc, q, ob.NN_BRUTE)
    tb = time.perf_counter() - t0
    out["brute"] = {"value": nq / tb, "unit": "matched pairs/s (matching only)", "cores": 1,
                    "ms_per_iteration_matching_extrapolated": 1e3 * tb / nq * 2 * n,
                    "sample": f"all-vs-all nearest neighbour of {nq} of the {n} queries against {n} targets ({tb:.2f} s), "
                              "extrapolated linearly to the 2 x n queries of an iteration"}
    return out


def code_stamp():
    """sha256 (16 hex) of the kernel sources: ties committed PMC measurements to the code they were taken on
    (.git does not travel to the GPU box)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("lrr_icp.hip", "lrr_icp_k1.inc", "lrr_icp_solve.inc", "lrr_icp_passes.inc", "lrr_icp_fused.inc", "lrr_icp_grid.inc", "lrr_icp_batch.inc",
              "lrr_icp_cdc.inc"):
        with open(os.path.join(ROOT, "lrr_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_file(args, name):
    """A committed PMC measurement (profiles/<name>), or None when it was not taken on THIS code (code_stamp) or
    at another configuration: counters cannot be collected from inside this process."""
    path = os.path.join(ROOT, "profiles", name)
    if args.nn != "grid" or args.n != N_FEATURES or not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
    except Exception:
        return None
    return d if d.get("code_sha") == code_stamp() else None


def schedule_run(E, pair, T0, nn, timer):
    """The faithful run: 1000 -> 2000 -> ... -> N points, every level to its convergence test, then the finest level
    to convergence (<= 70 iterations, the caller's limit).  Per iteration: wall time and K1 time (HIP events)."""
    import torch
    est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c,
                      moving_inside=pair.moving, fixed_inside=pair.fixed, seed=7, nn_method=nn)
    n = pair.moving.n
    rows, conv, it = [], False, 0
    timer.match_profile(enable=True, reset=True)
    timer.synchronize()
    t_all = time.perf_counter()
    while not conv and it < 70:
        npts = est.number_points(0)
        t0 = time.perf_counter()
        conv = est.one_iteration(it)
        timer.synchronize()
        wall = time.perf_counter() - t0
        k1, launches = timer.match_profile(enable=True, reset=True)
        rows.append((npts, 1e3 * wall, k1 / max(launches, 1)))
        it += 1
    total = time.perf_counter() - t_all
    timer.match_profile(enable=False, reset=True)
    est.close()
    # the same run as ONE call (the caller's loop in C++: no interpreter, no per-iteration synchronisation or event
    # records between the iterations): the time a caller of run() sees
    est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c,
                      moving_inside=pair.moving, fixed_inside=pair.fixed, seed=7, nn_method=nn)
    timer.synchronize()
    t0 = time.perf_counter()
    conv2, it2, pairs2 = est.run_counted(70)
    timer.synchronize()
    total_run = time.perf_counter() - t0
    est.close()
    npts = np.array([r[0] for r in rows]); wall = np.array([r[1] for r in rows]); k1 = np.array([r[2] for r in rows])
    first = np.r_[True, npts[1:] != npts[:-1]]
    coarse, finest = npts * 4 < n, npts == n
    def mean(mask, v):
        return float(v[mask].mean()) if mask.any() else None
    return {"iterations": it, "converged": bool(conv), "total_ms": 1e3 * total, "run_call_ms": 1e3 * total_run, "run_call_iterations": it2,
            "levels": int(first.sum()),
            "matched_pairs": int(2 * npts.sum()),
            "k1_ms_per_direction": {"coarse_levels": mean(coarse, k1), "dense_levels_first_search": mean(~coarse & first, k1),
                                    "dense_levels_later_searches": mean(~coarse & ~first & ~finest, k1),
                                    "finest_level_first_search": mean(finest & first, k1),
                                    "finest_level_later_searches": mean(finest & ~first, k1)},
            "iteration_ms": {"coarse_levels": mean(coarse, wall), "finest_level": mean(finest, wall)},
            "per_iteration": [{"n_points": int(a), "ms": round(float(b), 4), "k1_ms_per_direction": round(float(c), 5)} for a, b, c in rows]}


def cpu_locations(pair, rois, inits, rows_gpu, count, seed=7):
    """config 5 on the CPU oracle, 1 thread: the first `count` query locations registered one after the other as the
    reference's caller does (set_features + set_transform, one_iteration until converged / 70 / an invalid transform).
    Pairs per location are the GPU rows' (the oracle forms the same ones: tests/test_gpu_batch.py)."""
    from lrr_amd import estimation as E
    from lrr_amd import locations
    from oracle import binding as ob
    C = E.LocationBatch.COL
    count = max(1, min(count, len(rois)))
    same, o, dt = 0, None, 0.0
    for i in range(count):
        _, _, mi, fi = rois[i]
        T0 = inits[i]
        m_in, f_in = pair.moving.subset(mi), pair.fixed.subset(fi)  # (FeaturesInROIs; numpy, outside the timed part on both sides)
        if o is None:  # the search structures of the two volumes are built once, as the reference reads its Voronoi maps once
            o = ob.OracleEstimator(ob.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c, moving_inside=m_in, fixed_inside=f_in,
                                   seed=1, nn_backend=ob.NN_KDTREE)
        t0 = time.perf_counter()
        o.set_features(m_in, f_in)          # SetupEstimation (:1635-1637)
        o.set_transform(T0.A, T0.t, T0.c)
        o.set_subset_seed(locations.location_seed(seed, i))
        conv, its = o.run(locations.MAX_ITERATIONS)
        dt += time.perf_counter() - t0
        same += int(bool(rows_gpu[i, C["converged"]]) == conv and int(rows_gpu[i, C["iterations"]]) == its)
    o.close()
    pairs = float(rows_gpu[:count, C["matched_pairs"]].sum())
    return {"value": pairs / dt, "unit": "pairs/s", "cores": 1, "kind": "port", "locations_per_s": count / dt,
            "sample": f"the first {count} of the {len(rois)} query locations, registered one after the other by the C oracle (kd-tree NN) "
                      f"to convergence: set_features + set_transform + the caller's loop ({dt:.2f} s; the kd-trees of the two volumes are built once, "
                      "outside the timed part; no decision measurements)",
            "same_flags_and_iterations_as_gpu": f"{same}/{count}"}


def locations_measure(args, rank, local_rank, world, cpu=True):
    """config 5: `--locations` query locations x one 200k volume pair, static blocks per rank (32 per GPU at 256/8), one
    all_gather of the result rows.  Strong scaling: total work fixed.  A step = one query location registered to
    convergence plus the caller's decision measurements.  Returns the record on rank 0 (None elsewhere)."""
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
    times, pairs, rows, phases = [], 0, None, []
    if sequential:
        ests = []  # the estimators (and the two uploaded volumes) live across the calls: inputs resident before timing
        locations.register_locations_sequential(pair, rois[: max(1, args.workers)], inits, 0, 1, workers=args.workers,
                                                device_index=local_rank, estimators=ests)  # warm-up
        run = lambda: locations.register_locations_sequential(pair, rois, inits, rank, world, gather_device=dev, workers=args.workers,
                                                              device_index=local_rank, estimators=ests)
    else:
        batch = E.LocationBatch(pair.moving, pair.fixed)  # the two volumes resident before timing
        # warm-up: libraries and kernels, then the rank's own block once so that every buffer has its final size (the first
        # full run otherwise pays the allocations: 13 against 10 ms)
        locations.register_locations(pair, rois[:4], inits, 0, 1, batch=batch)
        locations.register_locations(pair, rois, inits, rank, world, gather_device=dev, batch=batch)
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
        if not sequential:
            phases.append(batch.last_timing())
    vor = None
    if sequential:
        for e in ests:
            if e is not None:
                e.close()
    else:
        if world == 1 and not args.no_voronoi:
            # the same batch with the lookup semantics of the Voronoi maps (what rrl_location_batch::set_voronoi_maps() selects)
            batch.set_nn_method(E.NN_VORONOI)
            locations.register_locations(pair, rois, inits, 0, 1, batch=batch)  # builds the site clouds, sizes the buffers
            vt = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                vrows, vpairs = locations.register_locations(pair, rois, inits, 0, 1, batch=batch)
                torch.cuda.synchronize()
                vt.append(time.perf_counter() - t0)
            vdt = float(np.median(vt))
            VC = E.LocationBatch.COL
            vor = {"what": "the same locations matched with the lookup semantics of the 1 mm Voronoi maps (LRR_ICP_NN_VORONOI)",
                   "locations_per_s": len(rois) / vdt, "value": vpairs / vdt, "unit": "pairs/s", "ms_total": 1e3 * vdt,
                   "converged": int(vrows[:, VC["converged"]].sum()), "mean_iterations": float(vrows[:, VC["iterations"]].mean())}
        batch.close()
    dt = float(np.median(times))
    ph = phases[int(np.argsort(times)[len(times) // 2])] if phases else {}
    # [median wall, this rank's pairs, this rank's device-bound phases (ROI selection + the iterations)]
    tt = torch.tensor([dt, float(pairs), float(ph.get("iterate", 0.0)), float(ph.get("roi", 0.0))], dtype=torch.float64,
                      device=dev if dev is not None else "cuda")
    if world > 1:
        dist.all_reduce(tt[0:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(tt[1:2], op=dist.ReduceOp.SUM)
        dist.all_reduce(tt[2:4], op=dist.ReduceOp.MAX)
    if rank != 0:
        return None
    dt, pairs, it_ms, roi_ms = float(tt[0]), float(tt[1]), float(tt[2]), float(tt[3])
    C = E.LocationBatch.COL
    rec = {
        "metric": "ICP matched-feature-pairs/sec", "value": pairs / dt, "unit": "pairs/s", "n_gpus": world,
        "steps": len(rois), "warmup": 1, "ms_per_step": 1e3 * dt / max(len(rois), 1), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{len(rois)} query locations (50 mm ROIs) x {args.n}-feature volume pair, "
                               "symmetric ICP to convergence per location (<= 70 iterations, validTransform break) + decision measurements",
                   "mode": args.locations_mode,
                   "parallelism": (f"{world} rank(s), static blocks of locations, one batch per rank, one all_gather" if not sequential else
                                   f"{world} rank(s) x {args.workers} host thread(s), static blocks of locations, one all_gather")},
        "locations_per_s": len(rois) / dt, "ms_total": 1e3 * dt, "converged": int(rows[:, 0].sum()),
        "mean_iterations": float(rows[:, 1].mean()), "max_iterations": int(rows[:, 1].max()),
        "matched_pairs": int(pairs), "repetitions": reps, "seconds_all": times, "row_bytes": int(rows.shape[1] * 8),
        "phases_ms": ph, "roofline": None, "cpu_baseline": None}
    if not sequential and it_ms > 0.0:
        # every matched pair of every iteration goes once through K1 (k_b_match + the wave-per-query phase) and the five
        # passes of the estimate (k_b_residual, k_b_stats, k_b_weights, k_b_select, k_b_accumulate) + k_b_errors:
        # SURVEY 8(d)'s 180 algorithmic bytes per pair, over the time the device works on the iterations (the slowest
        # rank's lrr_icp_batch_iterate: the host queues ahead of the device there)
        ach = pairs / world * ITER_BYTES_PER_PAIR / (it_ms * 1e-3) / 1e9
        rec["roofline"] = {"bound": "hbm", "kernel": "one lockstep iteration of all locations of a rank = k_b_match + k_match_hard2 "
                                                     "+ k_b_residual, k_b_stats, k_b_weights, k_b_select, k_b_accumulate + k_run_solve, k_b_errors, k_run_decide "
                                                     "(lrr_icp_batch_iterate); latency-bound: a location is a few thousand pairs",
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "kernel_ms": it_ms, "algorithmic_bytes_per_launch": pairs / world * ITER_BYTES_PER_PAIR,
                           "note": "achieved = (matched pairs of the slowest rank's block x 180 B) / its iterate phase; ROI selection "
                                   f"{roi_ms:.3f} ms and the decision measurements are outside it"}
    if cpu and world == 1 and not args.no_cpu_baseline and not sequential:
        rec["cpu_baseline"] = cpu_locations(pair, rois, inits, rows, args.cpu_locations)
        rec["speedup_vs_cpu_1thread"] = rec["value"] / rec["cpu_baseline"]["value"]
    if vor is not None:
        rec["voronoi"] = vor
    return rec


def bench_locations(args, rank, local_rank, world):
    import torch.distributed as dist
    rec = locations_measure(args, rank, local_rank, world)
    if rank == 0:
        rec["code_sha"] = code_stamp()
        print(json.dumps(rec))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_cdc(args):
    """one EM step of cdcl_estimation on n x n features with covariances — compute_weights (E-step), then what
    the two L-BFGS calls of the M-step evaluate (five objective + gradient evaluations each, cdcl/cdcl_estimation.txx:230-239,
    298-304) — device time by HIP events, the CPU oracle on a bounded sample beside it."""
    import dataclasses
    from lrr_amd import estimation as E
    from lrr_amd import synth
    n, dof, mm = args.n, 12, 100
    pair = synth.make_pair(n, n, 20246, kind="affine")
    mv = dataclasses.replace(pair.moving, cov=synth.covariances(pair.moving))
    fx = dataclasses.replace(pair.fixed, cov=synth.covariances(pair.fixed))
    T0 = synth.perturbed(pair.planted, 5, rot_deg=0.3, trans_mm=0.4)
    rng = np.random.default_rng(2)
    B = rng.normal(size=(dof, dof)) * 2e-3
    B[9:] *= 50.0
    Ct = B @ B.T
    xu = np.linalg.cholesky(Ct).T[np.triu_indices(dof)]
    cm, cf = E.DeviceCloud(mv), E.DeviceCloud(fx)
    cm.set_covariances(mv.cov)
    cf.set_covariances(fx.cov)
    S = E.CdcSet(cm, cf, max_matches=mm)
    timer = E.DeviceTimer()
    S.compute_weights(dof, T0.A, T0.t, T0.c, Ct, seed=33)  # warm-up
    S.objective_par(dof, T0.A, T0.t, T0.c, Ct)
    S.objective_cov(dof, T0.A, T0.t, T0.c, xu)

    def timed(fn, reps):
        out = []
        for _ in range(reps):
            timer.synchronize()
            timer.start()
            fn()
            out.append(timer.stop())
        return float(np.median(out))

    reps = max(1, args.reps)
    e_ms = timed(lambda: S.compute_weights(dof, T0.A, T0.t, T0.c, Ct, seed=33), reps)
    norm = S.normalize_stats()
    n_ms = timed(lambda: S.normalize_stats(), reps)
    p_ms = timed(lambda: S.objective_par(dof, T0.A, T0.t, T0.c, Ct), reps)
    c_ms = timed(lambda: S.objective_cov(dof, T0.A, T0.t, T0.c, xu), reps)
    rc, ti, w = S.rows()
    pairs = int(rc.sum())
    step_ms = e_ms + n_ms + 5 * p_ms + 5 * c_ms
    out = {"metric": "CDC-EM matched pairs/s (one EM step: E-step + 5 + 5 objective evaluations)", "value": pairs / (step_ms * 1e-3),
           "unit": "pairs/s", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{n}x{n} synthetic 3-D features with covariances, cdcl_estimation<3,12> EM step, <= {mm} matches per moving feature",
                      "pairs": pairs, "rows": int((rc > 0).sum()), "mean_matches_per_row": float(rc[rc > 0].mean())},
           "device_ms": {"compute_weights": e_ms, "normalize_stats": n_ms, "objective_par": p_ms, "objective_cov": c_ms},
           "roofline": {"bound": "hbm", "kernel": "k_cdc_obj<12> (cdcl_obj_fun_par::compute): per pair 4 B index + 8 B weight + 32 B point + 72 B covariance, gathered",
                        "achieved": pairs * 116.0 / (p_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": pairs * 116.0 / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel_ms": p_ms,
                        "algorithmic_bytes_per_launch": pairs * 116.0},
           "repetitions": reps}
    if not args.no_cpu_baseline:
        from oracle import binding as ob
        ns = min(2000, n)  # the oracle's radius search is brute force: a sample of the moving set against the whole fixed set
        sub = mv.subset(np.arange(ns))
        t0 = time.perf_counter()
        orc, oti, ow = ob.cdc_compute_weights(sub, fx, dof, T0.A, T0.t, T0.c, Ct, seed=33, max_matches=mm)
        te = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(3):
            ob.cdc_obj_par(sub, fx, dof, T0.A, T0.t, T0.c, Ct, orc, oti, ow)
        tp = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3):
            ob.cdc_obj_cov(sub, fx, dof, T0.A, T0.t, T0.c, xu, orc, oti, ow)
        tc = (time.perf_counter() - t0) / 3
        op = int(orc.sum())
        cpu_step = te + 5 * tp + 5 * tc
        out["cpu_baseline"] = {"value": op / cpu_step, "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"the first {ns} moving features against all {n} fixed ones ({op} pairs): E-step {te:.2f} s (brute-force radius "
                                         f"search, the oracle's stand-in for the kd-tree), objective_par {1e3 * tp:.1f} ms, objective_cov {1e3 * tc:.1f} ms",
                               "objective_par_pairs_per_s": op / tp, "objective_cov_pairs_per_s": op / tc}
        out["device_objective_par_pairs_per_s"] = pairs / (p_ms * 1e-3)
        out["device_objective_cov_pairs_per_s"] = pairs / (c_ms * 1e-3)
    print(json.dumps(out))
    S.close()
    cm.close()
    cf.close()


def sample_stats(ms, per):
    """median / spread of HIP-event samples (ms per submission) as time per `per`-th of a submission"""
    v = np.sort(np.asarray(ms, dtype=np.float64)) / per
    if v.size == 0:
        return None
    q = lambda f: float(v[min(v.size - 1, int(f * v.size))])
    return {"n": int(v.size), "median_ms": float(np.median(v)), "mean_ms": float(v.mean()), "p10_ms": q(0.10), "p90_ms": q(0.90),
            "min_ms": float(v[0]), "max_ms": float(v[-1])}


def moving_transform(est, E, pair, timer, reps):
    """The regime a caller of set_transform() sees at the finest level: the transform is set to planted (+) a fresh
    perturbation (0.5 deg, 0.5 mm: what T0 is), every point of both ROIs is active (set_number_points), and the caller's
    loop runs until the class reports convergence.  The first iteration takes its scale from MUSE and searches without a
    certificate; the certificate comes in as the transform settles.  Wall time of the whole loop (one C++ call, no
    synchronisation inside) over `reps` repetitions + one more repetition with HIP events on every K1 submission."""
    from lrr_amd import synth
    n = pair.moving.n
    rows = []
    k1 = None
    for r in range(reps + 2):  # 0: warm-up (untimed); last: K1 events on every submission (its wall time is not used)
        Tp = synth.perturbed(pair.planted, 500 + r)
        est.set_transform(Tp.A, Tp.t, Tp.c)
        est.set_number_points(10**9)
        events = r == reps + 1
        if events:
            timer.match_profile(enable=1, reset=True)
            timer.cert_stats(reset=True)
            timer.k1_path_counts(reset=True)
        timer.synchronize()
        t0 = time.perf_counter()
        conv, its, pairs = est.run_counted(70)
        timer.synchronize()
        dt = time.perf_counter() - t0
        if events:
            ms = timer.match_samples()
            seen, searched = timer.cert_stats()
            k1 = {"per_direction": sample_stats(ms, 2), "per_iteration_ms_per_direction": [round(float(x) / 2, 5) for x in ms],
                  "certificate": {"queries": seen, "searched": searched, "skipped_frac": (1.0 - searched / seen) if seen else None},
                  "searches_skipped_frac_of_all": (1.0 - (searched + (2 * n * its - seen)) / (2.0 * n * its)) if its else None,
                  "paths": timer.k1_path_counts(reset=True)}
            timer.match_profile(enable=False, reset=True)
        elif r > 0:
            rows.append((dt, its, pairs, conv))
    dts = np.array([x[0] for x in rows])
    mid = int(np.argsort(dts)[len(dts) // 2])
    dt, its, pairs, conv = rows[mid]
    k1_bytes = n * K1_BYTES_PER_PAIR_FIXED + 24.0 * n
    out = {"what": "set_transform(planted (+) perturbation) at the finest level, then one_iteration until converged (<= 70)",
           "repetitions": reps, "iterations": [int(x[1]) for x in rows], "converged": [bool(x[3]) for x in rows],
           "ms_total_all": [1e3 * float(x[0]) for x in rows], "ms_total": 1e3 * dt, "matched_pairs": int(pairs),
           "value": pairs / dt, "unit": "pairs/s", "ms_per_iteration": 1e3 * dt / max(its, 1), "k1": k1}
    if k1 and k1["per_direction"]:
        ach = k1_bytes / (k1["per_direction"]["mean_ms"] * 1e-3) / 1e9
        out["k1_roofline"] = {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                              "kernel_ms": k1["per_direction"]["mean_ms"], "algorithmic_bytes_per_launch": k1_bytes,
                              "note": "mean over ALL K1 submissions of the loop (ball search while the transform moves, certificate once it settles)"}
    return out


def voronoi_gpu(E, pair, T0, timer, steps, reps):
    """The production matcher's semantics on the GPU (LRR_ICP_NN_VORONOI: the quantised lookup in the 1 mm Voronoi volume,
    rrl/rrl_estimation_ICP_matching_all.txx:530-543): same workload, finest-level iterations after the faithful schedule."""
    est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c, moving_inside=pair.moving, fixed_inside=pair.fixed,
                      seed=7, nn_method=E.NN_VORONOI)
    it = 0
    timer.synchronize()
    t0 = time.perf_counter()
    while not est.finest_level() and it < 70:
        est.one_iteration(it)
        it += 1
    timer.synchronize()
    sched_ms = 1e3 * (time.perf_counter() - t0)
    for _ in range(3):
        est.one_iteration(it)
        it += 1
    n_pairs = est.number_points(0) + est.number_points(1)
    times = []
    for _ in range(reps):
        timer.synchronize()
        t0 = time.perf_counter()
        est.iterate(it, steps)
        it += steps
        timer.synchronize()
        times.append(time.perf_counter() - t0)
    timer.match_profile(enable=1, reset=True)
    est.iterate(it, 16)
    ms = timer.match_samples()
    timer.match_profile(enable=False, reset=True)
    werr = est.weighted_error()
    dt = float(np.median(times))
    # the other two regimes, as for the exact search: the transform set anew at the finest level, and the caller's whole run()
    mv = moving_transform(est, E, pair, timer, reps)
    est.close()
    est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c, moving_inside=pair.moving, fixed_inside=pair.fixed,
                      seed=7, nn_method=E.NN_VORONOI)
    runs = []
    for r in range(3):
        est.set_transform(T0.A, T0.t, T0.c)
        est.set_subset_seed(7)
        timer.synchronize()
        t0 = time.perf_counter()
        conv, its, pairs = est.run_counted(70)
        timer.synchronize()
        runs.append((time.perf_counter() - t0, its, pairs, conv))
    est.close()
    rt, rits, rpairs, rconv = sorted(runs)[len(runs) // 2]
    return {"what": "the same workload with the rrl classes' own matcher semantics (nearest occupied voxel of the 1 mm Voronoi volume; "
                    "different match sets than the exact search of `value`): what set_voronoi_maps() selects by default",
            "value": steps * n_pairs / dt, "unit": "pairs/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "repetitions": reps,
            "schedule_iterations": it - 3 - steps * reps - 16, "schedule_ms_incl_first_voronoi_build": sched_ms,
            "k1_per_direction": sample_stats(ms, 2), "final_weighted_error": werr,
            "regimes": {"unit": "pairs/s", "steady_converged": steps * n_pairs / dt, "moving_transform_finest_level": mv["value"],
                        "whole_schedule_run_call": rpairs / rt},
            "moving_transform": {k: mv[k] for k in ("iterations", "converged", "ms_total", "matched_pairs", "ms_per_iteration")},
            "moving_transform_k1_per_direction": (mv.get("k1") or {}).get("per_direction"),
            "schedule": {"run_call_ms": 1e3 * rt, "iterations": int(rits), "converged": bool(rconv), "matched_pairs": int(rpairs)}}


def main():
    args = parse()
    if args.code_stamp:
        print(code_stamp())
        return 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)
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
    if args.workload == "cdc":
        if world > 1:
            raise SystemExit("--workload cdc is a one-GPU measurement")
        return bench_cdc(args)
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

    # the timed region: K steps between barriers, repeated `reps` times; the MEDIAN repetition is reported
    reps = max(1, args.reps)
    times = []
    # K1's launch duration is taken live with HIP events, on the stream that runs it, around every k1_sample-th K1
    # submission of the timed region (each timed submission costs two event records on the critical path: every
    # submission 5-8 % of `value`, every 8th 2.6 %).  The timed region yields steps x reps / k1_sample samples; a sampling
    # pass of its own after it (every submission timed, same steady state) brings the total to >= 64 whatever --steps is.
    want_samples = 64
    k1_sample = int(os.environ.get("LRR_BENCH_K1_SAMPLE", "0")) or max(4, min(32, (args.steps * reps) // 24))
    timer.match_profile(enable=k1_sample, reset=True)
    timer.cert_stats(reset=True)
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        est.iterate(it, args.steps)  # exactly K one_iteration() calls, in one C++ loop
        it += args.steps
        barrier()
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    samples_timed = timer.match_samples()
    cert_seen, cert_searched = timer.cert_stats()
    extra = max(0, want_samples - samples_timed.size)
    samples_extra = np.zeros(0)
    if extra:
        timer.match_profile(enable=1, reset=True)
        est.iterate(it, extra)
        it += extra
        samples_extra = timer.match_samples()
    timer.match_profile(enable=False, reset=True)
    samples = np.concatenate([samples_timed, samples_extra])
    event_pair_ms = timer.event_overhead(64)  # an empty event pair on the same stream: the markers' share of a timed submission

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

    out = None
    if rank == 0:
        total_pairs = args.steps * n_pairs_step * world
        value = total_pairs / dt
        # a timed submission = one HIP-event pair for the launch(es) serving BOTH directions.  The steady state's single
        # kernel carries its two events itself (hipExtLaunchKernelGGL: the kernel's own start and end timestamps, what
        # rocprofv3 reports under profiles/); submissions of several kernels are bracketed by two recorded events, whose
        # marker packets add ~2 us (an empty pair alone reads event_pair_ms).
        st = sample_stats(samples, 2)
        k1_ms = st["median_ms"] if st else 0.0
        n_dir = n_pairs_step // 2
        k1_bytes = n_dir * K1_BYTES_PER_PAIR_FIXED + 24.0 * n  # per launch: n_dir queries vs one n-point cloud
        achieved = k1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        pt, pu = pmc_file(args, "pmc_traffic.json"), pmc_file(args, "pmc_utilisation.json")
        traffic = float(pt["traffic_bytes_per_launch"]) if pt else None  # measured HBM bytes per launch (PMC), null when not taken on this code
        valu_frac = float(pu["valu_frac"]) if pu else None
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
                       "regime": "steady: the estimate has converged when the timed steps start (the transform no longer moves and K1's "
                                 "certificate skips nearly every search); see regimes / moving_transform / schedule for the other two",
                       "nn": args.nn, "schedule_iterations_before_timing": schedule_iters,
                       "parallelism": f"{world} independent problem(s), one per GPU; one all_gather of the final transforms"},
            "repetitions": reps, "seconds_all": times, "code_sha": code_stamp(),
            "roofline": {"bound": "hbm",
                         "kernel": ("K1 find_closest_euclidean at the finest level, steady state = k_match_cert_inline2 (certificate pass whose wavefronts "
                                    "also search the few sources that fail it): one launch serves the forward and the backward search "
                                    "(lrr_icp_match_pair); kernel_ms = MEDIAN HIP-event time of that launch (events attached to the launch itself) / 2 "
                                    "= one direction" if args.nn == "grid" else "K1 find_closest_euclidean = k_match_brute + k_brute_merge"),
                         "event_pair_ms": event_pair_ms, "timed_every": k1_sample,
                         "samples": {"per_direction": st, "in_timed_region": int(samples_timed.size), "in_sampling_pass": int(samples_extra.size),
                                     "timed_region_only": sample_stats(samples_timed, 2)},
                         "valu_frac": valu_frac,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_GBs": (traffic / (k1_ms * 1e-3) / 1e9) if (traffic and k1_ms > 0) else None,
                         "traffic_frac": (traffic / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and k1_ms > 0) else None,
                         "kernel_ms": k1_ms, "launches": int(samples.size),
                         "certificate": {"queries": cert_seen, "searched": cert_searched,
                                         "skipped_frac": (1.0 - cert_searched / cert_seen) if cert_seen else None},
                         "algorithmic_bytes_per_launch": k1_bytes,
                         "iteration_achieved_GBs": value / world * ITER_BYTES_PER_PAIR / 1e9,
                         "iteration_frac": value / world * ITER_BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBS},
            "upload_s": upload_s,
            "final_weighted_error": float(rows[0, 17]),
        }
    # the other two regimes of the same workload, the production matcher's semantics, the CPU beside them (one GPU only)
    if world == 1:
        if not args.no_moving and args.nn == "grid":
            out["moving_transform"] = moving_transform(est, E, pair, timer, 5)
        if not args.no_schedule:
            out["schedule"] = schedule_run(E, pair, T0, nn, timer)
            sc = out["schedule"]
            sc["value"] = sc["matched_pairs"] / (sc["run_call_ms"] * 1e-3)
            sc["unit"] = "pairs/s"
        out["regimes"] = {
            "unit": "pairs/s",
            "steady_converged": value,
            "moving_transform_finest_level": out.get("moving_transform", {}).get("value"),
            "whole_schedule_run_call": out.get("schedule", {}).get("value"),
            "note": "steady = `value` (every timed step is past convergence); moving = set_transform(perturbed) at the finest level, "
                    "iterated to convergence; whole schedule = the faithful 1000 -> N run() a caller of the class makes"}
        if not args.no_voronoi and args.nn == "grid":
            out["voronoi"] = voronoi_gpu(E, pair, T0, timer, min(args.steps, 50), 3)
    est.close()
    # config 5 beside it: every rank takes its block of the query locations (the driver's --gpus N run then measures it too)
    if not args.no_locations and args.nn == "grid":
        rec = locations_measure(args, rank, local_rank, world)
        if rank == 0:
            for k in ("metric", "higher_is_better", "vs_baseline", "dtype", "data", "warmup"):
                rec.pop(k, None)
            out["locations"] = rec
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pair, T0, n, args.cpu_iters)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            out["cpu_baseline"]["compared_with"] = ("`value` is the steady regime; the CPU path's cost does not depend on the regime — the "
                                                    "ratios against the other two regimes are speedup_vs_cpu_1thread_by_regime")
            out["speedup_vs_cpu_1thread"] = value / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_1thread_by_regime"] = {k: (v / out["cpu_baseline"]["value"] if v else None)
                                                        for k, v in out["regimes"].items() if k not in ("unit", "note")}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
