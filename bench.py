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
    return ap.parse_args()


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
    # submission of the timed region (each timed submission costs two event records on the critical path)
    k1_sample = int(os.environ.get("LRR_BENCH_K1_SAMPLE", "32"))  # (every 8th submission costs 2.6 % of `value`, every one 5-8 %)
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
    match_ms, match_launches = timer.match_profile(enable=False, reset=False)
    cert_seen, cert_searched = timer.cert_stats()
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

    if rank == 0:
        total_pairs = args.steps * n_pairs_step * world
        value = total_pairs / dt
        # a timed submission = one HIP-event pair for the launch(es) serving BOTH directions; launches counts directions.
        # The steady state's single kernel carries its two events itself (hipExtLaunchKernelGGL: the kernel's own start
        # and end timestamps, what rocprofv3 reports under profiles/); submissions of several kernels are bracketed by
        # two recorded events, whose marker packets add ~2 us (an empty pair alone reads event_pair_ms).
        k1_ms = match_ms / max(match_launches, 1)
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
                       "nn": args.nn, "schedule_iterations_before_timing": schedule_iters,
                       "parallelism": f"{world} independent problem(s), one per GPU; one all_gather of the final transforms"},
            "repetitions": reps, "seconds_all": times, "code_sha": code_stamp(),
            "roofline": {"bound": "hbm",
                         "kernel": ("K1 find_closest_euclidean at the finest level = k_match_cert_inline2 (certificate pass whose wavefronts "
                                    "also search the few sources that fail it; k_match_cert2 + k_match_list2 + k_match_hard2 while the lists are "
                                    "long): one launch serves the forward and the backward search (lrr_icp_match_pair); kernel_ms = HIP-event "
                                    "time of that launch (events attached to the launch itself) / 2 = one direction"
                                    if args.nn == "grid" else "K1 find_closest_euclidean = k_match_brute + k_brute_merge"),
                         "event_pair_ms": event_pair_ms, "timed_every": k1_sample,
                         "valu_frac": valu_frac,
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
        if world == 1 and not args.no_schedule:
            out["schedule"] = schedule_run(E, pair, T0, nn, timer)
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
