#!/usr/bin/env python3
"""Summaries of rocprofv3 result databases (the rocpd .db files `rocprofv3 -d DIR` writes).

  python tools/prof_summary.py kernels DIR [--iters N]   per-kernel durations (CSV on stdout): whole run
                                                         and the last N symmetric iterations; K1 wall time
  python tools/prof_summary.py pmc DIR [--last N]        per-kernel average of every collected counter over
                                                         each kernel's last N dispatches (JSON on stdout)

The files under profiles/ are the outputs of this script for the commands named in their headers.
"""
import argparse
import glob
import json
import sqlite3
import sys


def _db(path):
    hits = sorted(glob.glob(path + "/**/*.db", recursive=True)) if not path.endswith(".db") else [path]
    if not hits:
        sys.exit(f"no .db under {path}")
    return sqlite3.connect(hits[0])


def _short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def kernels(path, iters):
    c = _db(path)
    rows = c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc").fetchall()
    tot = sum(r[3] for r in rows)
    print("# whole run")
    print("kernel,calls,avg_us,total_us,percent")
    for n, k, a, t in rows:
        print(f"{_short(n)},{k},{a / 1e3:.2f},{t / 1e3:.1f},{100.0 * t / tot:.2f}")
    # the last `iters` symmetric iterations: every iteration holds exactly one fused accumulate launch
    acc = [r[0] for r in c.execute("select start from kernels where name like '%k_f_accumulate%' order by start")]
    if len(acc) < iters + 1:
        return
    first = [r[0] for r in c.execute("select start from kernels where name like '%k_match_%' or name like '%k_slots_%' order by start")]
    t0 = max(t for t in first if t < acc[-iters])  # a search kernel of that iteration ...
    t0 = min(t for t in first if t0 - t < 200_000 and t <= t0)  # ... and the first one of its pair
    rows = c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels where start>=? group by name order by 4 desc", (t0,)).fetchall()
    span = acc[-1] - acc[-iters - 1]  # accumulate launch to accumulate launch, `iters` iterations apart
    busy = sum(r[3] for r in rows)
    print(f"# last {iters} symmetric iterations: wall {span / iters / 1e3:.1f} us/iteration, sum of kernel durations {busy / iters / 1e3:.1f} us/iteration")
    print("kernel,launches_per_iteration,avg_us,us_per_iteration")
    for n, k, a, t in rows:
        print(f"{_short(n)},{k / iters:.2f},{a / 1e3:.2f},{t / iters / 1e3:.2f}")
    # K1 wall: both directions of an iteration run concurrently on two streams; wall = first start .. last end
    ks = c.execute("select start, end from kernels where start>=? and name like '%k_match_%' order by start", (t0,)).fetchall()
    groups, cur = [], None
    for s, e in ks:
        if cur is None or s - cur[1] > 60_000:  # > 60 us after the previous search kernel ended: next iteration
            cur = [s, e]
            groups.append(cur)
        else:
            cur[1] = max(cur[1], e)
    walls = [(g[1] - g[0]) / 1e3 for g in groups]
    if walls:
        print(f"# K1 (find_closest_euclidean, forward + backward concurrently): {len(walls)} submissions, "
              f"wall {sum(walls) / len(walls):.1f} us per pair = {sum(walls) / len(walls) / 2:.1f} us per direction")


def pmc(path, last):
    c = _db(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in tables:
        sys.exit("no counters_collection view in this database: " + ", ".join(tables))
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {name_col}, counter_name, dispatch_id, sum(value) from counters_collection "
                     f"group by {name_col}, counter_name, dispatch_id order by dispatch_id").fetchall()
    acc = {}
    for n, cn, _, v in rows:
        acc.setdefault((_short(n), cn), []).append(v)
    out = {}
    for (n, cn), vs in acc.items():
        tail = vs[-last:]
        out.setdefault(n, {})[f"{cn}_avg_last{last}"] = sum(tail) / len(tail)
        out[n][f"{cn}_dispatches"] = len(vs)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["kernels", "pmc"])
    ap.add_argument("dir")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--last", type=int, default=20)
    a = ap.parse_args()
    kernels(a.dir, a.iters) if a.what == "kernels" else pmc(a.dir, a.last)
