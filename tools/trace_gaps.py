"""Busy time and idle gaps of the device over the last run of a workload in a rocprofv3 --kernel-trace result directory.
usage: python tools/trace_gaps.py <dir> <first kernel of the run (name fragment)>"""
import sys

from prof_summary import _db, _short


def main():
    c = _db(sys.argv[1])
    mark = sys.argv[2]
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if mark in r[0]]
    # the last run: from the last marker dispatch that follows a pause of > 1 ms
    first = starts[-1]
    for i in reversed(starts):
        first = i
        if i == 0 or rows[i][1] - rows[i - 1][2] > 1_000_000:
            break
    run = rows[first:]
    span = run[-1][2] - run[0][1]
    busy = 0
    cur_s, cur_e = run[0][1], run[0][2]
    gaps = []
    for n, s, e in run[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, _short(n)))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"dispatches {len(run)} span {span / 1e3:.1f} us busy {busy / 1e3:.1f} us idle {(span - busy) / 1e3:.1f} us")
    by = {}
    for g, n in gaps:
        a = by.setdefault(n, [0, 0.0, 0.0])
        a[0] += 1; a[1] += g / 1e3; a[2] = max(a[2], g / 1e3)
    for n, a in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"  before {n:28s} gaps {a[0]:4d} total {a[1]:8.1f} us max {a[2]:7.1f} us")


if __name__ == "__main__":
    main()
