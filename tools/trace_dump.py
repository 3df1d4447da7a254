"""Every kernel dispatch of a rocprofv3 --kernel-trace result directory in launch order: start (us since the first
dispatch), duration (us), gap to the end of the previous dispatch (us), grid, kernel.  CSV on stdout.
usage: python tools/trace_dump.py <dir> [<first> [<count>]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import _db, _short  # noqa: E402


def main():
    c = _db(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    grid = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    q = f"select start, end, name{', ' + grid if grid else ''} from kernels order by start"
    rows = c.execute(q).fetchall()
    if not rows:
        return
    t0, prev_end = rows[0][0], rows[0][0]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
    print("index,start_us,dur_us,gap_us,grid,kernel")
    for i, r in enumerate(rows):
        s, e, n = r[0], r[1], r[2]
        g = r[3] if grid else ""
        if first <= i < first + count:
            print(f"{i},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.2f},{(s - prev_end) / 1e3:.2f},{g},{_short(n)}")
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main()
