"""Per-dispatch durations (us) of the named kernels from a rocprofv3 --kernel-trace result directory, in launch order.
usage: python tools/trace_series.py <dir> <kernel> [<kernel> ...]"""
import sys

from prof_summary import _db


def main():
    c = _db(sys.argv[1])
    for n in sys.argv[2:]:
        ser = [round((e - s) / 1e3, 1) for s, e in c.execute("select start, end from kernels where name like ? order by start", (f"%{n}(%",))]
        print(n, len(ser), ser)


if __name__ == "__main__":
    main()
