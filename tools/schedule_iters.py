"""Per-iteration view of tools/schedule_trace.sh's dispatches.csv: the last run() of the trace, one line per ICP iteration:
span to the next iteration's first kernel, sum of kernel durations, launches, and every kernel with its duration (+gap before it).
usage: python tools/schedule_iters.py <dispatches.csv> [n_iterations]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    nit = int(sys.argv[2]) if len(sys.argv) > 2 else 27
    acc = [i for i, r in enumerate(rows) if 'k_f_accumulate' in r['kernel']]
    i = acc[-nit]
    firsts = ('k_slots_coarse2', 'k_match_ball2', 'k_match_cert', 'k_match_tile', 'k_match_direct', 'k_knn_cert')
    while not any(f in rows[i]['kernel'] for f in firsts):
        i -= 1
    out, cur = [], []
    for r in rows[i:]:
        k = r['kernel']
        if any(f in k for f in firsts) and cur and any('k_errors' in x['kernel'] for x in cur):
            out.append(cur)
            cur = []
        cur.append(r)
    out.append(cur)
    tot_busy = tot_span = 0.0
    for n, itr in enumerate(out):
        s = float(itr[0]['start_us'])
        e = float(itr[-1]['start_us']) + float(itr[-1]['dur_us'])
        busy = sum(float(x['dur_us']) for x in itr)
        nxt = float(out[n + 1][0]['start_us']) if n + 1 < len(out) else e
        tot_busy += busy
        tot_span += nxt - s
        print(n, f"span {nxt - s:7.1f} busy {busy:6.1f} launches {len(itr):2d}",
              ' '.join(f"{x['kernel'].replace('k_', '')[:14]}:{float(x['dur_us']):.1f}(+{float(x['gap_us']):.1f})" for x in itr))
    print(f"total span {tot_span:.1f} us, busy {tot_busy:.1f} us over {len(out)} iterations")


if __name__ == "__main__":
    main()
