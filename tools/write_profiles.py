#!/usr/bin/env python3
"""Copies one measurement set from gpurun_out/<tag>/ (written by tools/measure.sh) into profiles/ under <name>_* and refreshes
profiles/pmc_traffic.json and profiles/pmc_utilisation.json, which bench.py reports as `roofline.traffic` / `roofline.valu_frac`
when their code stamp (sha256 of the kernel sources, bench.code_stamp) matches the running code.

  python tools/write_profiles.py <tag> <name>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag, out = sys.argv[1], sys.argv[2]
G = f"gpurun_out/{tag}"
b = json.load(open(f"{G}/bench.json"))
sha = bench.code_stamp()
if b.get("code_sha") != sha:
    sys.exit(f"the measurement set was taken on code {b.get('code_sha')}, the tree holds {sha}: re-measure")
f = json.load(open(f"{G}/pmc_fetch_summary.json"))
w = json.load(open(f"{G}/pmc_write_summary.json"))
u = json.load(open(f"{G}/pmc_util_summary.json"))
u2 = json.load(open(f"{G}/pmc_util2_summary.json"))
merged = {k: {"FETCH_SIZE_KB_avg_last20": f[k].get("FETCH_SIZE_avg_last20"),
              "WRITE_SIZE_KB_avg_last20": w.get(k, {}).get("WRITE_SIZE_avg_last20"),
              "dispatches": f[k].get("FETCH_SIZE_dispatches")} for k in f}
json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --no-cpu-baseline --no-schedule --reps 1 --steps 40 ; same with "
                      "--pmc WRITE_SIZE (separate passes); summarised by tools/prof_summary.py pmc",
           "code_sha": sha, "unit": "KB as reported (FETCH_SIZE uncorrected)", "kernels": merged},
          open(f"profiles/{out}_pmc_fetch_write_per_kernel.json", "w"), indent=1)
# the kernels of a steady finest-level search: the certificate kernel that searches its own failures (k_match_cert_inline2); the
# separate certificate / list / wave-per-query kernels run only while the lists are long (the first searches of the level), so a
# kernel is counted only when it was dispatched about as often as the most frequent one
if f.get("k_match_cert_inline2", {}).get("FETCH_SIZE_dispatches", 0) >= 5:
    k1 = ["k_match_cert_inline2"]  # the steady finest-level search is this one kernel (k_match_hard2 also serves the coarse levels)
else:
    cand = [k for k in ("k_match_cert2", "k_match_list2", "k_match_hard2") if k in f]
    most = max(f[k].get("FETCH_SIZE_dispatches", 0) for k in cand)
    k1 = [k for k in cand if f[k].get("FETCH_SIZE_dispatches", 0) * 2 >= most]
fe = sum(f[k]["FETCH_SIZE_avg_last20"] for k in k1) / 2
wr = sum(w[k]["WRITE_SIZE_avg_last20"] for k in k1) / 2
traffic = (2 * fe + wr) * 1024
json.dump({
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --no-cpu-baseline --no-schedule --reps 1 --steps 40, "
              f"finest-level launches (last 20 dispatches per kernel); profiles/{out}_pmc_fetch_write_per_kernel.json",
    "code_sha": sha,
    "kernel": "K1 = (" + " + ".join(k1) + ") / 2: every launch handles both find_closest_euclidean directions "
              "(200k queries against 200k features each); figures are per direction",
    "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr,
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is; unit KB. "
                  "The doubling is stated for wide streaming reads; the certificate kernel streams 32-B source records and gathers 32-B "
                  "target points, so the true figure lies between the uncorrected and the corrected one",
    "traffic_bytes_per_launch": traffic, "traffic_bytes_per_launch_uncorrected": (fe + wr) * 1024},
    open("profiles/pmc_traffic.json", "w"), indent=1)
# utilisation of the K1 kernels: VALU issue time against the SIMD time of the launch
lines = [l.strip().split(",") for l in open(f"{G}/kernel_stats.csv") if l.strip() and not l.startswith("#")]
dur = {}
for l in lines:
    if len(l) >= 3 and l[0] in k1:
        try:
            dur[l[0]] = float(l[2])  # avg_us, whole-run table first, then the steady-state table overrides it
        except ValueError:
            pass
util = {}
for k in k1:
    a = dict(u.get(k, {}))
    a.update(u2.get(k, {}))
    util[k] = {n.replace("_avg_last20", ""): v for n, v in a.items() if n.endswith("_avg_last20")}
    util[k]["avg_us"] = dur.get(k)
main = k1[0]
valu_q = util[main].get("SQ_ACTIVE_INST_VALU", 0.0)  # quad-cycles, summed over all SIMDs
simd_cycles = dur.get(main, 0.0) * 1e-6 * 2.4e9 * 1024  # 256 CUs x 4 SIMDs at 2.4 GHz
valu_frac = (4.0 * valu_q / simd_cycles) if simd_cycles else None
json.dump({"source": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES | SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD "
                     "SQ_THREAD_CYCLES_VALU (two passes), bench.py --no-cpu-baseline --no-schedule --reps 1 --steps 40, last 20 dispatches per kernel; "
                     "SQ cycle counters in quad-cycles, summed over all SIMDs",
           "code_sha": sha, "kernels": util,
           "valu_frac_definition": f"4 x SQ_ACTIVE_INST_VALU (quad-cycles -> cycles) of {main} / (its average duration x 2.4 GHz x 1024 SIMDs): "
                                   "the share of the chip's VALU issue time the dominant K1 kernel uses while it runs",
           "valu_frac": valu_frac},
          open("profiles/pmc_utilisation.json", "w"), indent=1)
r = b["roofline"]
r["traffic"] = traffic
r["traffic_GBs"] = traffic / (r["kernel_ms"] * 1e-3) / 1e9
r["traffic_frac"] = r["traffic_GBs"] / 8000.0
r["valu_frac"] = valu_frac
b["_note"] = ("traffic / valu_frac fields filled from the PMC passes of the same measurement set (profiles/pmc_traffic.json, "
              "profiles/pmc_utilisation.json); the line as printed carried null for them")
json.dump(b, open(f"profiles/{out}_bench.json", "w"))
for extra in ("locations", "bench_2m", "cdc"):
    if os.path.exists(f"{G}/{extra}.json"):
        open(f"profiles/{out}_{extra}.json", "w").write(open(f"{G}/{extra}.json").read())
with open(f"profiles/{out}_kernel_stats.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -d ... -- python3 bench.py --no-cpu-baseline --no-schedule --reps 1   (MI355X, 200k x 200k, rrl symmetric "
            "dof 12; summarised by tools/prof_summary.py kernels)\n")
    o.write(open(f"{G}/kernel_stats.csv").read())
print(f"{out}: ms/step {b['ms_per_step']:.4f}  pairs/s {b['value']:.3e}  K1 ms {r['kernel_ms']:.4f}  frac {r['frac']:.4f}  "
      f"traffic {traffic / 1e6:.1f} MB ({r['traffic_GBs']:.0f} GB/s)  valu_frac {valu_frac}")
