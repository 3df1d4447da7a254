#!/usr/bin/env python3
"""Copies one measurement set from gpurun_out/<tag>/ (written by the round's measurement script:
bench.json, kernel_stats.csv, pmc_fetch_summary.json, pmc_write_summary.json) into profiles/ under
<name>_* and refreshes profiles/pmc_traffic.json, which bench.py reports as `roofline.traffic`.

  python tools/write_profiles.py <tag> <name>
"""
import json
import sys

tag, out = sys.argv[1], sys.argv[2]
f = json.load(open(f"gpurun_out/{tag}/pmc_fetch_summary.json"))
w = json.load(open(f"gpurun_out/{tag}/pmc_write_summary.json"))
merged = {k: {"FETCH_SIZE_KB_avg_last20": f[k].get("FETCH_SIZE_avg_last20"),
              "WRITE_SIZE_KB_avg_last20": w.get(k, {}).get("WRITE_SIZE_avg_last20"),
              "dispatches": f[k].get("FETCH_SIZE_dispatches")} for k in f}
json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --no-cpu-baseline --steps 10 ; same with "
                      "--pmc WRITE_SIZE (separate passes); summarised by tools/prof_summary.py pmc",
           "unit": "KB as reported (FETCH_SIZE uncorrected)", "kernels": merged},
          open(f"profiles/{out}_pmc_fetch_write_per_kernel.json", "w"), indent=1)
k1 = ["k_match_ball2", "k_match_hard2"]
fe = sum(f[k]["FETCH_SIZE_avg_last20"] for k in k1) / 2
wr = sum(w[k]["WRITE_SIZE_avg_last20"] for k in k1) / 2
traffic = (2 * fe + wr) * 1024
json.dump({
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --no-cpu-baseline --steps 10, "
              f"finest-level launches (last 20 dispatches per kernel); profiles/{out}_pmc_fetch_write_per_kernel.json",
    "kernel": "K1 = (k_match_ball2 + k_match_hard2) / 2: every launch handles both find_closest_euclidean directions "
              "(200k queries against 200k features each); figures are per direction",
    "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr,
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is; unit KB",
    "traffic_bytes_per_launch": traffic, "traffic_bytes_per_launch_uncorrected": (fe + wr) * 1024},
    open("profiles/pmc_traffic.json", "w"), indent=1)
b = json.load(open(f"gpurun_out/{tag}/bench.json"))
r = b["roofline"]
r["traffic"] = traffic
r["traffic_GBs"] = traffic / (r["kernel_ms"] * 1e-3) / 1e9
r["traffic_frac"] = r["traffic_GBs"] / 8000.0
b["_note"] = ("traffic fields recomputed with the PMC passes of the same measurement set (profiles/pmc_traffic.json); "
              "the line as printed carried the previous set's figure")
json.dump(b, open(f"profiles/{out}_bench.json", "w"))
with open(f"profiles/{out}_kernel_stats.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -d ... -- python3 bench.py --no-cpu-baseline   (MI355X, 200k x 200k, rrl symmetric "
            "dof 12; summarised by tools/prof_summary.py kernels)\n")
    o.write(open(f"gpurun_out/{tag}/kernel_stats.csv").read())
print(f"{out}: ms/step {b['ms_per_step']:.4f}  pairs/s {b['value']:.3e}  K1 ms {r['kernel_ms']:.4f}  frac {r['frac']:.4f}  "
      f"traffic {traffic / 1e6:.1f} MB ({r['traffic_GBs']:.0f} GB/s)  cpu {b['cpu_baseline']['ms_per_iteration']:.0f} ms/it  "
      f"x{b['speedup_vs_cpu_1thread']:.0f}")
