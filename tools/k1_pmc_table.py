"""Joins the PMC summaries tools/k1_pmc.sh wrote into one table for the search kernels: duration (us, kernel trace), waves,
VALU instructions per wave, lane utilisation of the VALU instructions (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): both in
quad-cycles, MI355X_MICROARCH.md), share of a wave's resident time it issues VALU work / waits (SQ_ACTIVE_INST_VALU, SQ_WAIT_INST_ANY over
SQ_WAVE_CYCLES), the SIMDs' VALU busy fraction over the launch (4 x SQ_ACTIVE_INST_VALU / (duration x 2.4 GHz x 1024 SIMDs)),
vector-memory read instructions per wave, FETCH / WRITE KB per launch (uncorrected).
usage: python tools/k1_pmc_table.py gpurun_out/<tag>"""
import csv
import json
import os
import sys

d = sys.argv[1]
KERNELS = ["k_slots_coarse2", "k_match_hard2", "k_match_ball2", "k_match_tile", "k_match_overflow2", "k_match_cert_inline2", "k_match_cert2", "k_match_list2", "k_b_match"]
out = {"source": "tools/k1_pmc.sh: rocprofv3 --kernel-trace --pmc <group> -- python3 tools/schedule_probe.py 200000 3 (sched) / bench.py --workload locations "
                 "--reps 2 (loc); averages over each kernel's last 40 dispatches; one counter group per pass", "kernels": {}}
if os.path.exists(os.path.join(d, "code_stamp.txt")):
    out["code_stamp"] = open(os.path.join(d, "code_stamp.txt")).read().strip()
for w in ("sched", "loc"):
    try:
        u1 = json.load(open(f"{d}/{w}_util1_summary.json")); u2 = json.load(open(f"{d}/{w}_util2_summary.json"))
        fe = json.load(open(f"{d}/{w}_fetch_summary.json")); wr = json.load(open(f"{d}/{w}_write_summary.json"))
    except FileNotFoundError:
        continue
    dur = {}
    for line in open(f"{d}/{w}_kernels.csv"):
        if line.startswith("#") or line.startswith("kernel"):
            if line.startswith("# last"):
                break
            continue
        p = line.strip().split(",")
        if len(p) >= 3:
            dur[p[0]] = float(p[2])
    for k in KERNELS:
        if k not in u1:
            continue
        a, b = u1[k], u2.get(k, {})
        g = lambda dd, key: dd.get(key + "_avg_last40")
        waves, insts, active, wcyc = g(a, "SQ_WAVES"), g(a, "SQ_INSTS_VALU"), g(a, "SQ_ACTIVE_INST_VALU"), g(a, "SQ_WAVE_CYCLES")
        wait, vmem, tcyc = g(b, "SQ_WAIT_INST_ANY"), g(b, "SQ_INSTS_VMEM_RD"), g(b, "SQ_THREAD_CYCLES_VALU")
        row = {"workload": w, "avg_us": dur.get(k), "dispatches": a.get("SQ_WAVES_dispatches"), "waves": waves,
               "valu_insts_per_wave": insts / waves if waves else None,
               "valu_lane_utilisation": (tcyc / (64.0 * active)) if (tcyc and active) else None,
               "valu_share_of_wave_time": (active / wcyc) if (active and wcyc) else None,
               "simd_valu_busy_frac": (4.0 * active / (dur[k] * 1e-6 * 2.4e9 * 1024)) if (active and dur.get(k)) else None,
               "wait_share_of_wave_time": (wait / wcyc) if (wait and wcyc) else None,
               "vmem_reads_per_wave": vmem / waves if (vmem and waves) else None,
               "FETCH_KB": g(fe.get(k, {}), "FETCH_SIZE"), "WRITE_KB": g(wr.get(k, {}), "WRITE_SIZE")}
        out["kernels"][f"{k} ({w})"] = row
json.dump(out, sys.stdout, indent=1)
print()
