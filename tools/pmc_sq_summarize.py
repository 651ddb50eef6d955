#!/usr/bin/env python3
"""gpurun_out/pmc_sq/run/.../pmc_counter_collection.csv (tools/pmc_sq.sh) -> profiles/rNN_pmc_sq_msm.csv:
one row per MSM accumulate dispatch with its counters, effective clock (GRBM_GUI_ACTIVE / 8 XCDs /
duration) and wave cycles per VALU instruction.  usage: pmc_sq_summarize.py <round tag>"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
cc = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_sq", "run", "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_sq", "run", "**", "*kernel_trace.csv"), recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt))}
rows = {}
for r in csv.DictReader(open(cc)):
    if "msm_accumulate" not in r["Kernel_Name"]:
        continue
    d = rows.setdefault(r["Dispatch_Id"], {"Dispatch_Id": r["Dispatch_Id"],
                                           "Kernel": r["Kernel_Name"].split("(")[0].replace("void zk::", ""),
                                           "VGPR_Count": r.get("VGPR_Count", ""), "Scratch_Size": r.get("Scratch_Size", "")})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
names = ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
         "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES"]
out = os.path.join(ROOT, "profiles", f"{tag}_pmc_sq_msm.csv")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Dispatch_Id", "Kernel", "Duration_ms", "VGPR_Count", "Scratch_Size"] + names +
               ["clock_GHz", "wave_cycles_per_valu"])
    for k in sorted(rows, key=int):
        d = rows[k]
        ms = dur.get(k, 0.0)
        clk = d.get("GRBM_GUI_ACTIVE", 0) / 8 / (ms * 1e6) if ms else 0
        wpv = d.get("SQ_WAVE_CYCLES", 0) / d["SQ_INSTS_VALU"] if d.get("SQ_INSTS_VALU") else 0
        w.writerow([k, d["Kernel"], f"{ms:.6f}", d["VGPR_Count"], d["Scratch_Size"]] +
                   [int(d.get(n, 0)) for n in names] + [f"{clk:.3f}", f"{wpv:.3f}"])
print(open(out).read())
