#!/usr/bin/env python3
"""gpurun_out/pmc/{FETCH_SIZE,WRITE_SIZE}/pmc_counter_collection.csv (tools/pmc_collect.sh) ->
profiles/rNN_pmc_summary.json + the per-dispatch rows of the MSM kernels as profiles/rNN_pmc_*.csv.
usage: pmc_summarize.py <round tag, e.g. r02> <bench json written by the same pass>"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, bench_json = sys.argv[1], sys.argv[2]
bench = json.load(open(bench_json))
out = {"source": f"profiles/{tag}_pmc_fetch_size.csv + {tag}_pmc_write_size.csv (rocprofv3 --pmc "
                 "FETCH_SIZE / WRITE_SIZE in separate passes, tools/pmc_collect.sh: bench.py --steps 1 "
                 "--warmup 1 --cpu-sample 0 --no-pipeline --worst-case-steps 0); last four launches "
                 "of the G1 comb kernel = the timed step (A, B1, K, Z)",
       "note": "counter values x 1024 (the counters are in KB).  MI355X_MICROARCH.md (HBM section): "
               "FETCH_SIZE under-counts wide coalesced streams by 2x and is uncalibrated for other "
               "access widths -> calibrate on a known byte count in the kernel's own pattern.  This "
               "kernel's pattern is one 64-byte table entry per lane; the Z launch is the "
               "calibration: 3450 groups x 255 windows x 1024 proofs x 64 B = 57.6 GB gathered, "
               "FETCH_SIZE reports 56.3-57.7 GB (a few entries repeat within a wave) -> factor 1.0 "
               "for these gathers, no doubling applied.  WRITE_SIZE is exact for 16-byte-per-lane "
               "stores (the partial sums)",
       "batch": bench["config"]["batch_per_gpu"], "levels": 160, "populated": 10,
       "g1_windows": bench["config"]["msm_window_tables"]["g1_windows"],
       "g1_comb_k": bench["config"]["msm_window_tables"]["g1_comb_k"]}
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(ROOT, "gpurun_out", "pmc", c, "pmc_counter_collection.csv")
    rows = [r for r in csv.DictReader(open(path)) if "msm_accumulate" in r["Kernel_Name"]
            and r["Counter_Name"] == c]
    keep = os.path.join(ROOT, "profiles", f"{tag}_pmc_{c.lower()}.csv")
    with open(keep, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name",
                                          "Counter_Value"])
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in w.fieldnames})
    g1 = [float(r["Counter_Value"]) for r in rows if "Fq2" not in r["Kernel_Name"]
          and "accumulate_comb" in r["Kernel_Name"]][-4:]
    g2 = [float(r["Counter_Value"]) for r in rows if "Fq2" in r["Kernel_Name"]
          and "accumulate_comb" in r["Kernel_Name"]][-1:]
    per[c] = (g1, g2)
out["msm_g1_fetch_kb_per_launch"] = per["FETCH_SIZE"][0]
out["msm_g1_write_kb_per_launch"] = per["WRITE_SIZE"][0]
out["msm_g1_bytes_per_launch"] = (sum(per["FETCH_SIZE"][0]) + sum(per["WRITE_SIZE"][0])) * 1024 / 4
out["msm_g2_fetch_kb"] = per["FETCH_SIZE"][1]
out["msm_g2_write_kb"] = per["WRITE_SIZE"][1]
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
