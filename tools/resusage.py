#!/usr/bin/env python3
"""Print VGPR / spill / scratch / occupancy per kernel of a .hip file (gfx950 device pass only)."""
import re
import subprocess
import sys

out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950",
                      "--offload-device-only", "-mllvm", "-pragma-unroll-threshold=1000000",
                      "-Rpass-analysis=kernel-resource-usage", "-c",
                      sys.argv[1], "-o", "/dev/null"], capture_output=True, text=True).stderr
rows, cur = [], None
keys = {"VGPRs": "vgpr", "VGPR Spill": "spill", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occ", "TotalSGPRs": "sgpr", "LDS Size [bytes/block]": "lds"}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for k, short in keys.items():
        m = re.search(r"\s" + re.escape(k) + r": (\d+)", line)
        if m and cur is not None:
            cur[short] = m.group(1)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"zk::Fp<zk::(F.)Params>", r"\1", name)[:100]
    print(f"{name:100s} vgpr={r.get('vgpr')} sgpr={r.get('sgpr')} spill={r.get('spill')} "
          f"scratch={r.get('scratch')} occ={r.get('occ')}")
