#!/bin/bash
# Per-kernel register / scratch summary of the HIP unit (one line per kernel): name, SGPRs, VGPRs, spilled VGPRs, scratch bytes, occupancy.
cd "$(dirname "$0")/../rtcuda_amd/csrc" && make resource-usage 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
    cur[k] = v
    if k.startswith("LDS"):
        import subprocess
        name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"{name:70s} sgpr {cur.get(\"TotalSGPRs\"):>4} vgpr {cur.get(\"VGPRs\"):>4} spill {cur.get(\"VGPRs Spill\"):>3} scratch {cur.get(\"ScratchSize [bytes/lane]\"):>4} occ {cur.get(\"Occupancy [waves/SIMD]\"):>2}")
'
