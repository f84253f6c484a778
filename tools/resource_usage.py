#!/usr/bin/env python3
"""Per-kernel register / scratch summary of the HIP unit, one line per kernel (wraps `make resource-usage`)."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rtcuda_amd", "csrc")
KEYS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "VGPRs Spill": "spill", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds"}


def main():
    out = subprocess.run(["make", "-C", CSRC, "resource-usage"] + (["DEFS=" + os.environ["DEFS"]] if os.environ.get("DEFS") else []), capture_output=True, text=True)
    text = out.stdout + out.stderr
    cur = None
    rows = []
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?: \[[^\]]+\])?): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None and k in KEYS:
            cur[KEYS[k]] = v
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for r, n in zip(rows, names):
        n = n.split("(")[0].replace("void ", "")
        if pat and not re.search(pat, n):
            continue
        print(f"{n:64s} sgpr {r.get('sgpr', '?'):>4} vgpr {r.get('vgpr', '?'):>4} spill {r.get('spill', '?'):>3} "
              f"scratch {r.get('scratch', '?'):>4} occ {r.get('occ', '?'):>2} lds {r.get('lds', '?')}")


if __name__ == "__main__":
    main()
