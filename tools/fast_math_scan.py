import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from rtcuda_amd import api
for kind, name in ((0, "rcp"), (1, "sqrt")):
    # by exponent range: report mismatch counts per sign/exponent byte
    tot = 0
    bad_ranges = []
    for hi in range(256):  # top 8 bits of the pattern: sign + 7 exponent bits
        n, ex = api.test_fast_math(kind, hi << 24, 1 << 24)
        tot += n
        if n:
            bad_ranges.append((hex(hi << 24), n, [hex(int(e)) for e in ex[:3]]))
    print(name, "total mismatches", tot)
    for r in bad_ranges:
        print("   ", r)
