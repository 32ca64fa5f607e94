#!/usr/bin/env python
"""Per-kernel mean of every counter in a rocprofv3 --pmc counter_collection.csv (one or more files)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
grid = {}
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"][:60]
            if "lo_" not in k:
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
            grid[k] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["LDS_Block_Size"]), int(r["VGPR_Count"]))
for k, cs in acc.items():
    g = grid[k]
    waves = g[0] // 64
    print(f"{k}  grid={g[0]} wg={g[1]} lds={g[2]} vgpr={g[3]} waves={waves}")
    for c, (v, n) in sorted(cs.items()):
        print(f"   {c:32s} {v / n:16.0f}   per wave {v / n / waves:12.1f}")
