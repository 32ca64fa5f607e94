#!/usr/bin/env python
"""Turn two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md prescribes) into
profiles/<tag>_traffic.json: HBM bytes per launch and per kernel.

gfx950 corrections (guide, §HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly half of the bytes
of wide (16 B/lane) coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    out = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"]
                out[k][0] += float(r["Counter_Value"])
                out[k][1] += 1
    return out


def short(name):
    for key in ("lo_wgrad3x3_mt", "lo_wgrad_tn", "lo_igemm_nt", "lo_conv3x3_pp", "lo_t_attn_folded", "lo_bn_apply", "lo_gn_bwd_apply", "lo_gn_bwd_reduce", "lo_gn_fwd", "lo_adamw", "lo_wgrad_reduce"):
        if key in name:
            return key
    return name.split("(")[0][:40]


def main():
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    agg = collections.defaultdict(lambda: {"fetch_bytes": 0.0, "write_bytes": 0.0, "launches": 0})
    for k, (v, n) in fe.items():
        a = agg[short(k)]
        a["fetch_bytes"] += 2.0 * v * 1024.0          # KiB -> bytes, x2 gfx950 correction
        a["launches"] += n
    for k, (v, n) in wr.items():
        agg[short(k)]["write_bytes"] += v * 1024.0
    res = {}
    for k, a in agg.items():
        n = max(a["launches"], 1)
        res[k] = {"hbm_bytes_per_launch": (a["fetch_bytes"] + a["write_bytes"]) / n, "fetch_bytes_per_launch": a["fetch_bytes"] / n,
                  "write_bytes_per_launch": a["write_bytes"] / n, "launches_profiled": a["launches"]}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), KiB units", "kernels": res},
              open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{k:28s} {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch  (fetch {v['fetch_bytes_per_launch'] / 1e6:.2f}, write {v['write_bytes_per_launch'] / 1e6:.2f})")


if __name__ == "__main__":
    main()
