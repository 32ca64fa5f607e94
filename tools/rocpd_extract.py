#!/usr/bin/env python
"""Read rocprofv3's rocpd SQLite output (the default output format of ROCm 7.2) and write what profiles/ keeps:

  rocpd_extract.py stats <results.db> <out.csv>          per-kernel Calls / total / average / min / max (ns) of a
                                                         `rocprofv3 --kernel-trace --stats` run, largest total first
  rocpd_extract.py traffic <fetch.db> <write.db> <out.json>
      HBM bytes per launch and kernel from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE: separate passes, as
      MI355X_MICROARCH.md prescribes).  gfx950 corrections from the same guide: both counters are in KiB; FETCH_SIZE reports
      half of the bytes of wide (16 B per lane) coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-byte stores.
"""
import collections
import csv
import json
import sqlite3
import sys

FAMILIES = ("lo_wgrad3x3_mt", "lo_wgrad_tn", "lo_igemm_nt", "lo_conv3x3_pp", "lo_t_attn_folded", "lo_bn_apply", "lo_gn_bwd_apply",
            "lo_gn_bwd_reduce", "lo_gn_fwd", "lo_adamw", "lo_wgrad_reduce", "lo_final_conv_bwd", "lo_final_conv_fwd")


def family(name):
    for key in FAMILIES:
        if key in name:
            return key
    return name.split("(")[0][:48]


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                     "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, n, tot, avg, mn, mx in rows:
            w.writerow([name, n, tot, round(avg, 3), round(100.0 * tot / total, 2), mn, mx])


def counter(db, which):
    c = sqlite3.connect(db)
    out = collections.defaultdict(lambda: [0.0, 0])
    for name, val in c.execute("select name, counter_value from pmc_events where counter_name = ?", (which,)):
        out[name][0] += float(val)
        out[name][1] += 1
    return out


def traffic(fetch_db, write_db, out):
    fe, wr = counter(fetch_db, "FETCH_SIZE"), counter(write_db, "WRITE_SIZE")
    agg = collections.defaultdict(lambda: {"fetch_bytes": 0.0, "write_bytes": 0.0, "launches": 0})
    for k, (v, n) in fe.items():
        a = agg[family(k)]
        a["fetch_bytes"] += 2.0 * v * 1024.0
        a["launches"] += n
    for k, (v, n) in wr.items():
        agg[family(k)]["write_bytes"] += v * 1024.0
    res = {}
    for k, a in agg.items():
        if a["launches"]:
            res[k] = {"launches": a["launches"], "fetch_bytes_per_launch": a["fetch_bytes"] / a["launches"],
                      "write_bytes_per_launch": a["write_bytes"] / a["launches"],
                      "hbm_bytes_per_launch": (a["fetch_bytes"] + a["write_bytes"]) / a["launches"]}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), KiB units",
               "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))},
              open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)
