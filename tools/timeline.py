#!/usr/bin/env python
"""Timeline of one training step from a rocprofv3 --kernel-trace rocpd database: every kernel dispatch between two consecutive
gradient-norm finalize launches, with start / end relative to the first one and the queue (stream) it ran on.  Shows which stream finishes the
backward last, i.e. what the step's critical path is.   python tools/timeline.py <results.db> [step_index_from_end]"""
import sqlite3, sys
db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = next(t for t in tabs if t == "kernels" or t.endswith("kernels"))
cols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
sys.stderr.write(f"table {kt} columns {cols}\n")
qcol = next((x for x in ("queue_id", "stream_id", "queue", "stream") if x in cols), None)
rows = c.execute(f"select name, start, end, {qcol or 0} from {kt} order by start").fetchall()
# one lo_gradnorm_finalize launch per step (the AdamW of a pipelined step is several launches): the step boundary
bounds = [i for i, r in enumerate(rows) if "lo_gradnorm_finalize" in r[0]]
lo, hi = bounds[-back - 1], bounds[-back]
t0 = rows[lo][1]
for name, s, e, q in rows[lo:hi]:
    short = name.split("(")[0].replace("void ", "")[:58]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  q{q}  {short}")
print(f"step span {(rows[hi][1] - t0) / 1e3:.1f} us, {hi - lo} dispatches")
