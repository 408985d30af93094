#!/usr/bin/env python3
"""print the kernel launches of the LAST step (from its natac_frag_gather on) of a rocprofv3 --kernel-trace database
(rocpd sqlite): name, start relative to the gather, duration, queue / stream -- shows which launches overlap.
usage: python tools/trace_last_step.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
i0 = [i for i, r in enumerate(rows) if "frag_gather" in r[0]][-1]
t0 = rows[i0][1]
for name, a, b, q, st in rows[i0:]:
    name = name.split("(")[0].replace("void ", "").replace("natac::", "")
    print("%-44s start %8.3f ms   %8.3f ms   queue %s stream %s" % (name[:44], (a - t0) / 1e6, (b - a) / 1e6, q, st))
