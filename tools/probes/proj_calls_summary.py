#!/usr/bin/env python
"""Per-kernel durations of tools/probes/proj_calls.py from the rocprofv3 database, split by call kind
(FlowProjection without hole filling: the 150 warm-up calls; with: the next N; DepthFlowProjection with: the last N)."""
import re
import sqlite3
import sys
from collections import defaultdict

db, warm = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 150
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end, duration from kernels order by start").fetchall()
acc = defaultdict(list)
gaps = defaultdict(list)
seen, phase, prev_end = 0, None, None
for name, start, end, dur in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "")
    if "proj_owner5<false" in short:
        seen += 1
        phase = "FlowProjection fill=0" if seen <= warm else "FlowProjection fill=1"
        first = True
    elif "proj_owner5<true" in short:
        phase = "DepthFlowProjection fill=1"
        first = True
    elif "proj_" not in short:
        continue
    else:
        first = False
    acc[(phase, short)].append(dur / 1e3)
    if not first and prev_end is not None:
        gaps[(phase, short)].append((start - prev_end) / 1e3)
    prev_end = end
for (ph, k), v in sorted(acc.items()):
    v2 = v[len(v) // 3:]                                    # (the first third of a phase is still settling)
    g = gaps.get((ph, k))
    print("%-28s %-62s n=%4d  avg %8.2f us  min %8.2f%s" % (ph, k[:62], len(v), sum(v2) / len(v2), min(v),
          ("   gap before: %.2f us" % (sum(g) / len(g))) if g else ""))
