"""print the kernel timeline (start offset, duration, name) of a few consecutive updates from a rocprofv3 trace

    python tools/timeline.py <results.db> [first_dispatch] [count]
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
namecol = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {namecol}, start, end, grid_x, grid_y, workgroup_x, queue_id from kernels order by start").fetchall()
rows = rows[first:first + count]
t0 = rows[0][1]
prev_end = t0
for name, s, e, gx, gy, wx, q in rows:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", name).split("(")[0][:58]
    print(f"{(s - t0) / 1e3:9.2f} us  +{(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:7.2f}  q{q}  {name} [{gx // max(wx, 1)}x{gy}]")
    prev_end = max(prev_end, e)
