"""Write profiles/traffic.json (what bench.py's `roofline.traffic` reports) from the two offline PMC passes of the bench
command:

    python tools/traffic_record.py <FETCH_SIZE results.db> <WRITE_SIZE results.db> "<taken_at>" "<evidence path>"

HBM bytes per launch of the chained kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB units; the doubling is the guide's gfx950
correction for 16-byte streaming reads), averaged over the kernel's dispatches, together with a hash of the kernel sources
the passes ran on (bench.kernel_source_hash): a bench line printed from other sources reports `traffic: null`."""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def avg(db_path, counter, pat="fused_chain_pc_kernel"):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall()
    tot, disp = 0.0, set()
    for k, c, v, d in rows:
        if pat in k and c == counter:
            tot += v
            disp.add(d)
    return tot / max(len(disp), 1), len(disp)


def main():
    import bench
    fetch, nf = avg(sys.argv[1], "FETCH_SIZE")
    write, nw = avg(sys.argv[2], "WRITE_SIZE")
    rec = {"kernel": "fused_chain_pc_kernel", "fetch_kib": round(fetch), "write_kib": round(write),
           "bytes": int((2 * fetch + write) * 1024), "dispatches": [nf, nw], "taken_at": sys.argv[3],
           "evidence": sys.argv[4], "source_hash": bench.kernel_source_hash(), "sources": list(bench.TRAFFIC_SOURCES)}
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
