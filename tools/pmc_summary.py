"""Per-kernel averages of the counters in a rocprofv3 --pmc (rocpd sqlite) run.

    python tools/pmc_summary.py gpurun_out/pmc_x/x_results.db [kernel-name-substring]
"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = db.execute("select kernel_name, grid_size, workgroup_size, counter_name, value, dispatch_id "
                      "from counters_collection").fetchall()
    agg, names = {}, []
    for k, grid, wg, c, v, d in rows:
        if pat not in k:
            continue
        k = re.sub(r"\(anonymous namespace\)::|void ", "", k).split("(")[0][:60] + f" wg={grid // max(wg, 1)}"
        a = agg.setdefault(k, {})
        s = a.setdefault(c, [0.0, set()])
        s[0] += v
        s[1].add(d)
        if c not in names:
            names.append(c)
    print("| kernel | dispatches | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    for k, a in agg.items():
        n = max(len(s[1]) for s in a.values())
        print(f"| {k} | {n} | " + " | ".join(f"{a[c][0] / len(a[c][1]):.0f}" if c in a else "-" for c in names) + " |")


if __name__ == "__main__":
    main()
