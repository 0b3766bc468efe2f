"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / avg / min / max / total.

    python tools/rocpd_summary.py gpurun_out/prof_r1/r1_results.db [skip_first_n_dispatches]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    m = re.match(r"(ens_gemm_kernel<[^>]*>)", name)
    if m:
        modes = {"true, true": "NT", "true, false": "NN", "false, false": "TN"}
        epi = {"0": "store", "1": "bias", "2": "bias+relu", "3": "relu-mask", "4": "adam", "5": "grad"}
        a = re.match(r"ens_gemm_kernel<(\w+), (\w+), (\d)>", m.group(1))
        if a:
            return f"ens_gemm<{modes.get(a.group(1) + ', ' + a.group(2), '?')},{epi.get(a.group(3), a.group(3))}>"
    return name.split("(")[0][:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {namecol}, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = rows[skip:]
    agg = {}
    for name, s, e, gx, gy, gz, wx in rows:
        k = short(name) + f" grid={gx // max(wx,1)}x{gy}x{gz}"
        a = agg.setdefault(k, [0, 0, 10**18, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"| kernel (grid in workgroups) | calls | avg us | min us | max us | total ms | % |")
    print("|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {a[0]} | {a[1]/a[0]/1e3:.2f} | {a[2]/1e3:.2f} | {a[3]/1e3:.2f} | {a[1]/1e6:.2f} | {100*a[1]/tot:.1f} |")
    span = rows[-1][2] - rows[0][1]
    print(f"\nkernel time total {tot/1e6:.2f} ms over a {span/1e6:.2f} ms span ({100*tot/span:.1f}% busy), {len(rows)} dispatches")


if __name__ == "__main__":
    main()
