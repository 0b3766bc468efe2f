"""(needs the LAB build of the library: ./build.sh --lab)
Per-workgroup (start, end) of the two launches of a critic update (ssac_debug_timeline, s_memrealtime at 100 MHz):
when does each workgroup start (dispatch skew), which class finishes last.      python tools/wg_timeline.py [B] [N]"""
import os, sys
os.environ.setdefault("SSAC_LAB_BUILD", "1")   # the lab library (./build.sh --lab -> libssac_hip_lab.so)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv, args = sys.argv[:1] + ["__none__"], sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(ROOT, "tools", "bench_configs.py"))
bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
import numpy as np
import torch
import super_sac_amd as ssa
B = int(args[0]) if args else 512
N = int(args[1]) if len(args) > 1 else 10
ssa.learning.USE_GRAPHS = False
critic, _ = bc.build(17, 6, B, N, 2)
for _ in range(5):
    critic()
tl = torch.zeros(2048, dtype=torch.int64, device="cuda")
ssa._lib.check(ssa._lib.lib.ssac_debug_timeline(tl.data_ptr()))
for rep in range(3):
    tl.zero_()
    critic()
    torch.cuda.synchronize()
    t = tl.cpu().numpy().reshape(2, 512, 2) * 10e-3   # us
    for name, a, classes in (("chained launch", t[0], None), ("weight-gradient launch", t[1], None)):
        n = int((a[:, 1] > 0).sum())
        a = a[:n]
        t0 = a[:, 0].min()
        s, e = a[:, 0] - t0, a[:, 1] - t0
        print(f"[{rep}] {name}: {n} workgroups; starts {s.min():.2f} .. {s.max():.2f} us (median {np.median(s):.2f}); "
              f"ends {e.min():.2f} .. {e.max():.2f}; durations min {np.min(e - s):.2f} median {np.median(e - s):.2f} max {np.max(e - s):.2f}")
        if rep == 2:
            if name.startswith("chained"):
                # producer / consumer form (fused_chain_pc_kernel<32>): producers, 32-row critic tiles, consumers; with 16-row
                # critic tiles: producers, consumers, critic tiles (tools/r5/co_timeline.py knows the co-resident form too)
                T = (B + 15) // 16
                nc32 = N * ((B + 31) // 32)
                if n in (3 * T + nc32, 3 * T + nc32 + 1):
                    groups = (("producers", 0, T), ("critic tiles", T, T + nc32), ("consumers", T + nc32, 3 * T + nc32))
                else:
                    groups = (("producers", 0, T), ("consumers", T, 3 * T), ("critic tiles", 3 * T, n))
            else:
                groups = (("fc2 tiles", 0, 16 * N), ("fc1 tiles", 16 * N, 20 * N), ("head", 20 * N, 24 * N), ("TD", 24 * N, 24 * N + 1))
            if not name.startswith("chained"):
                # the GEMM / head workgroups take their tile in XCD-contiguous order (ssac_internal.h): sort by LOGICAL id
                nm = n - 1
                q, r = nm >> 3, nm & 7
                b = np.arange(nm)
                x, slot = b & 7, b >> 3
                L = np.where(x < r, x * (q + 1), r * (q + 1) + (x - r) * q) + slot
                t0, t01, ht = 16 * N, 20 * N, 4 * N
                if t0 % 8 == 0 and (t01 - t0) % 8 == 0 and ht % 8 == 0:   # per-class order (ssac_gemm.hip, xcd_mix)
                    a8, f8, h8 = t0 // 8, (t01 - t0) // 8, ht // 8
                    L = np.where(slot < a8, x * a8 + slot,
                                 np.where(slot < a8 + f8, t0 + x * f8 + (slot - a8), t01 + x * h8 + (slot - a8 - f8)))
                order = np.concatenate([np.argsort(L), [nm]])
                s, e = s[order], e[order]
                d = (e - s)[:16 * N].reshape(N, 16)
                print("      fc2 tile durations by net (rows) x tile (columns):")
                for row in d:
                    print("       ", " ".join(f"{v:5.1f}" for v in row))
            for gname, lo, hi in groups:
                hi = min(hi, n)
                if hi > lo:
                    print(f"      {gname:14s} start {s[lo:hi].min():6.2f}..{s[lo:hi].max():6.2f}  end {e[lo:hi].min():6.2f}..{e[lo:hi].max():6.2f}  "
                          f"duration {np.min((e - s)[lo:hi]):6.2f}..{np.max((e - s)[lo:hi]):6.2f} (median {np.median((e - s)[lo:hi]):6.2f})")
    w0 = t[1][:, 0][t[1][:, 1] > 0].min()
    print(f"[{rep}] chained launch first start -> weight-gradient launch first start: {w0 - t[0][:, 0][t[0][:, 1] > 0].min():.2f} us; "
          f"chained last end -> wgrad first start: {w0 - t[0][:, 1].max():.2f} us")
ssa._lib.lib.ssac_debug_timeline(0)
