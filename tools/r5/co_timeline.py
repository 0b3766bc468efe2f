"""(LAB build) per-class workgroup timeline of the chained launch, both forms:  python tools/r5/co_timeline.py [B] [N]
   SSAC_CHAIN_FORM=0: producers [0, T), 32-row critic tiles, consumers;  =1 (co-resident): producers, consumers, 16-row critic tiles"""
import os, sys
os.environ.setdefault("SSAC_LAB_BUILD", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1] + ["__none__"], sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(ROOT, "tools", "bench_configs.py"))
bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
import numpy as np
import torch
import super_sac_amd as ssa
B = int(args[0]) if args else 512
N = int(args[1]) if len(args) > 1 else 10
form = int(os.environ.get("SSAC_CHAIN_FORM", "1"))
ssa.learning.USE_GRAPHS = False
critic, _ = bc.build(17, 6, B, N, 2)
for _ in range(5):
    critic()
tl = torch.zeros(2048, dtype=torch.int64, device="cuda")
ssa._lib.check(ssa._lib.lib.ssac_debug_timeline(tl.data_ptr()))
T = (B + 15) // 16
for rep in range(3):
    tl.zero_()
    critic()
    torch.cuda.synchronize()
    t = tl.cpu().numpy().reshape(2, 512, 2) * 10e-3   # us
    a = t[0]
    n = int((a[:, 1] > 0).sum())
    a = a[:n]
    t0 = a[:, 0].min()
    s, e = a[:, 0] - t0, a[:, 1] - t0
    if n == T + 2 * T + N * T:
        cf = max(0, min(N * T, 256 - 3 * T))   # fused_chain_co_kernel: [critic tiles, first part][producers][consumers][critic tiles, rest]
        groups = (("critic16a", 0, cf), ("producers", cf, cf + T), ("consumers", cf + T, cf + 3 * T), ("critic16", cf + 3 * T, n))
    else:
        nc = n - 3 * T
        groups = (("producers", 0, T), ("critic32", T, T + nc), ("consumers", T + nc, n))
    print(f"[{rep}] form {form}: chained launch {n} workgroups, last end {e.max():.2f} us")
    for g, lo, hi in groups:
        d = (e - s)[lo:hi]
        print(f"      {g:10s} n {hi-lo:4d} start {s[lo:hi].min():6.2f}..{s[lo:hi].max():6.2f} end {e[lo:hi].min():6.2f}..{e[lo:hi].max():6.2f} "
              f"duration min {d.min():6.2f} median {np.median(d):6.2f} max {d.max():6.2f}")
    if rep == 2 and groups[-1][0] == "critic16":
        lo, hi = groups[-1][1:]
        st = np.sort(s[lo:hi])
        print("      critic16 start deciles:", " ".join(f"{v:.1f}" for v in np.percentile(s[lo:hi], [0, 10, 25, 50, 75, 90, 100])))
        print("      critic16 end deciles:  ", " ".join(f"{v:.1f}" for v in np.percentile(e[lo:hi], [0, 10, 25, 50, 75, 90, 100])))
    w = t[1]
    nw = int((w[:, 1] > 0).sum())
    if nw:
        print(f"      weight-gradient launch: {nw} workgroups, first start {w[:nw, 0].min() - t0:.2f}, last end {w[:nw, 1].max() - t0:.2f}")
ssa._lib.lib.ssac_debug_timeline(0)
