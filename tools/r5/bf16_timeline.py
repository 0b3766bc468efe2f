"""(LAB build) per-class workgroup timeline of the bf16 update's two launches (bf_chain_pc_kernel: producers, critic tiles,
consumers; bf_wgrad_kernel):   python tools/r5/bf16_timeline.py [B] [N]"""
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
B = int(args[0]) if args else 256
N = int(args[1]) if len(args) > 1 else 10
ssa.learning.USE_GRAPHS = False
critic, _ = bc.build(17, 6, B, N, 2, precision="bf16")
for _ in range(5):
    critic()
tl = torch.zeros(2048, dtype=torch.int64, device="cuda")
ssa._lib.check(ssa._lib.lib.ssac_debug_timeline(tl.data_ptr()))
T = (B + 31) // 32
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
    nc = N * T
    groups = (("producers", 0, T), ("critic", T, T + nc), ("consumers", T + nc, n))
    print(f"[{rep}] bf16 chained launch {n} workgroups, last end {e.max():.2f} us")
    for g, lo, hi in groups:
        if hi <= lo:
            continue
        d = (e - s)[lo:hi]
        print(f"      {g:10s} n {hi-lo:4d} start {s[lo:hi].min():6.2f}..{s[lo:hi].max():6.2f} end {e[lo:hi].min():6.2f}..{e[lo:hi].max():6.2f} "
              f"duration min {d.min():6.2f} median {np.median(d):6.2f} max {d.max():6.2f}")
    w = t[1]
    nw = int((w[:, 1] > 0).sum())
    if nw:
        ws, we = w[:nw, 0] - t0, w[:nw, 1] - t0
        print(f"      weight-gradient launch: {nw} workgroups, first start {ws.min():.2f}, last end {we.max():.2f}; durations min {np.min(we - ws):.2f} "
              f"median {np.median(we - ws):.2f} max {np.max(we - ws):.2f}")
        # logical id of a workgroup (ssac_xcd_contiguous); per net: 16 fc2 tiles, 4 fc1 tiles (in_dim <= 64: 4 row tiles x 1), 1 head
        b = np.arange(nw); x, slot, q, r = b & 7, b >> 3, nw >> 3, nw & 7
        L = np.where(x < r, x * (q + 1), r * (q + 1) + (x - r) * q) + slot
        per = nw // N
        tt = L % per
        d = we - ws
        for name, m in (("fc2 tile 0 (loss statistics)", tt == 0), ("fc2 tiles 1..", (tt > 0) & (tt < 16)), ("fc1 tiles", (tt >= 16) & (tt < per - 1)),
                        ("head", tt == per - 1)):
            if m.any():
                print(f"         {name:30s} n {int(m.sum()):3d} duration min {d[m].min():6.2f} median {np.median(d[m]):6.2f} max {d[m].max():6.2f}  end max {(we - ws.min())[m].max():6.2f}")
ssa._lib.lib.ssac_debug_timeline(0)
