"""(LAB build) which workgroups of the weight-gradient launch are the slow ones, launch after launch?"""
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
B, N = 512, 10
ssa.learning.USE_GRAPHS = False
critic, _ = bc.build(17, 6, B, N, 2)
for _ in range(5):
    critic()
tl = torch.zeros(2048, dtype=torch.int64, device="cuda")
ssa._lib.check(ssa._lib.lib.ssac_debug_timeline(tl.data_ptr()))
for rep in range(16):
    tl.zero_()
    critic()
    torch.cuda.synchronize()
    t = tl.cpu().numpy().reshape(2, 512, 2) * 10e-3
    for li, name in ((0, "chain"), (1, "wgrad")):
        a = t[li]
        n = int((a[:, 1] > 0).sum())
        a = a[:n]
        t0 = a[:, 0].min()
        s, e = a[:, 0] - t0, a[:, 1] - t0
        d = e - s
        order = np.argsort(-e)[:3]
        med = np.median(d[:160]) if li == 1 else np.median(d[32:192])
        print(f"[{rep:2d}] {name}: last end {e.max():6.2f} median tile {med:5.2f} | slowest: " +
              "  ".join(f"bid {int(b)} (xcd {int(b) % 8}) start {s[b]:.2f} end {e[b]:.2f}" for b in order))
ssa._lib.lib.ssac_debug_timeline(0)
