"""time the merged weight-gradient launch with and without the per-row scale (rank-1 backward)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import super_sac_amd as ssa
dev = torch.device("cuda")
N, B, H, in_dim = 10, 512, 256, 23
ar = ssa.engine.MlpArena(N, in_dim, H, 1, dev)
ar.params.normal_(0, 0.05)
x = torch.randn(B, in_dim, device=dev)
h1 = torch.randn(N, B, H, device=dev).relu_(); h2 = torch.randn(N, B, H, device=dev).relu_()
dz2 = torch.randn(N, B, H, device=dev); dz1 = torch.randn(N, B, H, device=dev)
dq = torch.randn(N, B, 1, device=dev) * 1e-3
opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1, device=dev))], lr=3e-4)
adam = ssa.engine.adam_group(opt, dev)
ss = torch.zeros(N * ssa.engine.wgrad_tiles_total(ar), device=dev)
for name, kw in (("plain", {}), ("row-scaled", {"rowscale": dq})):
    for _ in range(5):
        ssa.engine.weight_grads(ar, x, in_dim, 0, h1, h2, dq, dz2, dz1, B, adam=adam, adam_key=("c", 0), sumsq=ss, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        ssa.engine.weight_grads(ar, x, in_dim, 0, h1, h2, dq, dz2, dz1, B, adam=adam, adam_key=("c", 0), sumsq=ss, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 10:.2f} us per launch")
