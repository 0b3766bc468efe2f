"""a few launches of the ensemble-Q forward and the fused critic kernel at one batch size (PMC runs)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import super_sac_amd as ssa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N, in_dim, H = 10, 23, 256
dev = torch.device("cuda"); lib = ssa._lib.lib
ar = ssa.engine.MlpArena(N, in_dim, H, 1, dev)
for j in range(N):
    for seg in ssa.engine.SEGS:
        v = ar.view(j, seg); v.copy_(torch.randn(v.shape) * 0.05)
ws = ssa.engine.Workspace(dev)
x = torch.randn(B, in_dim, device=dev); td = torch.randn(B, 1, device=dev)
h1 = torch.empty(N, B, H, device=dev); h2 = torch.empty_like(h1); dz2 = torch.empty_like(h1); dz1 = torch.empty_like(h1)
q = torch.empty(N, B, 1, device=dev); dq = torch.empty_like(q)
tiles = int(lib.ssac_fused_row_tiles(C.byref(ar.desc()), B, N)); parts = torch.zeros(N * tiles * 2, device=dev)
for _ in range(6):
    ssa.engine.mlp_forward(ar, x, in_dim, 0, B, ws, "q", save=False)
    ssa._lib.check(lib.ssac_critic_fwd_bwd_fused(C.byref(ar.desc()), x.data_ptr(), in_dim, B, td.data_ptr(), 0, 0, 1, 0, 0,
        float(N), h1.data_ptr(), h2.data_ptr(), q.data_ptr(), dq.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), parts.data_ptr(), 0, ssa.engine.stream()))
torch.cuda.synchronize()
