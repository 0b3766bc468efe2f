"""back-to-back launch time of the weight-gradient launch variants at the headline shape (B 512, N 10, 23 -> 256 -> 256 -> 1)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import super_sac_amd as ssa
dev = torch.device("cuda")
B, in_dim, H, N = 512, 23, 256, 10
if len(sys.argv) > 2:
    B, N = int(sys.argv[1]), int(sys.argv[2])
print('B', B, 'N', N)
ar = ssa.engine.MlpArena(N, in_dim, H, 1, dev); ar.params.normal_(std=0.05)
x = torch.randn(B, in_dim, device=dev)
h1 = torch.randn(N, B, H, device=dev).relu(); h2 = torch.randn(N, B, H, device=dev).relu()
dz2 = torch.randn(N, B, H, device=dev); dz1 = torch.randn(N, B, H, device=dev)
q = torch.randn(N, B, 1, device=dev); td = torch.randn(B, 1, device=dev); dq = torch.randn(N, B, 1, device=dev)
grp = ssa.engine.AdamGroup(torch.optim.Adam([torch.zeros(1)], lr=3e-4), dev); grp.advance()
ttot = ssa.engine.wgrad_tiles_total(ar)
ss = torch.zeros(N * ttot, device=dev)
parts = torch.zeros(N * 2, device=dev)
tgt = ar.params.clone()
E = ssa.engine
def t(fn, n=200):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
lib, chk, st = ssa._lib.lib, ssa._lib.check, E.stream()
m_, v_ = grp.moments_for("k", ar.params)
d = ar.desc()
def layer(l, xin, ld, sx, dy):
    return lambda: chk(lib.ssac_mlp_layer_wgrad(C.byref(d), l, 0, N, xin.data_ptr(), ld, sx, dy.data_ptr(), H, B * H, B,
                                                m_.data_ptr(), v_.data_ptr(), grp.ctl.ptr, 0, 0, 0, 0, 0.0, st))
print(f"fc2 only            {t(layer(1, h1, H, B * H, dz2)):6.2f} us")
print(f"fc1 only            {t(layer(0, x, in_dim, 0, dz1)):6.2f} us")
print(f"fc2+fc1 (pair)      {t(lambda: E.weight_grads(ar, x, in_dim, 0, h1, h2, dq, dz2, dz1, B, adam=grp, adam_key='k', sumsq=ss) if False else chk(lib.ssac_mlp_wgrad_fc12(C.byref(d), 0, N, x.data_ptr(), in_dim, 0, h1.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), B, m_.data_ptr(), v_.data_ptr(), grp.ctl.ptr, 0, ss.data_ptr() + 4 * ar.tiles(0), ss.data_ptr(), ttot, 0, 0.0, st))):6.2f} us")
print(f"all (head merged)   {t(lambda: E.weight_grads(ar, x, in_dim, 0, h1, h2, dq, dz2, dz1, B, adam=grp, adam_key='k', sumsq=ss)):6.2f} us")
print(f"all + rowscale      {t(lambda: E.weight_grads(ar, x, in_dim, 0, h1, h2, dq, dz2, dz1, B, adam=grp, adam_key='k', sumsq=ss, rowscale=dq)):6.2f} us")
lf = dict(q=q, td_ptr=td.data_ptr(), spec_ptr=0, weight_ptr=0, popart_ptr=0, pop=0, denom=float(N), partials=parts)
print(f"all + loss fold     {t(lambda: E.weight_grads(ar, x, in_dim, 0, h1, h2, dq, dz2, dz1, B, adam=grp, adam_key='k', sumsq=ss, lossfold=lf)):6.2f} us")
print(f"  + polyak target   {t(lambda: E.weight_grads(ar, x, in_dim, 0, h1, h2, dq, dz2, dz1, B, adam=grp, adam_key='k', sumsq=ss, lossfold=lf, target=tgt, tau=0.005)):6.2f} us")
