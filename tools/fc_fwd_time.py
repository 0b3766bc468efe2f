"""time the encoder fc's split-K forward, tiled kernel against the operand stream:
    python tools/fc_fwd_time.py [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import super_sac_amd as ssa
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 50, 39200)
dev = torch.device("cuda:0")
lib, st, check = ssa._lib.lib, ssa.engine.stream(), ssa._lib.check
X, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
def run(fn, kps):
    slices = (K + kps - 1) // kps
    part = torch.empty(slices, M, N, device=dev)
    def f():
        check(fn(X.data_ptr(), K, W.data_ptr(), K, part.data_ptr(), M, N, K, kps, st))
        check(lib.ssac_reduce_slices_bias(part.data_ptr(), slices, M, N, b.data_ptr(), out.data_ptr(), N, st))
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3, slices, out.clone()
t, sl, ref = run(lib.ssac_linear_fwd_splitk, 832)
print(f"tiled split-K ({sl} slices) + reduce: {t:.1f} us")
for kps in (824, 616, 512, 416):
    t, sl, o = run(lib.ssac_linear_fwd_stream, kps)
    print(f"operand stream ({sl} slices of {kps}) + reduce: {t:.1f} us   max |diff| {float((o - ref).abs().max()):.2e}")
