"""time a layer's weight gradient, slice form against the whole-image LDS form:
    python tools/conv_wgrad_time.py B H ci co k s rows_per_slice"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import super_sac_amd as ssa
B, H, ci, co, k, s, rps = (int(v) for v in sys.argv[1:8])
dev = torch.device("cuda:0")
lib, st, check = ssa._lib.lib, ssa.engine.stream(), ssa._lib.check
Ho = (H - k) // s + 1
x = torch.rand(B, H, H, ci, device=dev)
dy = torch.randn(B, Ho, Ho, co, device=dev)
sl0 = int(lib.ssac_conv_wgrad_slices(B, Ho, Ho, rps))
sl1 = int(lib.ssac_conv_wgrad_img_slices(B, H, H, ci, co, k, s))
pw0, pb0 = torch.empty(sl0, co, ci, k, k, device=dev), torch.empty(sl0, co, device=dev)
pw1, pb1 = torch.empty(max(sl1, 1), co, ci, k, k, device=dev), torch.empty(max(sl1, 1), co, device=dev)
def slices():
    check(lib.ssac_conv_wgrad(dy.data_ptr(), x.data_ptr(), pw0.data_ptr(), pb0.data_ptr(), B, H, H, ci, co, k, s, rps, st))
def img():
    check(lib.ssac_conv_wgrad_img(dy.data_ptr(), x.data_ptr(), pw1.data_ptr(), pb1.data_ptr(), B, H, H, ci, co, k, s, st))
for name, fn in ((f"slice form ({sl0} slices of {rps} pixels)", slices), (f"whole images in LDS ({sl1} slices)", img)):
    if fn is img and not sl1:
        print("whole-image form: not covered"); continue
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{sys.argv[1:8]}: {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
if sl1:
    a, b_ = pw0.sum(0), pw1.sum(0)
    print("max |difference| / max |gradient|:", float((a - b_).abs().max() / a.abs().max()))
