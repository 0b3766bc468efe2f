"""time the first layer's weight gradient, gather form against LDS-band form:
    python tools/first_wgrad_time.py [dmc|atari]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import super_sac_amd as ssa
cfg = sys.argv[1] if len(sys.argv) > 1 else "dmc"
B, C, co, k, s, H, div, shift = (512, 9, 32, 3, 2, 84, 255.0, -0.5) if cfg == "dmc" else (1024, 4, 32, 8, 4, 84, 255.0, 0.0)
dev = torch.device("cuda:0")
lib, st, check = ssa._lib.lib, ssa.engine.stream(), ssa._lib.check
Ho = (H - k) // s + 1
img = torch.rand(B, C, H, H, device=dev) * 255
dy = torch.randn(B, Ho, Ho, co, device=dev)
rps = int(os.environ.get("RPS", "1792" if cfg == "dmc" else "896"))
sl0 = int(lib.ssac_conv_wgrad_slices(B, Ho, Ho, rps))
sl1 = int(lib.ssac_conv_first_wgrad_band_slices(B, C, H, H, co, k, s))
pw0, pb0 = torch.empty(sl0, co, C, k, k, device=dev), torch.empty(sl0, co, device=dev)
pw1, pb1 = torch.empty(sl1, co, C, k, k, device=dev), torch.empty(sl1, co, device=dev)
def gather():
    check(lib.ssac_conv_first_wgrad(dy.data_ptr(), img.data_ptr(), pw0.data_ptr(), pb0.data_ptr(), B, C, H, H, co, k, s, div, shift, rps, st))
def band():
    check(lib.ssac_conv_first_wgrad_band(dy.data_ptr(), img.data_ptr(), pw1.data_ptr(), pb1.data_ptr(), B, C, H, H, co, k, s, div, shift, st))
for name, fn in ((f"gather form ({sl0} slices)", gather), (f"LDS bands ({sl1} slices)", band)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{cfg}: {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
a, b_ = pw0.sum(0), pw1.sum(0)
print("max |difference| / max |gradient|:", float((a - b_).abs().max() / a.abs().max()))
