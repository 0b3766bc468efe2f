"""time the shift-fused first convolution against the two launches it replaces:
    python tools/first_shift_time.py [dmc|atari]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import super_sac_amd as ssa
cfg = sys.argv[1] if len(sys.argv) > 1 else "dmc"
B, C, co, k, s, H, pad, div, shift = (512, 9, 32, 3, 2, 84, 4, 255.0, -0.5) if cfg == "dmc" else (1024, 4, 32, 8, 4, 84, 4, 255.0, 0.0)
dev = torch.device("cuda:0")
lib, st, check = ssa._lib.lib, ssa.engine.stream(), ssa._lib.check
rows = 4000
store = torch.randint(0, 256, (rows, C, H, H), dtype=torch.uint8, device=dev)
idx = torch.randint(0, rows, (B,), device=dev)
sh = torch.randint(0, 2 * pad + 1, (B, 2), device=dev)
w = torch.randn(co, C, k, k, device=dev) * 0.1
b = torch.randn(co, device=dev) * 0.1
Ho = (H - k) // s + 1
img = torch.empty(B, C, H, H, device=dev)
y0 = torch.empty(B, Ho, Ho, co, device=dev)
y1 = torch.empty_like(y0)
n_aug = int(0.75 * B)
def two():
    check(lib.ssac_drq_shift(store.data_ptr(), 1, idx.data_ptr(), B, C, H, pad, sh.data_ptr(), 0, 0, n_aug, img.data_ptr(), st))
    check(lib.ssac_conv_first_fwd(img.data_ptr(), w.data_ptr(), b.data_ptr(), y0.data_ptr(), B, C, H, H, co, k, s, div, shift, st))
def one():
    check(lib.ssac_conv_first_shift_fwd(store.data_ptr(), idx.data_ptr(), sh.data_ptr(), pad, n_aug, w.data_ptr(), b.data_ptr(),
                                        y1.data_ptr(), B, C, H, co, k, s, div, shift, st))
for name, fn in (("shift + conv1 (two launches)", two), ("conv1 with the shift in its staging", one)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 50
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{cfg}: {name}: {e0.elapsed_time(e1) / n * 1e3:.1f} us")
print("bit-identical:", torch.equal(y0, y1), " rows per band:", lib.ssac_conv_first_shift_supported(C, co, k, s, H, B, pad))
