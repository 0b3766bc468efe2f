"""bf16 ensemble-Q forward: both large-batch forms over a range of batch sizes (N 10, 23 -> 256 -> 256 -> 1):  python tools/r5/bf_sizes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from super_sac_amd import engine
from super_sac_amd._lib import check, lib
dev = torch.device("cuda")
N, IN = 10, 23
ar = engine.MlpArena(N, IN, 256, 1, dev)
torch.manual_seed(1)
ar.params.copy_(torch.randn_like(ar.params) * 0.05)
ar.enable_bf16()
print("| B | streaming kernel us | register-chained us | TFLOP/s (chained) |")
print("|---|---|---|---|")
for B in (4096, 6144, 8192, 12288, 16384, 32768, 65536, 131072, 262144):
    x = torch.randn(B, IN, device=dev)
    y = torch.empty(N, B, 1, device=dev)
    us = []
    for form in (0, 1):
        check(lib.ssac_bf16_fwd_form(form))
        run = lambda: check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), 0, N, x.data_ptr(), IN, B, y.data_ptr(), engine.stream()))
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3 / 20)
    fl = 2.0 * B * N * (IN * 256 + 256 * 256 + 256)
    print(f"| {B} | {us[0]:.1f} | {us[1]:.1f} | {fl / us[1] / 1e6:.0f} |", flush=True)
check(lib.ssac_bf16_fwd_form(1))
