"""the bf16 ensemble-Q forward alone (bench.py's secondary.config2_bf16 rows) + its agreement with the tile kernel
    python tools/bf16_fwd_rows.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
import super_sac_amd as ssa
from super_sac_amd import engine
from super_sac_amd._lib import check, lib
dev = torch.device("cuda")
if os.environ.get("SSAC_BF16_FWD_FORM"):   # A/B: 0 = the streaming kernel everywhere, 1 = register-chained where it applies
    check(lib.ssac_bf16_fwd_form(int(os.environ["SSAC_BF16_FWD_FORM"])))
rows = bench.bf16_rows(ssa, dev)["ensemble_q_kernel_bf16"]["rows"]
for k, v in rows.items():
    print(k, v)
# agreement of the streaming kernel (large batches) with the per-tile kernel on the same rows
N, IN = 10, 23
ar = engine.MlpArena(N, IN, 256, 1, dev)
torch.manual_seed(1)
ar.params.copy_(torch.randn_like(ar.params) * 0.05)
ar.enable_bf16()
B = 8192 + 37
x = torch.randn(B, IN, device=dev)
y_big = torch.empty(N, B, 1, device=dev)
check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), 0, N, x.data_ptr(), IN, B, y_big.data_ptr(), engine.stream()))
y_ref = torch.empty(N, B, 1, device=dev)
for b0 in range(0, B, 1024):   # small calls take the per-tile kernel
    n = min(1024, B - b0)
    yy = torch.empty(N, n, 1, device=dev)
    check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), 0, N, x[b0:b0 + n].data_ptr(), IN, n, yy.data_ptr(), engine.stream()))
    y_ref[:, b0:b0 + n] = yy
torch.cuda.synchronize()
print("stream vs tile kernel: max abs diff", float((y_big - y_ref).abs().max()), "of max |y|", float(y_ref.abs().max()))
